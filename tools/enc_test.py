import sys, time, torch
sys.path.insert(0,'/root/repo')
from cer_mvs_amd import RAFT
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
dev=torch.device('cuda')
model = RAFT(test_mode=True); model.load_state_dict(fill_state_dict(model.state_dict(), seed=5)); model=model.to(dev).eval()
images,_,_,_ = synthetic_scene(1184,1600,10,seed=0)
imgs = images.to(dev).float()*(2/255.)-1
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
with torch.no_grad():
    print("default", t(lambda: model.encode(imgs, list(range(1,11)))))
    torch.backends.cudnn.benchmark=True
    print("benchmark=True", t(lambda: model.encode(imgs, list(range(1,11)))))
    m2 = model.to(memory_format=torch.channels_last)
    x = imgs[0].contiguous(memory_format=torch.channels_last)
    print("channels_last fnet only", t(lambda: m2.fnet(x)))
    torch.backends.cudnn.benchmark=False
    print("nchw fnet only", t(lambda: model.fnet(imgs[0])))
    print("cnet only", t(lambda: model.cnet(imgs[:, [0]])))
    with torch.autocast("cuda", dtype=torch.float16):
        print("amp fnet", t(lambda: model.fnet(imgs[0])))
