import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 300)
dist.init_process_group("gloo", rank=0, world_size=1)
from cer_mvs_amd import RAFT
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
dev = torch.device("cuda")
g = np.load("tests/golden/e2e_cfg1.npz")
H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
casc = [tuple(int(x) for x in c) for c in g["cascade"]]
images, poses, intr, scale = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
ref = torch.from_numpy(g["disp"]).double()
for shard in ("views", "slab"):
    model = RAFT(cascade=casc, test_mode=True, view_group=dist.group.WORLD, shard=shard)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    model = model.to(dev).eval()
    x = (images.to(dev), poses.to(dev), intr.to(dev))
    errs = []
    with torch.no_grad():
        for i in range(2):
            o = model(*x, scale=scale).cpu().double()
            errs.append(float((o - ref).abs().sum() / ref.abs().sum()))
    print(shard, "errs min %.3e max %.3e" % (min(errs), max(errs)), "n_bad", sum(e > 1e-6 for e in errs))
