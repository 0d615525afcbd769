import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cer_mvs_amd import RAFT
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
dev = torch.device("cuda")
g = np.load("tests/golden/e2e_cfg1.npz")
H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
casc = [tuple(int(x) for x in c) for c in g["cascade"]]
images, poses, intr, scale = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
ref = torch.from_numpy(g["disp"]).double()
model = RAFT(cascade=casc, test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
model = model.to(dev).eval()
x = (images.to(dev), poses.to(dev), intr.to(dev))
bad = 0
with torch.no_grad():
    for i in range(int(sys.argv[1])):
        o = model(*x, scale=scale).cpu().double()
        e = float((o - ref).abs().sum() / ref.abs().sum())
        bad += e > 1e-6
print("solo n_bad", bad)
