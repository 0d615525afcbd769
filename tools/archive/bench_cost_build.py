import sys, os, torch
sys.path.insert(0, os.getcwd())
from cer_mvs_amd import RAFT, _lib as L, ops
from cer_mvs_amd.projective import pij_matrices
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
dev = torch.device("cuda")
H, W, V = 1184, 1600, 10
model = RAFT(cascade=[(64, 64, 16), (-1, 320, 16)], test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
h, w = H // 4, W // 4
P = h * w
with torch.no_grad():
    imgs = images.to(dev).float() * (2 / 255.0) - 1
    net_l, inp_l, f1, f2 = model.encode(imgs, list(range(1, V + 1)))
    intr4 = intr.clone(); intr4[:, :, :2] /= 4
    Pij = pij_matrices(poses[0], intr4[0], [0] * V, list(range(1, V + 1))).to(dev)
    (D0, i0, _), (D1, i1, _) = model.stages()
    disp0 = torch.zeros(P, device=dev)
    lib = L.load()
    def t(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    for algo, name in ((1, "walk"), (0, "line tiles")):
        lib.cer_cost_build_algo(algo)
        us = t(lambda: ops.cost_build(f1, f2, Pij, disp0, D0, i0, True, h, w, 3, fold=True, pyramid_scale=0.1))
        print(f"stage0 {name:10s} {us:8.1f} us")
    lib.cer_cost_build_algo(0)
