#!/usr/bin/env python3
"""cost_build on the REAL cfg2 synthetic scene (encoded features, true epipolar geometry), both stages - HIP events.
The stage-1 input disparity is the model's own output.  A/B a library build with CER_MVS_LIB=..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cer_mvs_amd import RAFT, ops
from cer_mvs_amd.projective import pij_matrices
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

H, W, V, cascade = bench.WORKLOADS["dtu_1600x1184_v10_it32"]
dev = torch.device("cuda")
model = RAFT(cascade=cascade, test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
with torch.no_grad():
    out = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale)
    s = float(scale)
    h, w = H // 4, W // 4
    disp1 = (out / s).reshape(-1).contiguous()
    p = poses.clone().float(); p[..., :3, 3] *= s
    k = intr.clone().float(); k[:, :, :2] /= 4
    Pij = pij_matrices(p[0], k[0], [0] * V, list(range(1, V + 1))).to(dev)
    net_l, inp_l, f1, f2 = model.encode(images.to(dev).float() * (2 / 255.0) - 1, list(range(1, V + 1)))
    stages = list(model.stages())
    from cer_mvs_amd import _lib as L
    lib = L.load()
    split = (ops.feat_split(f1), ops.feat_split(f2))
    for st, (D, incre, T) in enumerate(stages):
        d_in = torch.zeros(h * w, device=dev) if st == 0 else disp1
        res = {}
        for algo, name in ((1, "walk"), (0, "lines")):
            lib.cer_cost_build_algo(algo)
            fn = lambda: ops.cost_build(f1, f2, Pij, d_in, D, incre, st == 0, h, w, model.update_block.num_levels, fold=True,
                                        pyramid_scale=1.0 / V, split=split, two_term=os.environ.get("CER_COST_X2", "0") == "1")
            res[name] = fn()[0]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            print(f"stage {st}: D={D} cost_build[{name}] {1e3 * e0.elapsed_time(e1) / 5:.1f} us")
        lib.cer_cost_build_algo(0)
        a, b = res["walk"], res["lines"]
        print(f"   max |walk - lines| = {float((a - b).abs().max()):.3e}  (max |walk| = {float(a.abs().max()):.3e}, rel-L1 {float((a - b).abs().sum() / a.abs().sum()):.3e})")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.feat_split(f1); ops.feat_split(f2)
    e1.record(); torch.cuda.synchronize()
    print(f"feat_split (f1 + f2): {1e3 * e0.elapsed_time(e1) / 5:.1f} us")
