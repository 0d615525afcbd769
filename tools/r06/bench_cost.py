#!/usr/bin/env python3
"""Cost volume at the bench workload (1600x1184, 10 views), both stages, as RAFT.forward builds it (split operand planes, compact level-0 rows,
fused view-mean scale): sustained time of the loaded library's epipolar-line-tile build (events around 10 back-to-back builds) and its distance
from the wave-per-pixel walk (cer_cost_build_algo(1): the reference's fp32 expressions) on the same inputs.  Stage 1 starts from the disparity a
one-stage forward produces.  usage: [CER_MVS_LIB=...] python tools/archive/r05/bench_cost.py [label]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT, _lib as L, ops
from cer_mvs_amd.projective import pij_matrices
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

label = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda")
H, W, V = 1184, 1600, 10
cascade = [(64, 64, 16), (-1, 320, 16)]
model = RAFT(cascade=cascade, test_mode=True, gru_precision="s16f8")
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
h, w = H // 4, W // 4
P = h * w
lib = L.load()
with torch.no_grad():
    imgs = images.to(dev).float() * (2 / 255.0) - 1
    net_l, inp_l, f1, f2 = model.encode(imgs, list(range(1, V + 1)))
    split = (ops.feat_split(f1), ops.feat_split(f2))
    intr4 = intr.clone(); intr4[:, :, :2] /= 4
    Pij = pij_matrices(poses[0], intr4[0], [0] * V, list(range(1, V + 1))).to(dev)
    (D0, i0, _), (D1, i1, _) = model.stages()
    m0 = RAFT(cascade=cascade[:1], test_mode=True, gru_precision="s16f8")
    m0.load_state_dict(fill_state_dict(m0.state_dict(), seed=5), strict=False)
    m0 = m0.to(dev).eval()
    d1 = m0(images.to(dev), poses.to(dev), intr.to(dev), scale=scale).reshape(-1).float().contiguous()
    disp0 = torch.zeros(P, device=dev)

    def t(fn, reps=10):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    for stage, (D, inc, d, s0) in enumerate(((D0, i0, disp0, True), (D1, i1, d1, False))):
        build = lambda: ops.cost_build(f1, f2, Pij, d, D, inc, s0, h, w, 3, fold=True, pyramid_scale=1.0 / V, split=split, compact=True,
                                       two_term=os.environ.get("CER_COST_X2", "0") == "1")      # (CER_COST_X2=1: the two-term form, round 6)
        a = build()[0].clone()
        us = t(build)
        prev = lib.cer_cost_build_algo(1)
        try:
            b = ops.cost_build(f1, f2, Pij, d, D, inc, s0, h, w, 3, fold=True, pyramid_scale=1.0 / V)[0]
        finally:
            lib.cer_cost_build_algo(prev)
        a, b = a[:, :D], b[:, :D]
        print(f"{label}: stage {stage} D={D}: {us:8.1f} us;  vs walk: rel L1 {float((a - b).abs().sum() / b.abs().sum()):.3e}  max |diff| "
              f"{float((a - b).abs().max()):.3e} of max |value| {float(b.abs().max()):.3e}")
