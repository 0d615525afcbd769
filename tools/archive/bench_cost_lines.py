#!/usr/bin/env python3
"""Cost volume at the bench workload (1600x1184, 10 views): the multi-line form (cer_cost_lines_form 1, round 4; CER_COST_LINES_NW / _OUTD pick its variant) against the one-line
form (0, round 3), both stages; stage 1 starts from the disparity a one-stage forward produces.  Prints sustained times (events
around 10 back-to-back builds) and the largest difference between the two volumes."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT, _lib as L, ops
from cer_mvs_amd.projective import pij_matrices
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

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
    intr4 = intr.clone(); intr4[:, :, :2] /= 4
    Pij = pij_matrices(poses[0], intr4[0], [0] * V, list(range(1, V + 1))).to(dev)
    (D0, i0, _), (D1, i1, _) = model.stages()
    m0 = RAFT(cascade=cascade[:1], test_mode=True, gru_precision="s16f8")
    m0.load_state_dict(fill_state_dict(m0.state_dict(), seed=5), strict=False)
    m0 = m0.to(dev).eval()
    d_full = m0(images.to(dev), poses.to(dev), intr.to(dev), scale=scale)
    d1 = d_full.reshape(-1).float().contiguous()
    assert d1.numel() == P
    disp0 = torch.zeros(P, device=dev)

    def t(fn, reps=10):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    for stage, (D, inc, d, s0) in enumerate(((D0, i0, disp0, True), (D1, i1, d1, False))):
        vols = {}
        for form in (0, 1):
            lib.cer_cost_lines_form(form)
            build = lambda: ops.cost_build(f1, f2, Pij, d, D, inc, s0, h, w, 3, fold=True, pyramid_scale=1.0 / V)
            vols[form] = build()[0].clone()
            us = t(build)
            print(f"stage {stage} D={D} form {form} ({'several lines' if form == 1 else 'one line'} per block): {us:8.1f} us")
        a, b = vols[0], vols[1]
        n = D + D // 2 + D // 4
        print(f"   max |difference| {float((a[:, :n] - b[:, :n]).abs().max()):.3e} of max |value| {float(a[:, :n].abs().max()):.3e}; "
              f"rel L1 {float((a[:, :n] - b[:, :n]).abs().sum() / a[:, :n].abs().sum()):.3e}; bit-identical rows {float((a[:, :n] == b[:, :n]).all(1).float().mean()):.4f}")
    lib.cer_cost_lines_form(0)
