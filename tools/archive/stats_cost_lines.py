#!/usr/bin/env python3
"""Counters of the eight-line cost-volume kernel at the bench workload (library built with -DC8_STATS=1:
make -C cer-mvs_amd/csrc variants/libcermvs_c8stats.so; run with CER_MVS_LIB=cer-mvs_amd/csrc/variants/libcermvs_c8stats.so)."""
import ctypes, os, sys, torch
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
names = {0: "blocks", 1: "lines", 2: "windows (sum over blocks)", 3: "Wt (sum)", 4: "R' (sum)", 5: "R_w (sum over lines)", 6: "gather iterations (sum over lines)",
         7: "direct-path samples", 8: "cycles per line, total", 9: "cycles: prologue", 10: "cycles: fragments + MFMA + dots", 13: "cycles: gather",
         11: "cycles: commit + barrier", 14: "cycles: leftover loop", 15: "cycles: rows out", 12: "lines handed to the one-line form"}
with torch.no_grad():
    imgs = images.to(dev).float() * (2 / 255.0) - 1
    net_l, inp_l, f1, f2 = model.encode(imgs, list(range(1, V + 1)))
    intr4 = intr.clone(); intr4[:, :, :2] /= 4
    Pij = pij_matrices(poses[0], intr4[0], [0] * V, list(range(1, V + 1))).to(dev)
    (D0, i0, _), (D1, i1, _) = model.stages()
    m0 = RAFT(cascade=cascade[:1], test_mode=True, gru_precision="s16f8")
    m0.load_state_dict(fill_state_dict(m0.state_dict(), seed=5), strict=False)
    m0 = m0.to(dev).eval()
    d1 = m0(images.to(dev), poses.to(dev), intr.to(dev), scale=scale).reshape(-1).float().contiguous()
    disp0 = torch.zeros(P, device=dev)
    lib.cer_cost_lines_form(1)
    out = (ctypes.c_ulonglong * 32)()
    for stage, (D, inc, d, s0) in enumerate(((D0, i0, disp0, True), (D1, i1, d1, False))):
        ops.cost_build(f1, f2, Pij, d, D, inc, s0, h, w, 3, fold=True, pyramid_scale=1.0 / V)
        torch.cuda.synchronize()
        lib.cer_cost_lines_stats(out, 1)
        ops.cost_build(f1, f2, Pij, d, D, inc, s0, h, w, 3, fold=True, pyramid_scale=1.0 / V)
        torch.cuda.synchronize()
        lib.cer_cost_lines_stats(out, 1)
        c = list(out)
        nb, nl = max(c[0], 1), max(c[1], 1)
        print(f"stage {stage} (D = {D}): {c[0]} blocks, {c[1]} lines, {c[12]} lines handed over")
        print(f"   per block: windows {c[2] / nb:.1f}, Wt {c[3] / nb:.2f}, R' {c[4] / nb:.2f}; per line: R_w {c[5] / nl:.2f}, gather iterations {c[6] / nl:.1f} "
              f"({c[6] / max(c[2] * nl / nb, 1):.2f} per window), direct-path samples {c[7] / nl:.2f}")
        print(f"   cycles per line (wave): total {c[8] / nl:.0f} = prologue {c[9] / nl:.0f} + [fragments/MFMA/dots {c[10] / nl:.0f} + gather {c[13] / nl:.0f} + "
              f"commit/barrier {c[11] / nl:.0f} (of which commit incl. the wait for the pieces {c[16] / nl:.0f})] + leftover {c[14] / nl:.0f} + rows out {c[15] / nl:.0f}")
