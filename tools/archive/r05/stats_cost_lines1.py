#!/usr/bin/env python3
"""Phase cycles of the one-line cost-volume kernel (variant build -DCL_STATS=1; CER_MVS_LIB=.../variants/libcermvs_clstats.so): wave 0's cycle stamps
summed over all tiles of one build at the bench workload, both stages."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
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
fn = lib.cer_cost_lines1_stats
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = {3: "prologue (fragment requests, band analysis, first barrier)", 4: "projection of the lane's 8 samples -> descriptors", 5: "chunk: wait for fragments + 12 MFMAs + dots -> LDS",
         6: "chunk: barrier 1", 7: "chunk: next fragments requested + gather loop", 8: "chunk: barrier 2", 9: "leftover (direct path) + barrier", 10: "rows out"}
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
    out = (ctypes.c_ulonglong * 16)()
    for stage, (D, inc, d, s0) in enumerate(((D0, i0, torch.zeros(P, device=dev), True), (D1, i1, d1, False))):
        build = lambda: ops.cost_build(f1, f2, Pij, d, D, inc, s0, h, w, 3, fold=True, pyramid_scale=1.0 / V, split=split, compact=True)
        build(); torch.cuda.synchronize()
        fn(out, 1)
        build(); torch.cuda.synchronize()
        fn(out, 1)
        c = list(out)
        nt = max(c[0], 1)
        print(f"stage {stage} (D = {D}): {c[0]} tiles, {c[1] / nt:.2f} chunks per tile, {c[2] / nt:.0f} cycles per tile")
        for i in (3, 4, 5, 6, 7, 8, 9, 10):
            per = c[i] / nt
            print(f"   {names[i]:70s} {per:8.0f} cycles per tile ({100 * c[i] / max(c[2], 1):4.1f} %)" + (f"   = {c[i] / max(c[1], 1):6.0f} per chunk" if 5 <= i <= 8 else ""))
