#!/usr/bin/env python3
"""Where a tile of cost_lines_kernel spends its cycles (the -DCL_TRACE build: `make -C cer-mvs_amd/csrc variants/libcermvs_cltrace.so`,
run with CER_MVS_LIB=cer-mvs_amd/csrc/variants/libcermvs_cltrace.so).  Bench scene, both stages."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cer_mvs_amd import RAFT, ops, _lib as L
from cer_mvs_amd.projective import pij_matrices
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

H, W, V, cascade = bench.WORKLOADS["dtu_1600x1184_v10_it32"]
dev = torch.device("cuda")
model = RAFT(cascade=cascade, test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
lib = ctypes.CDLL(L.LIB_PATH)
names = ["prologue: barrier b", "A wait + MFMA issue", "MFMA drain + dot stores", "barrier 1", "gather", "barrier 2", "leftovers", "rows out",
         "prologue: tile decode", "prologue: loads + projections", "prologue: barrier a", "prologue: band analysis"]
with torch.no_grad():
    out = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale)
    s = float(scale)
    h, w = H // 4, W // 4
    disp1 = (out / s).reshape(-1).contiguous()
    p = poses.clone().float(); p[..., :3, 3] *= s
    k = intr.clone().float(); k[:, :, :2] /= 4
    Pij = pij_matrices(p[0], k[0], [0] * V, list(range(1, V + 1))).to(dev)
    net_l, inp_l, f1, f2 = model.encode(images.to(dev).float() * (2 / 255.0) - 1, list(range(1, V + 1)))
    split = (ops.feat_split(f1), ops.feat_split(f2))
    for st, (D, incre, T) in enumerate(model.stages()):
        d_in = torch.zeros(h * w, device=dev) if st == 0 else disp1
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 32)()
        lib.cer_cost_lines_trace(buf, 1)
        ops.cost_build(f1, f2, Pij, d_in, D, incre, st == 0, h, w, 3, fold=True, pyramid_scale=1.0 / V, split=split)
        torch.cuda.synchronize()
        lib.cer_cost_lines_trace(buf, 0)
        t = list(buf)
        tiles = max(t[24], 1)
        tot = sum(t[:16])
        print(f"stage {st}: {tiles} tiles, {t[25] / tiles:.1f} chunks/tile, R avg {t[26] / tiles:.2f}, cycles/tile {tot / tiles:.0f}")
        for i, n in enumerate(names):
            print(f"   {n:26s} {t[i] / tiles:9.0f} cyc/tile  {100.0 * t[i] / tot:5.1f} %")
        it = max(t[27], 1)
        print(f"   gather iterations/tile (wave 0) {t[27] / tiles:.1f}, active lanes/iteration {t[28] / it:.1f}, behind samples/tile {t[29] / tiles:.2f}, leftover samples/tile {t[30] / tiles:.2f}")
