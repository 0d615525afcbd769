#!/usr/bin/env python3
"""Per-kernel timing at BASELINE configs[1] shapes (1600x1184 -> 296x400, V=10) on realistic inputs
(features of the synthetic scene through the real encoders).  HIP events on the launch stream.
usage: python tools/archive/bench_kernels.py [--reps N] [--only name,name]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT, _lib as L, ops                                     # noqa: E402
from cer_mvs_amd import dist as cdist                                            # noqa: E402
from cer_mvs_amd.projective import pij_matrices                                  # noqa: E402
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene               # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--size", default="1184x1600")
    ap.add_argument("--views", type=int, default=10)
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    H, W = (int(x) for x in args.size.split("x"))
    V = args.views
    dev = torch.device("cuda")
    model = RAFT(cascade=[(64, 64, 16), (-1, 320, 16)], test_mode=True, gru_precision="f16x3")   # (s16 convs: tools/bench_conv_s16.py)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
    model = model.to(dev).eval()
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
    h, w = H // 4, W // 4
    P = h * w
    with torch.no_grad():
        imgs = images.to(dev).float() * (2 / 255.0) - 1
        net_l, inp_l, f1, f2 = model.encode(imgs, list(range(1, V + 1)))
        intr4 = intr.clone()
        intr4[:, :, :2] /= 4
        Pij = pij_matrices(poses[0], intr4[0], [0] * V, list(range(1, V + 1))).to(dev)
        ub = model.update_block
        res = {}

        def run(name, fn):
            if only and name not in only:
                return
            res[name] = timeit(fn, args.reps)
            print(f"{name:28s} {res[name]:10.1f} us", flush=True)

        disp0 = torch.zeros(P, device=dev)
        g = torch.Generator(device="cpu").manual_seed(0)
        disp1 = (0.0012 + 0.0004 * torch.rand(P, generator=g)).to(dev)
        smooth = torch.linspace(0.0010, 0.0018, w).repeat(h).to(dev)
        (D0, i0, _), (D1, i1, _) = model.stages()
        run("cost_build_stage0", lambda: ops.cost_build(f1, f2, Pij, disp0, D0, i0, True, h, w, 3, fold=True))
        run("cost_build_stage1_noisy", lambda: ops.cost_build(f1, f2, Pij, disp1, D1, i1, False, h, w, 3, fold=True))
        run("cost_build_stage1_smooth", lambda: ops.cost_build(f1, f2, Pij, smooth, D1, i1, False, h, w, 3, fold=True))
        vol, origin = ops.cost_build(f1, f2, Pij, disp0, D0, i0, True, h, w, 3, fold=True)
        run("pyramid", lambda: ops.pyramid(vol, D0, 3, 1.0 / V))
        p = ub.packed(0, dev)
        ws = ub.workspace(h, w, dev)
        hz, hq = ub.hoist(inp_l, h, w)
        dd = disp1.clone()
        run("lookup_encode", lambda: ops.lookup_encode(vol, origin, dd, p["w0t"], p["b0"], D0, i0, 3, 5, out=ws["c1"]))
        run("conv_corr2_64", lambda: ops.conv3x3(p["corr2"], [ws["c1"]], h, w, L.EPI_RELU, out=ws["c2"]))
        run("conv_zr_128", lambda: ops.conv3x3(p["zr_rest"], [net_l, dd, ws["c2"]], h, w, L.EPI_GATES, out=ws["z"], out2=ws["rn"], aux=net_l, init=hz))
        net2 = net_l.clone()
        run("conv_q_64", lambda: ops.conv3x3(p["q_rest"], [ws["rn"], dd, ws["c2"]], h, w, L.EPI_GRU, out=net2, aux=net_l, aux2=ws["z"], init=hq))
        run("conv_delta1_256", lambda: ops.conv3x3(p["d1"], [net_l], h, w, L.EPI_RELU, out=ws["hid"]))
        run("delta_tail", lambda: ops.delta_tail(ws["hid"], p["d2w"], p["d2b"], dd, h, w, disp_out=torch.empty_like(dd), want_delta=False))
        run("hoist_inp", lambda: ub.hoist(inp_l, h, w))
        run("encode_all_views", lambda: model.encode(imgs, list(range(1, V + 1))))


if __name__ == "__main__":
    main()
