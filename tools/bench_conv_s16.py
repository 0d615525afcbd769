#!/usr/bin/env python3
"""A/B timing of the update block's convolutions at BASELINE configs[1] shapes (296 x 400 feature pixels): round-2 s16 kernels
(csrc/conv_s16.hip) against the round-1 f16x3 kernels, interleaved rounds in one process, HIP events on the launch stream,
random (not zero) operands.  usage: python tools/bench_conv_s16.py [--reps N] [--rounds R] [--size HxW] [--mt M]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cer_mvs_amd import _lib as L, ops                                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--size", default="296x400")
    ap.add_argument("--mt", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--no-init", action="store_true", help="experiment: run the gate / GRU convs without the hoisted init term")
    ap.add_argument("--f8", action="store_true", help="the s16 side in the fp8-correction form (CER_EPI_CORR_FP8: gru_precision='s16f8')")
    args = ap.parse_args()
    h, w = (int(x) for x in args.size.split("x"))
    P = h * w
    dev = torch.device("cuda")
    ops.TILE_MT = args.mt
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s, lo=-1.0, hi=1.0: (lo + (hi - lo) * torch.rand(*s, generator=g))
    U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
    net = torch.tanh(rnd(P, 64, lo=-2, hi=2)).to(dev)
    c1 = torch.relu(rnd(P, 64, lo=-1, hi=2)).to(dev)
    disp = rnd(P, lo=0.0005, hi=0.0025).to(dev)
    wzr, wq = rnd(128, 177, 3, 3, lo=-0.05, hi=0.05), rnd(64, 177, 3, 3, lo=-0.05, hi=0.05)
    wc, bc = rnd(64, 64, 3, 3, lo=-0.1, hi=0.1), rnd(64, lo=-0.1, hi=0.1)
    w1, b1 = rnd(256, 64, 3, 3, lo=-0.08, hi=0.08), rnd(256, lo=-0.1, hi=0.1)
    w2 = rnd(1, 256, 3, 3, lo=-0.05, hi=0.05)
    initzr, initq = rnd(P, 128, lo=-0.3, hi=0.3).to(dev), rnd(P, 64, lo=-0.3, hi=0.3).to(dev)
    src_s = [(64, 2, U), (49, 1, Dp), (64, 2, R)]
    f8 = args.f8
    s = {"corr2": ops.PackedConvS16(wc, bc, [(64, 2, R)], dev, corr_fp8=f8), "zr": ops.PackedConvS16(wzr, None, src_s, dev, corr_fp8=f8),
         "q": ops.PackedConvS16(wq, None, src_s, dev, corr_fp8=f8), "d1": ops.PackedConvS16(w1, b1, [(64, 2, U)], dev, corr_fp8=f8)}
    proj_s = ops.delta_proj_pack_s16(w2, dev)
    src_o = [(64, 0), (49, 1), (64, 0)]
    o = {"corr2": ops.PackedConv3x3(wc, bc, [(64, 0)], dev), "zr": ops.PackedConv3x3(wzr, None, src_o, dev),
         "q": ops.PackedConv3x3(wq, None, src_o, dev), "d1": ops.PackedConv3x3(w1, b1, [(64, 0)], dev)}
    proj_o = ops.delta_proj_pack(w2, dev)
    e = lambda c: torch.empty(P, c, device=dev)
    es = lambda c: torch.zeros(ops.s16_pixels(h, w), c, device=dev)
    acc32 = lambda t: ops.s16_layout(t, h, w, L.S16_ACC32)
    # s16 buffers
    net_s, c1_s = ops.to_frag16(net, h, w, U), ops.to_frag16(c1, h, w, R)
    c2_s, z_s, rn_s, net2_s, T_s = es(64), es(64), es(64), es(64), torch.empty(2, 9, P, device=dev)
    initzr_s, initq_s = (None, None) if args.no_init else (acc32(initzr), acc32(initq))
    # f16x3 buffers (split32)
    net_o, c1_o = ops.split32(net), ops.split32(c1)
    c2_o, z_o, rn_o, net2_o, T_o = e(64), e(64), e(64), e(64), torch.empty(2, 9, P, device=dev)
    cases = {
        "corr2 64->64 relu": (
            lambda: ops.conv3x3_s16(s["corr2"], [c1_s], h, w, L.EPI_RELU, out=c2_s, log2s_out=R),
            lambda: ops.conv3x3(o["corr2"], [c1_o], h, w, L.EPI_RELU, out=c2_o, kinds=[3], out_split=True), 2 * 9 * 64 * 64),
        "z|r gates 177->128": (
            lambda: ops.conv3x3_s16(s["zr"], [net_s, disp, c2_s], h, w, L.EPI_GATES, out=z_s, out2=rn_s, aux=net_s, init=initzr_s, log2s_out=U, log2s_aux=U),
            lambda: ops.conv3x3(o["zr"], [net_o, disp, c2_o], h, w, L.EPI_GATES, out=z_o, out2=rn_o, aux=net_o, init=initzr, kinds=[3, 1, 3],
                                out_split=True, aux_split=True), 2 * 9 * 177 * 128),
        "q gru 177->64": (
            lambda: ops.conv3x3_s16(s["q"], [rn_s, disp, c2_s], h, w, L.EPI_GRU, out=net2_s, aux=net_s, aux2=z_s, init=initq_s, log2s_out=U, log2s_aux=U),
            lambda: ops.conv3x3(o["q"], [rn_o, disp, c2_o], h, w, L.EPI_GRU, out=net2_o, aux=net_o, aux2=z_o, init=initq, kinds=[3, 1, 3],
                                out_split=True, aux_split=True), 2 * 9 * 177 * 64),
        "delta 64->256 fused": (
            lambda: ops.conv3x3_s16(s["d1"], [net_s], h, w, L.EPI_DELTA, out=T_s, aux=proj_s),
            lambda: ops.conv3x3(o["d1"], [net_o], h, w, L.EPI_DELTA, mode="f16x3", out=T_o, aux=proj_o, kinds=[3]), 2 * 9 * 64 * 256 + 2 * 9 * 256),
    }
    only = set(x for x in args.only.split(",") if x)
    res = {}
    for name, (fs, fo, flop_px) in cases.items():
        if only and not any(k in name for k in only):
            continue
        fs(); fo()
        torch.cuda.synchronize()
        ts, to = [], []
        for _ in range(args.rounds):
            for fn, acc in ((fs, ts), (fo, to)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                acc.append(1e3 * e0.elapsed_time(e1) / args.reps)
        ms, mo = sorted(ts)[len(ts) // 2], sorted(to)[len(to) // 2]
        tf = flop_px * P / (ms * 1e-6) / 1e12
        res[name] = {"s16_us": ms, "s16_min_us": min(ts), "f16x3_us": mo, "f16x3_min_us": min(to), "s16_fp32eq_TF": tf,
                     "frac_of_833TF": tf / (2500 / 3)}
        print(f"{name:22s}  s16 {ms:7.1f} us (min {min(ts):7.1f})   f16x3 {mo:7.1f} us (min {min(to):7.1f})   s16: {tf:6.1f} TF fp32-eq = "
              f"{100 * tf / (2500 / 3):4.1f} % of 833 TF", flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"size": [h, w], "mt": args.mt, "results": res}, f, indent=1)


if __name__ == "__main__":
    main()
