#!/usr/bin/env python3
"""Cycle-stamp trace of the s16 z|r conv (variant build -DSX_TRACE=1: make -C cer-mvs_amd/csrc variants/libcermvs_sxtrace.so;
run with CER_MVS_LIB=.../variants/libcermvs_sxtrace.so).  Per wave: entry, prologue end, every 18-step body, loop end, exit,
HW_ID.  Prints residency (waves per SIMD over time), lifetimes and phase lengths."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cer_mvs_amd import _lib as L, ops                                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="296x400")
    ap.add_argument("--mt", type=int, default=4)
    ap.add_argument("--conv", default="zr", choices=["zr", "q"])
    ap.add_argument("--f8", action="store_true", help="fp8-correction form (CER_EPI_CORR_FP8): stamps after every 32-channel chunk")
    ap.add_argument("--f6", action="store_true", help="FP6-correction form (CER_EPI_CORR_FP6, round 6): same stamps as --f8")
    args = ap.parse_args()
    if args.f6:
        args.f8 = 6
    h, w = (int(x) for x in args.size.split("x"))
    P = h * w
    dev = torch.device("cuda")
    ops.TILE_MT = args.mt
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s, lo=-1.0, hi=1.0: (lo + (hi - lo) * torch.rand(*s, generator=g))
    U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
    net = ops.to_frag16(torch.tanh(rnd(P, 64, lo=-2, hi=2)).to(dev), h, w, U)
    c2 = ops.to_frag16(torch.relu(rnd(P, 64, lo=-1, hi=2)).to(dev), h, w, R)
    disp = rnd(P, lo=0.0005, hi=0.0025).to(dev)
    pc = ops.PackedConvS16(rnd(128, 177, 3, 3, lo=-0.05, hi=0.05), None, [(64, 2, U), (49, 1, Dp), (64, 2, R)], dev, corr_fp8=args.f8)
    init = ops.s16_layout(rnd(P, 128, lo=-0.3, hi=0.3).to(dev), h, w, L.S16_ACC32)
    PP = ops.s16_pixels(h, w)
    th = 2 * args.mt * (1 if args.conv == "zr" else 2)
    nblk = ((h + th - 1) // th) * ((w + 15) // 16)
    trace = torch.zeros(ops.s16_pixels(h, w), 64, device=dev, dtype=torch.float32)  # 32 x u64 per wave, in a tensor-shaped buffer
    assert trace.numel() >= nblk * 4 * 32 * 2
    z, rn = torch.rand(PP, 64, device=dev), torch.empty(PP, 64, device=dev)
    if args.conv == "zr":
        run = lambda: ops.conv3x3_s16(pc, [net, disp, c2], h, w, L.EPI_GATES, out=z, out2=rn, aux=net, aux2=trace, init=init, log2s_out=U, log2s_aux=U)
    else:
        pq = ops.PackedConvS16(rnd(64, 177, 3, 3, lo=-0.05, hi=0.05), None, [(64, 2, U), (49, 1, Dp), (64, 2, R)], dev, corr_fp8=args.f8)
        initq = ops.s16_layout(rnd(P, 64, lo=-0.3, hi=0.3).to(dev), h, w, L.S16_ACC32)
        net2 = torch.empty(PP, 64, device=dev)
        run = lambda: ops.conv3x3_s16(pq, [net, disp, c2], h, w, L.EPI_GRU, out=net2, out2=trace, aux=net, aux2=z, init=initq, log2s_out=U, log2s_aux=U)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1)[:nblk * 4 * 32 * 2].view(np.uint64).reshape(nblk, 4, 32).astype(np.int64)
    n = t[:, :, 0]
    hw = t[:, :, 1]
    xcc = t[:, :, 2] & 0xF
    t_exit = t[:, :, 3]
    stamps = t[:, :, 8:]
    t0 = stamps[:, :, 0]
    # every XCD has its own cycle counter: rebase per XCD
    for x in np.unique(xcc):
        m = xcc == x
        b = t0[m].min()
        t_exit[m] -= b
        stamps[m] -= b
    t0 = stamps[:, :, 0]
    base = 0
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    simd = (hw >> 4) & 3
    print(f"blocks {nblk}, kernel span {(t_exit.max() - base)} cycles; distinct CUs {len(np.unique(cu))}")
    life = t_exit - t0
    pro = stamps[:, :, 1] - t0
    nn = int(n.max())
    loop_end = np.take_along_axis(stamps, (n - 1)[:, :, None], axis=2)[:, :, 0]
    main_loop = loop_end - stamps[:, :, 1]
    epi = t_exit - loop_end
    nsteps = t[:, :, 4]
    q = lambda a: "p10 %d  p50 %d  p90 %d  max %d" % tuple(np.percentile(a, [10, 50, 90, 100]))
    print("wave lifetime      ", q(life))
    print("prologue           ", q(pro))
    print("main loop          ", q(main_loop))
    print("main loop / step   ", q(main_loop / nsteps))
    print("epilogue           ", q(epi))
    body = np.diff(stamps[:, :, 1:1 + 5], axis=2)          # first four 9-tap groups
    print("9-tap group        ", q(body[body > 0]))
    # residency per CU (cycle counters are only comparable inside one CU): span of the CU's work, and how many blocks overlap
    spans, conc, gaps = [], [], []
    for c in np.unique(cu):
        m = cu == c
        a0, e0 = t0[m].min(), t_exit[m].max()
        spans.append(e0 - a0)
        conc.append(life[m].sum() / 4.0 / max(e0 - a0, 1))         # average number of resident blocks
        starts = np.sort(np.unique(t0[m] // 2000))                   # block starts (waves of a block start within ~2k cycles)
    spans, conc = np.array(spans), np.array(conc)
    print("per-CU busy span   ", q(spans))
    print("avg resident blocks per CU  p10 %.2f  p50 %.2f  p90 %.2f" % tuple(np.percentile(conc, [10, 50, 90])))
    nb = np.array([len(np.unique(np.nonzero(cu == c)[0])) for c in np.unique(cu)])
    print("blocks per CU      ", q(nb))
    for k in np.unique(nb):
        print(f"   CUs with {k} blocks: {np.sum(nb == k)}, busy span p50 {np.percentile(spans[nb == k], 50):.0f} max {spans[nb == k].max()}")
    print("waves per CU (whole launch): ", q(np.bincount(np.unique(cu, return_inverse=True)[1].reshape(-1))))
    # first-round blocks (start with the launch) against second-round blocks (start when a slot frees), per CU
    r1, r2, r2lone = [], [], []
    for c in np.unique(cu):
        m = np.nonzero((cu == c).any(axis=1))[0]                 # blocks of this CU
        st = t0[m, 0]
        order = np.argsort(st)
        first = st[order[0]]
        for k, b in enumerate(m[order]):
            per_step = (main_loop[b, 0] / max(nsteps[b, 0], 1))
            if t0[b, 0] - first < 20000:
                r1.append(per_step)
            else:
                (r2 if len(m) == 4 else r2lone).append(per_step)
    print("K-loop cycles/step  round 1      ", q(np.array(r1)))
    print("K-loop cycles/step  round 2 (CUs with 4 blocks)", q(np.array(r2)))
    print("K-loop cycles/step  round 2 (CUs with 3 blocks: runs alone)", q(np.array(r2lone)))
    # lifetimes by tile position (border tiles run the rim correction) and by launch order
    if args.conv == "zr":
        tx = (w + 15) // 16
        ty = (h + th - 1) // th
        bid = np.arange(nblk)

        def xcd_order(n, j):
            qq, rr, xcd_b, kk = n >> 3, n & 7, j & 7, j >> 3
            return np.where(xcd_b < rr, xcd_b * (qq + 1), rr * (qq + 1) + (xcd_b - rr) * qq) + kk
        # the kernel's block -> tile map for convs with a disparity source: rim tiles first (columns, then rows), interior in XCD order
        nbord = 2 * ty + 2 * (tx - 2)
        i_int = xcd_order(nblk - nbord, np.maximum(bid - nbord, 0))
        r2 = bid - 2 * ty
        tyi = np.where(bid < 2 * ty, bid >> 1, np.where(bid < nbord, np.where(r2 & 1, ty - 1, 0), 1 + i_int // (tx - 2)))
        txi = np.where(bid < 2 * ty, np.where(bid & 1, tx - 1, 0), np.where(bid < nbord, 1 + (r2 >> 1), 1 + i_int % (tx - 2)))
        lf = life[:, 0]
        started = t0[:, 0]
        for name, m in (("top row", tyi == 0), ("bottom row", tyi == ty - 1), ("left / right column", ((txi == 0) | (txi == tx - 1)) & (tyi > 0) & (tyi < ty - 1)),
                        ("interior", (tyi > 0) & (tyi < ty - 1) & (txi > 0) & (txi < tx - 1))):
            print(f"   {name:22s} {int(m.sum()):4d} blocks: life p50 {np.percentile(lf[m], 50):.0f} p90 {np.percentile(lf[m], 90):.0f}; start p50 {np.percentile(started[m], 50):.0f}")
        print(f"   last finishing blocks (tile rows): {sorted(tyi[np.argsort(t_exit[:, 0])[-12:]].tolist())}")
    ng = 4 if args.f8 else 8
    g = np.diff(stamps[:, 0, 1:1 + ng + 1], axis=1)             # the tensor groups (f8: 32-channel chunks) of wave 0
    print("cycles per 9-tap group, by group index (p50): ", [int(np.percentile(g[:, k][g[:, k] > 0], 50)) for k in range(g.shape[1])])
    tail = stamps[:, 0, 1 + ng + 1] - stamps[:, 0, 1 + ng]
    print("disparity section (6 steps)", q(tail))


if __name__ == "__main__":
    main()
