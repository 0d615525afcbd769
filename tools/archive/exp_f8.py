#!/usr/bin/env python3
"""fp8-correction form of the update-block convolutions (CER_EPI_CORR_FP8) against the shipped split-f16 form: per-conv difference and
timing at 296 x 400 (interleaved rounds, HIP events), then the whole forward (gru_precision "s16" vs "s16f8") on the bench workload."""
import argparse, os, sys, time, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from importlib import import_module
pkg = import_module("cer-mvs_amd")
L, ops = import_module("cer-mvs_amd._lib"), import_module("cer-mvs_amd.ops")


def rel(a, b):
    return float((a.double() - b.double()).abs().sum() / b.double().abs().sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="296x400")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--mt", type=int, default=0)
    ap.add_argument("--no-e2e", action="store_true")
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
    packs = {}
    for f8 in (False, True):
        packs[f8] = {"corr2": ops.PackedConvS16(wc, bc, [(64, 2, R)], dev, corr_fp8=f8), "zr": ops.PackedConvS16(wzr, None, src_s, dev, corr_fp8=f8),
                     "q": ops.PackedConvS16(wq, None, src_s, dev, corr_fp8=f8), "d1": ops.PackedConvS16(w1, b1, [(64, 2, U)], dev, corr_fp8=f8)}
    proj_s = ops.delta_proj_pack_s16(w2, dev)
    es = lambda c: torch.zeros(ops.s16_pixels(h, w), c, device=dev)
    acc32 = lambda t: ops.s16_layout(t, h, w, L.S16_ACC32)
    net_s, c1_s = ops.to_frag16(net, h, w, U), ops.to_frag16(c1, h, w, R)
    initzr_s, initq_s = acc32(initzr), acc32(initq)
    bufs = {f8: {"c2": es(64), "z": es(64), "rn": es(64), "n2": es(64), "T": torch.empty(2, 9, P, device=dev)} for f8 in (False, True)}
    # a common c2 / rn / z input for the later convs (from the s16 form)
    c2_in = ops.conv3x3_s16(packs[False]["corr2"], [c1_s], h, w, L.EPI_RELU, log2s_out=R).clone()
    z_in, rn_in = ops.conv3x3_s16(packs[False]["zr"], [net_s, disp, c2_in], h, w, L.EPI_GATES, aux=net_s, init=initzr_s, log2s_out=U, log2s_aux=U)
    z_in, rn_in = z_in.clone(), rn_in.clone()

    def case(name, f8):
        s, b = packs[f8], bufs[f8]
        if name == "corr2":
            return lambda: ops.conv3x3_s16(s["corr2"], [c1_s], h, w, L.EPI_RELU, out=b["c2"], log2s_out=R)
        if name == "zr":
            return lambda: ops.conv3x3_s16(s["zr"], [net_s, disp, c2_in], h, w, L.EPI_GATES, out=b["z"], out2=b["rn"], aux=net_s, init=initzr_s, log2s_out=U, log2s_aux=U)
        if name == "q":
            return lambda: ops.conv3x3_s16(s["q"], [rn_in, disp, c2_in], h, w, L.EPI_GRU, out=b["n2"], aux=net_s, aux2=z_in, init=initq_s, log2s_out=U, log2s_aux=U)
        return lambda: ops.conv3x3_s16(s["d1"], [net_s], h, w, L.EPI_DELTA, out=b["T"], aux=proj_s)

    outs = {"corr2": lambda b: ops.from_frag16(b["c2"], h, w, R), "zr": lambda b: torch.cat([ops.s16_layout(b["z"], h, w, L.S16_F32X8, inverse=True), ops.from_frag16(b["rn"], h, w, U)], 1),
            "q": lambda b: ops.from_frag16(b["n2"], h, w, U), "d1": lambda b: b["T"]}
    for name in ("corr2", "zr", "q", "d1"):
        f0, f1 = case(name, False), case(name, True)
        f0(); f1()
        torch.cuda.synchronize()
        a, b = outs[name](bufs[False]), outs[name](bufs[True])
        d = rel(b, a)
        fin = bool(torch.isfinite(b).all())
        ts = {False: [], True: []}
        for _ in range(args.rounds):
            for f8, fn in ((False, f0), (True, f1)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts[f8].append(1e3 * e0.elapsed_time(e1) / args.reps)
        m0, m1 = sorted(ts[False])[len(ts[False]) // 2], sorted(ts[True])[len(ts[True]) // 2]
        print(f"{name:6s} s16 {m0:7.1f} us   s16f8 {m1:7.1f} us   ({m0 / m1:.2f}x)   rel-L1 f8 vs s16 {d:.3e}  max|d| {float((a - b).abs().max()):.3e} finite {fin}", flush=True)
    if args.no_e2e:
        return
    import bench
    syn = import_module("cer-mvs_amd.synthetic")
    H, W, V, cascade = bench.WORKLOADS["dtu_1600x1184_v10_it32"]
    images, poses, intr, scale = syn.synthetic_scene(H, W, V, seed=0)
    x = (images.to(dev), poses.to(dev), intr.to(dev))
    res = {}
    for prec in ("s16", "s16f8"):
        model = pkg.RAFT(cascade=cascade, test_mode=True, gru_precision=prec)
        model.load_state_dict(syn.fill_state_dict(model.state_dict(), seed=5))
        model = model.to(dev).eval()
        with torch.no_grad():
            for _ in range(2):
                out = model(*x, scale=scale)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                out = model(*x, scale=scale)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 8
        res[prec] = out.clone()
        print(f"forward {prec}: {ms:.2f} ms per depth map (one at a time)", flush=True)
    print(f"e2e rel-L1 s16f8 vs s16: {rel(res['s16f8'], res['s16']):.3e}", flush=True)


if __name__ == "__main__":
    main()
