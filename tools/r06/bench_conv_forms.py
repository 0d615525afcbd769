#!/usr/bin/env python3
"""The update block's four convolutions at the bench shape (296 x 400 feature pixels) in the three arithmetic forms of csrc/conv_s16.hip - three
f16 terms ("s16"), correction terms on the fp8 matrix instruction ("s16f8"), on its FP6 form ("s16f6", round 6) - interleaved rounds in one
process, HIP events, random operands; and each form's distance from the all-f16 form (relative L1 of the outputs).
usage: python tools/r06/bench_conv_forms.py [--reps N] [--rounds R] [--size HxW]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import _lib as L, ops                                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--size", default="296x400")
ap.add_argument("--mt", type=int, default=0, help="force the tile height (ops.TILE_MT; 8 needs a -DSX_MT8=1 build of conv_s16.hip)")
args = ap.parse_args()
ops.TILE_MT = args.mt
h, w = (int(x) for x in args.size.split("x"))
P = h * w
dev = torch.device("cuda")
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
src_s = [(64, 2, U), (49, 1, Dp), (64, 2, R)]
forms = {"s16": False, "s16f8": True, "s16f6": 6}
pk = {f: {"corr2": ops.PackedConvS16(wc, bc, [(64, 2, R)], dev, corr_fp8=v), "zr": ops.PackedConvS16(wzr, None, src_s, dev, corr_fp8=v),
          "q": ops.PackedConvS16(wq, None, src_s, dev, corr_fp8=v), "d1": ops.PackedConvS16(w1, b1, [(64, 2, U)], dev, corr_fp8=v)} for f, v in forms.items()}
proj = ops.delta_proj_pack_s16(w2, dev)
es = lambda c: torch.zeros(ops.s16_pixels(h, w), c, device=dev)
acc32 = lambda t: ops.s16_layout(t, h, w, L.S16_ACC32)
net_s, c1_s = ops.to_frag16(net, h, w, U), ops.to_frag16(c1, h, w, R)
c2_in = ops.to_frag16(torch.relu(rnd(P, 64, lo=-1, hi=2)).to(dev), h, w, R)
rn_in = ops.to_frag16((torch.rand(P, 64, generator=g).to(dev) * net), h, w, U)
z_in = ops.s16_layout(torch.rand(P, 64, generator=g).to(dev), h, w, L.S16_F32X8)
initzr, initq = acc32(rnd(P, 128, lo=-0.3, hi=0.3).to(dev)), acc32(rnd(P, 64, lo=-0.3, hi=0.3).to(dev))
out = {f: {"c2": es(64), "z": es(64), "rn": es(64), "net2": es(64), "T": torch.empty(2, 9, P, device=dev)} for f in forms}
cases = {
    "corr2 64->64 relu": lambda f: ops.conv3x3_s16(pk[f]["corr2"], [c1_s], h, w, L.EPI_RELU, out=out[f]["c2"], log2s_out=R),
    "z|r gates 177->128": lambda f: ops.conv3x3_s16(pk[f]["zr"], [net_s, disp, c2_in], h, w, L.EPI_GATES, out=out[f]["z"], out2=out[f]["rn"], aux=net_s,
                                                     init=initzr, log2s_out=U, log2s_aux=U),
    "q gru 177->64": lambda f: ops.conv3x3_s16(pk[f]["q"], [rn_in, disp, c2_in], h, w, L.EPI_GRU, out=out[f]["net2"], aux=net_s, aux2=z_in, init=initq,
                                               log2s_out=U, log2s_aux=U),
    "delta 64->256 fused": lambda f: ops.conv3x3_s16(pk[f]["d1"], [net_s], h, w, L.EPI_DELTA, out=out[f]["T"], aux=proj),
}
outs_of = {"corr2 64->64 relu": [("c2", R)], "z|r gates 177->128": [("rn", U)], "q gru 177->64": [("net2", U)], "delta 64->256 fused": [("T", None)]}
for name, fn in cases.items():
    for f in forms:
        fn(f)
    torch.cuda.synchronize()
    ts = {f: [] for f in forms}
    for _ in range(args.rounds):
        for f in forms:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn(f)
            e1.record()
            torch.cuda.synchronize()
            ts[f].append(1e3 * e0.elapsed_time(e1) / args.reps)
    med = {f: sorted(v)[len(v) // 2] for f, v in ts.items()}
    errs = {}
    for key, sc in outs_of[name]:
        get = (lambda t: ops.from_frag16(t, h, w, sc).double()) if sc is not None else (lambda t: t.double())
        ref = get(out["s16"][key])
        for f in ("s16f8", "s16f6"):
            errs[f] = float((get(out[f][key]) - ref).abs().sum() / ref.abs().sum())
    print(f"{name:22s} " + "  ".join(f"{f} {med[f]:6.1f} us (min {min(ts[f]):6.1f})" for f in forms) +
          f"   rel-L1 from s16: s16f8 {errs['s16f8']:.2e}  s16f6 {errs['s16f6']:.2e}", flush=True)
