#!/usr/bin/env python3
"""One process, two HIP streams: stream A runs the (experimental) lookup kernel in bursts and compares every output with the first one;
stream B runs ONE other kind of kernel in a loop.  Which co-resident kernel makes the packed lookup fail?
usage: CER_MVS_LIB=.../libcermvs_lkspec.so python tools/archive/repro_pair.py <companion> [launches]
companions: none lookup conv_s16_zr conv_s16_zr_f8 conv_s16_corr2 conv_f16x3_zr conv_fp32_zr delta_sum elementwise matmul"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import _lib as L, ops

comp = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = torch.device("cuda")
h, w, D, Lv, r = 296, 400, 64, 3, 5
P = h * w
g = torch.Generator().manual_seed(7)
rnd = lambda *s, lo=-1.0, hi=1.0: (lo + (hi - lo) * torch.rand(*s, generator=g))
vol = rnd(P, 112, lo=-8, hi=8).to(dev)
origin = torch.full((P,), 0.00125).to(dev)
incre = 0.0025 / 64
disp = rnd(P, lo=0.0, hi=60 * incre).to(dev)
wt, b = rnd(33, 64, lo=-0.5, hi=0.5).to(dev), rnd(64, lo=-0.5, hi=0.5).to(dev)
burst = 20
outs = [torch.zeros(ops.s16_pixels(h, w), 64, device=dev) for _ in range(burst)]
look = lambda o: ops.lookup_encode(vol, origin, disp, wt, b, D, incre, Lv, r, out=o, out_split=2, log2s=4, img_w=w)
first = torch.zeros_like(outs[0]); look(first)
# companions
U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
net = torch.tanh(rnd(P, 64, lo=-2, hi=2)).to(dev); c1 = torch.relu(rnd(P, 64, lo=-1, hi=2)).to(dev); dsp = rnd(P, lo=0.0005, hi=0.0025).to(dev)
wzr = rnd(128, 177, 3, 3, lo=-0.05, hi=0.05); wc, bc = rnd(64, 64, 3, 3, lo=-0.1, hi=0.1), rnd(64, lo=-0.1, hi=0.1)
es = lambda c: torch.zeros(ops.s16_pixels(h, w), c, device=dev)
fnB = None
if comp.startswith("conv_s16"):
    f8 = comp.endswith("_f8")
    net_s, c1_s = ops.to_frag16(net, h, w, U), ops.to_frag16(c1, h, w, R)
    c2_s, z_s, rn_s = es(64), es(64), es(64)
    if "corr2" in comp:
        pc = ops.PackedConvS16(wc, bc, [(64, 2, R)], dev, corr_fp8=f8)
        fnB = lambda: ops.conv3x3_s16(pc, [c1_s], h, w, L.EPI_RELU, out=c2_s, log2s_out=R)
    else:
        pc = ops.PackedConvS16(wzr, None, [(64, 2, U), (49, 1, Dp), (64, 2, R)], dev, corr_fp8=f8)
        fnB = lambda: ops.conv3x3_s16(pc, [net_s, dsp, c2_s], h, w, L.EPI_GATES, out=z_s, out2=rn_s, aux=net_s, log2s_out=U, log2s_aux=U)
elif comp in ("conv_f16x3_zr", "conv_fp32_zr"):
    po = ops.PackedConv3x3(wzr, None, [(64, 0), (49, 1), (64, 0)], dev)
    z_o, rn_o = torch.empty(P, 64, device=dev), torch.empty(P, 64, device=dev)
    mode = "f16x3" if "f16x3" in comp else "fp32"
    fnB = lambda: ops.conv3x3(po, [net, dsp, c1], h, w, L.EPI_GATES, out=z_o, out2=rn_o, aux=net, mode=mode)
elif comp == "lookup":
    o2 = torch.zeros_like(outs[0]); fnB = lambda: look(o2)
elif comp == "elementwise":
    a = torch.randn(8 * 1024 * 1024, device=dev); fnB = lambda: a.mul_(1.0001)
elif comp == "matmul":
    a = torch.randn(2048, 2048, device=dev, dtype=torch.float16); fnB = lambda: a @ a
elif comp == "delta_sum":
    T = torch.randn(2, 9, P, device=dev); d2 = torch.zeros(P, device=dev)
    lib = L.load()
    fnB = lambda: L.check(lib.cer_delta_sum_f32(L.dev_ptr(T, "T"), 2, 0.01, L.dev_ptr(T, "b"), L.dev_ptr(d2, "disp"), L.dev_ptr(d2, "d2"), h, w, L.cur_stream()), "delta_sum") if False else d2.add_(0)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
bad = 0
for it in range(0, n, burst):
    if fnB is not None:
        with torch.cuda.stream(sB):
            for _ in range(burst * (1 if comp != "elementwise" else 4)):
                fnB()
    with torch.cuda.stream(sA):
        for o in outs:
            look(o)
    torch.cuda.synchronize()
    bad += sum(0 if torch.equal(o, first) else 1 for o in outs)
print(f"companion {comp:16s}: {bad} of {n} lookup launches differ from the first")
