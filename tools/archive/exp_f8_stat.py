#!/usr/bin/env python3
"""Run-to-run determinism of the z|r conv (both arithmetic forms) over many launches: counts launches whose output differs from the first."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from importlib import import_module
L, ops = import_module("cer-mvs_amd._lib"), import_module("cer-mvs_amd.ops")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
h, w = 296, 400
P = h * w
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
rnd = lambda *s, lo=-1.0, hi=1.0: (lo + (hi - lo) * torch.rand(*s, generator=g))
U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
net = torch.tanh(rnd(P, 64, lo=-2, hi=2)).to(dev)
c1 = torch.relu(rnd(P, 64, lo=-1, hi=2)).to(dev)
disp = rnd(P, lo=0.0005, hi=0.0025).to(dev)
wzr, wq = rnd(128, 177, 3, 3, lo=-0.05, hi=0.05), rnd(64, 177, 3, 3, lo=-0.05, hi=0.05)
src_s = [(64, 2, U), (49, 1, Dp), (64, 2, R)]
net_s, c1_s = ops.to_frag16(net, h, w, U), ops.to_frag16(c1, h, w, R)
po = ops.PackedConv3x3(wzr, None, [(64, 0), (49, 1), (64, 0)], dev)
ref = ops.s16_layout(ops.conv3x3(po, [net, disp, c1], h, w, L.EPI_LINEAR), h, w, L.S16_ACC32)
for name, wt in (("zr 128", wzr), ("q 64", wq)):
    for f8 in (False, True):
        pc = ops.PackedConvS16(wt, None, src_s, dev, corr_fp8=f8)
        first, bad, worst = None, 0, 0.0
        for rep in range(n):
            o = ops.conv3x3_s16(pc, [net_s, disp, c1_s], h, w, L.EPI_LINEAR)
            if first is None:
                first = o.clone()
            elif not torch.equal(o, first):
                bad += 1
            if name.startswith("zr"):
                worst = max(worst, float((o - ref).abs().max()))
        print(f"{name} f8={f8}: {bad} of {n - 1} launches differ from the first; max |diff| vs f16x3 kernel {worst:.3e}", flush=True)
