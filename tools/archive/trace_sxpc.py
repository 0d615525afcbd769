#!/usr/bin/env python3
"""Phase cycle sums of the producer / consumer z|r gate convolution (csrc/conv_s16pc.hip built with -DSXPC_TRACE=1):
  make -C cer-mvs_amd/csrc variants/libcermvs_sxpctrace.so; CER_MVS_LIB=.../libcermvs_sxpctrace.so python tools/archive/trace_sxpc.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import _lib as L, ops

h, w = 296, 400
P = h * w
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
rnd = lambda *s, lo=-1.0, hi=1.0: (lo + (hi - lo) * torch.rand(*s, generator=g))
U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
net = torch.tanh(rnd(P, 64, lo=-2, hi=2)).to(dev)
c1 = torch.relu(rnd(P, 64, lo=-1, hi=2)).to(dev)
disp = rnd(P, lo=0.0005, hi=0.0025).to(dev)
wzr = rnd(128, 177, 3, 3, lo=-0.05, hi=0.05)
pzr = ops.PackedConvS16(wzr, None, [(64, 2, U), (49, 1, Dp), (64, 2, R)], dev, corr_fp8=True)
net_s, c1_s = ops.to_frag16(net, h, w, U), ops.to_frag16(c1, h, w, R)
init = ops.s16_layout(rnd(P, 128, lo=-0.3, hi=0.3).to(dev), h, w, L.S16_ACC32)
nblocks = torch.cuda.get_device_properties(0).multi_processor_count
trace = torch.zeros(ops.s16_pixels(h, w), 64, device=dev, dtype=torch.float32)   # (the wrapper checks aux2 as a frag-layout tensor)
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv3x3_s16(pzr, [net_s, disp, c1_s], h, w, L.EPI_GATES, aux=net_s, aux2=trace, init=init, log2s_out=U, log2s_aux=U)
    e1.record()
torch.cuda.synchronize()
print(f"z|r launch {e0.elapsed_time(e1) * 1e3:.1f} us (trace build)")
t = trace.view(-1)[:nblocks * 8 * 16 * 2].view(torch.int64).cpu().numpy().reshape(nblocks, 8, 16).astype(np.float64)
Pw, Cw = t[:, :4], t[:, 4:]
tiles = Cw[:, :, 15].mean()
print(f"blocks {nblocks}, tiles per block {tiles:.2f}; wave life P {Pw[:, :, 14].mean():.0f} / C {Cw[:, :, 14].mean():.0f} cycles")
for k, n in enumerate(["prologue", "wait b_0", "E-slice requests", "convert + store chunk", "issue next loads", "E-slice math + stores", "(disp store)",
                       "wait b_k", "left-over E-slices", "disparity generation", "wait b_D", "next tile chunk 0"]):
    print(f"  P {n:26s} {Pw[:, :, k].mean() / tiles:8.0f} cycles per tile ({100 * Pw[:, :, k].mean() / Pw[:, :, 14].mean():4.1f} %)")
for k, n in enumerate(["tile setup", "wait b_0", "taps 0-7 of the chunks", "wait b_k / b_D", "last taps", "disparity + rim", "dump"]):
    print(f"  C {n:26s} {Cw[:, :, k].mean() / tiles:8.0f} cycles per tile ({100 * Cw[:, :, k].mean() / Cw[:, :, 14].mean():4.1f} %)")
