#!/usr/bin/env python3
"""Cycle-stamp trace of the z|r conv (debug build of the library with -DHX_TRACE=1, see gru_f16x3.hip):
CER_MVS_LIB=cer-mvs_amd/csrc/variants/libcermvs_trace.so python tools/archive/trace_conv.py
Prints, for the blocks that shared one CU, the phase times of every wave: where a step's cycles go."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import _lib as L, ops                                     # noqa: E402

h, w = 296, 400
P = h * w
dev = torch.device("cuda")
torch.manual_seed(0)
r = lambda *s: (torch.randn(*s, device=dev) * 0.5)
net, c2 = torch.tanh(r(P, 64)), torch.relu(r(P, 64))
disp = (0.001 + 0.0005 * torch.rand(P, device=dev))
pc = ops.PackedConv3x3(torch.randn(128, 177, 3, 3) * 0.03, None, [(64, 0), (49, 1), (64, 0)], dev)
init = r(P, 128)
nblk = ((h + 3) // 4) * ((w + 31) // 32)
trace = torch.zeros(nblk * 8 * 64 * 4 * 2, device=dev, dtype=torch.float32)
for _ in range(3):
    ops.conv3x3(pc, [net, disp, c2], h, w, L.EPI_GATES, aux=net, aux2=trace, init=init, mode="f16x3")
torch.cuda.synchronize()
t = trace.view(torch.int64).cpu().numpy().reshape(nblk, 8, 64, 4)
hwid = t[:, 0, 62, 0]
xcc = t[:, 0, 62, 2] & 0xF
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 0x7
nsteps = t[:, 0, 62, 1]
key = xcc * 1000 + se * 100 + sh * 10 + cu
t0g = t[:, :, 63, 0].min()
print(f"blocks {nblk}; kernel span {(t[:, :, 63, 3].max() - t0g)} cycles (s_memtime units)")
vals, counts = np.unique(key, return_counts=True)
print(f"distinct CUs seen {len(vals)}; blocks per CU min/max {counts.min()}/{counts.max()}")
# pick the CU of block nblk//3
k0 = key[nblk // 3]
blocks = np.nonzero(key == k0)[0]
print(f"CU key {k0}: blocks {blocks.tolist()}")
for b in blocks:
    e = t[b, :, 63, :] - t0g
    print(f" block {b:4d} nsteps {nsteps[b]}  entry {e[:, 0].min():7d}  loop {e[:, 1].min():7d} .. {e[:, 2].max():7d}  exit {e[:, 3].max():7d}"
          f"  (prologue {int((e[:, 1] - e[:, 0]).mean())}, main {int((e[:, 2] - e[:, 1]).mean())}, epilogue {int((e[:, 3] - e[:, 2]).mean())})")
# per-step phase statistics over all blocks / waves
ns = int(np.median(nsteps))
sel = nsteps == ns
S = t[sel][:, :, :ns, :].astype(np.int64)
wait = S[..., 1] - S[..., 0]
k1 = S[..., 2] - S[..., 1]
k2 = S[..., 3] - S[..., 2]
nxt = S[:, :, 1:, 0] - S[:, :, :-1, 3]
print(f"steps/block {ns}; per-step means over {sel.sum()} blocks x 8 waves:")
print(f"  wait(DMA)+barrier {wait.mean():8.1f}   first k16 (LDS+6 MFMA issue) {k1.mean():8.1f}   second k16 {k2.mean():8.1f}   step->next {nxt.mean():8.1f}")
step_total = (S[:, :, 1:, 0] - S[:, :, :-1, 0])
print(f"  step period mean {step_total.mean():.1f}; by step index (mean over blocks/waves):")
per = step_total.mean(axis=(0, 1))
print("   period:", " ".join(f"{int(x)}" for x in per))
print("   wait  :", " ".join(f"{int(x)}" for x in wait.mean(axis=(0, 1))))
print("   k16a  :", " ".join(f"{int(x)}" for x in k1.mean(axis=(0, 1))))
print("   k16b  :", " ".join(f"{int(x)}" for x in k2.mean(axis=(0, 1))))
print("   gap   :", " ".join(f"{int(x)}" for x in nxt.mean(axis=(0, 1))))
print("per-block step periods (wave 0) and loop-top stamps relative to the first entry on this CU:")
base = min(t[b, :, 63, 0].min() for b in blocks)
for b in blocks:
    n = int(nsteps[b])
    tops = t[b, 0, :n, 0] - base
    print(f" block {b}: first top {tops[0]}, periods " + " ".join(str(int(x)) for x in np.diff(tops)))
for name, arr in (("wait+barrier", wait), ("k16a", k1), ("k16b", k2), ("gap(non-staging)", nxt)):
    a = arr.reshape(-1).astype(np.float64)
    if name.startswith("gap"):
        a = a[a < 2500]
    print(f" {name:18s} p10 {np.percentile(a,10):7.0f} p50 {np.percentile(a,50):7.0f} p90 {np.percentile(a,90):7.0f} p99 {np.percentile(a,99):7.0f} max {a.max():8.0f} mean {a.mean():7.0f}")
# the slowest wave of a block sets the pace: spread of loop-top stamps across the 8 waves of a block at the same step
spread = S[..., 0].max(axis=1) - S[..., 0].min(axis=1)
print(f" loop-top spread across the 8 waves of a block: p50 {np.percentile(spread,50):.0f} p90 {np.percentile(spread,90):.0f}")
after = S[..., 1].max(axis=1) - S[..., 1].min(axis=1)
print(f" post-barrier stamp spread: p50 {np.percentile(after,50):.0f} p90 {np.percentile(after,90):.0f}")
