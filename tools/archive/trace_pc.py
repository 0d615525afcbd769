#!/usr/bin/env python3
"""Phase cycle sums of the producer / consumer encoder convs (csrc/enc_pc.hip built with -DPC_TRACE=1):
  make -C cer-mvs_amd/csrc variants/libcermvs_pctrace.so; CER_MVS_LIB=cer-mvs_amd/csrc/variants/libcermvs_pctrace.so python tools/archive/trace_pc.py [32|64|s2] [dual]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT                                                # noqa: E402
from cer_mvs_amd import encoder_hip as E                                    # noqa: E402
from cer_mvs_amd.synthetic import fill_state_dict                           # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "32"
dual = len(sys.argv) > 2 and sys.argv[2] == "dual"
dev = torch.device("cuda")
model = RAFT(test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
eng = E.HipEncoder(model.fnet, dev)
N = 11
if which == "32":
    h, w, c = 592, 800, eng.blocks[0][0]
elif which == "64":
    h, w, c = 296, 400, eng.blocks[2][1]
else:
    h, w, c = 592, 800, eng.blocks[2][0]
x = torch.randn(N, h * w, c.cin, device=dev)
y = torch.randn(N, h * w, c.cin, device=dev)
st = torch.stack([torch.zeros(N * c.cin, device=dev), torch.ones(N * c.cin, device=dev)], 1).contiguous()
nblocks = torch.cuda.get_device_properties(0).multi_processor_count
trace = torch.zeros(nblocks * 8 * 8 * 2, device=dev, dtype=torch.float32)
inp = E._In(x, st, True, c.cin, y if dual else None, st if dual else None, True)
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng._pc(c, inp, N, h, w, out2=trace, want_stats=False)
    e1.record()
torch.cuda.synchronize()
print(f"{which} dual={dual}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (trace build)")
# sustained: 30 launches back to back (does the chip hold its clock under this kernel?)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
ev[0].record()
for i in range(30):
    eng._pc(c, inp, N, h, w, out2=trace, want_stats=False)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(30)]
print("  sustained launches (us): " + " ".join(f"{t:.0f}" for t in ts))
t = trace.view(torch.int64).cpu().numpy().reshape(nblocks, 8, 8).astype(np.float64)
P, Cw = t[:, :4], t[:, 4:]
units, tiles = P[:, :, 7].mean(), Cw[:, :, 7].mean()
print(f"blocks {nblocks}; producer units per block {units:.1f}, consumer tiles per block {tiles:.1f}; block life {P[:, :, 6].mean():.0f} / {Cw[:, :, 6].mean():.0f} cycles")
for k, n in enumerate(["wait for the halo loads", "commit (VALU + LDS writes)", "issue next loads", "barrier", "stats finalize"]):
    print(f"  P {n:28s} {P[:, :, k].mean() / units:8.0f} cycles per unit ({100 * P[:, :, k].mean() / P[:, :, 6].mean():4.1f} %)")
for k, n in enumerate(["weights / setup", "barrier", "K loop", "epilogue"]):
    print(f"  C {n:28s} {Cw[:, :, k].mean() / tiles:8.0f} cycles per tile ({100 * Cw[:, :, k].mean() / Cw[:, :, 6].mean():4.1f} %)")
