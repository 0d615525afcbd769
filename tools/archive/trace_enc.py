#!/usr/bin/env python3
"""Phase cycle sums of the streaming layer-1 encoder conv (debug build of the library with -DES_TRACE=1):
  make -C cer-mvs_amd/csrc && hipcc ... -DES_TRACE=1 (see DESIGN) ; CER_MVS_LIB=<trace lib> python tools/archive/trace_enc.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT                                                # noqa: E402
from cer_mvs_amd.encoder_hip import HipEncoder                              # noqa: E402
from cer_mvs_amd.synthetic import fill_state_dict                           # noqa: E402

dev = torch.device("cuda")
model = RAFT(test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
eng = HipEncoder(model.fnet, dev)
N, h, w = 11, 592, 800
x = torch.randn(N, h * w, 32, device=dev)
st = torch.stack([torch.zeros(N * 32, device=dev), torch.ones(N * 32, device=dev)], 1).contiguous()
c = eng.blocks[0][0]
nblocks = 2 * torch.cuda.get_device_properties(0).multi_processor_count
trace = torch.zeros(nblocks * 4 * 16 * 2, device=dev, dtype=torch.float32)
for _ in range(3):
    eng._conv(c, x, N, h, w, st, True, out2=trace)
torch.cuda.synchronize()
t = trace.view(torch.int64).cpu().numpy().reshape(nblocks, 4, 16).astype(np.float64)
tiles = t[:, :, 9]
names = ["commit (VALU + LDS writes)", "vmcnt(0)+barrier", "halo issue + 18 MFMA steps", "barrier", "epilogue", "barrier", "wait for the prefetched halo"]
tot = t[:, :, 8].mean()
print(f"blocks {nblocks}, tiles per block {tiles.mean():.1f}, cycles per block {tot:.0f} = {tot / tiles.mean():.0f} per tile")
for k, n in enumerate(names):
    print(f"  {n:28s} {t[:, :, k].mean() / tiles.mean():8.0f} cycles per tile  ({100 * t[:, :, k].mean() / tot:4.1f} %)")
