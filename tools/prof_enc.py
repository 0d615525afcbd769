#!/usr/bin/env python3
"""Time the encoder engine stage by stage at cfg2 (11 images of 1184x1600) - HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cer_mvs_amd import RAFT, _lib as L
from cer_mvs_amd.encoder_hip import HipEncoder
from cer_mvs_amd.synthetic import fill_state_dict

dev = torch.device("cuda")
model = RAFT(test_mode=True); model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
eng = HipEncoder(model.fnet, dev)
N, H, W = 11, 1184, 1600
x = torch.rand(N, 3, H, W, device=dev) * 2 - 1
rec = []
orig_conv, orig_merge, orig_stats = eng._conv, eng._merge, eng._stats
def wrap(name, fn):
    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **k); e1.record(); rec.append((name(*a, **k) if callable(name) else name, e0, e1)); return out
    return inner
eng._conv = wrap(lambda c, *a, **k: f"conv{c.taps}_s{c.stride}_{c.cin}->{c.cout}", orig_conv)
eng._merge = wrap("merge", orig_merge)
lib = L.load()
for _ in range(2):
    rec.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.features(x, n_ref=1); e1.record(); torch.cuda.synchronize()
print("total features()", e0.elapsed_time(e1), "ms")
for name, a, b in rec: print(f"{name:28s} {a.elapsed_time(b)*1e3:9.1f} us")
