#!/usr/bin/env python3
"""Time the encoder engine stage by stage at cfg2 (11 images of 1184x1600) - HIP events around every launching call.
usage: prof_enc.py [pc|tiled] [fnet|cnet]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT, _lib as L
from cer_mvs_amd import encoder_hip as E
from cer_mvs_amd.synthetic import fill_state_dict

E.ENGINE = sys.argv[1] if len(sys.argv) > 1 else "pc"
which = sys.argv[2] if len(sys.argv) > 2 else "fnet"
dev = torch.device("cuda")
model = RAFT(test_mode=True); model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
eng = E.HipEncoder(getattr(model, which), dev)
N, H, W = (11 if which == "fnet" else 1), 1184, 1600
x = torch.rand(N, 3, H, W, device=dev) * 255
rec = []
def wrap(name, fn):
    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **k); e1.record(); rec.append((name(*a, **k) if callable(name) else name, e0, e1)); return out
    return inner
eng._conv = wrap(lambda c, *a, **k: f"conv{c.taps}_s{c.stride}_{c.cin}->{c.cout}", eng._conv)
eng._pc = wrap(lambda c, x, *a, **k: f"pc{c.taps}_s{c.stride}_{c.cin}->{c.cout}{'_dual' if x.B is not None else ''}{'_mout' if k.get('merged') else ''}", eng._pc)
eng._merge = wrap("merge", eng._merge)
eng._stem = wrap("stem", eng._stem)
for _ in range(3):
    rec.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if which == "fnet": eng.features(x, n_ref=1, raw=True)
    else: eng.context(x, raw=True)
    e1.record(); torch.cuda.synchronize()
print(f"engine {E.ENGINE} {which}: total {e0.elapsed_time(e1):.3f} ms (host-paced events below include the stats reduce of each conv)")
for name, a, b in rec: print(f"  {name:34s} {a.elapsed_time(b)*1e3:9.1f} us")
# GPU-paced total: 5 back-to-back calls
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng2 = E.HipEncoder(getattr(model, which), dev)
torch.cuda.synchronize(); e0.record()
for _ in range(5):
    if which == "fnet": eng2.features(x, n_ref=1, raw=True)
    else: eng2.context(x, raw=True)
e1.record(); torch.cuda.synchronize()
print(f"engine {E.ENGINE} {which}: GPU-paced {e0.elapsed_time(e1)/5:.3f} ms per call")
