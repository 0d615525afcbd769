#!/usr/bin/env python3
"""Sustained launch time of one producer / consumer encoder conv (40 back-to-back launches at cfg2's shape; the chip drops its
clock under these kernels, so the first launches of an idle chip are ~25 % faster than the steady state a forward sees).
usage: [CER_MVS_LIB=variant.so] [CER_ENC_F6=1] bench_pc.py [32|64|s2] [dual]      (CER_ENC_F6=1: the FP6-correction form)"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cer_mvs_amd import RAFT
from cer_mvs_amd import encoder_hip as E
from cer_mvs_amd.synthetic import fill_state_dict

which = sys.argv[1] if len(sys.argv) > 1 else "32"
dual = len(sys.argv) > 2 and sys.argv[2] == "dual"
dev = torch.device("cuda")
model = RAFT(test_mode=True); model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
eng = E.HipEncoder(model.fnet, dev)
eng.f6 = os.environ.get("CER_ENC_F6", "0") == "1"
N = 11
h, w, c = {"32": (592, 800, eng.blocks[0][0]), "64": (296, 400, eng.blocks[2][1]), "s2": (592, 800, eng.blocks[2][0])}[which]
x = torch.randn(N, h * w, c.cin, device=dev); y = torch.randn(N, h * w, c.cin, device=dev)
st = torch.stack([torch.zeros(N * c.cin, device=dev), torch.ones(N * c.cin, device=dev)], 1).contiguous()
inp = E._In(x, st, True, c.cin, y if dual else None, st if dual else None, True)
L = 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(L + 1)]
ev[0].record()
for i in range(L):
    eng._pc(c, inp, N, h, w, want_stats=False)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(L)]
print(f"{os.path.basename(os.environ.get('CER_MVS_LIB', 'libcermvs.so')):28s} {which} dual={dual} f6={eng.f6}: first {ts[0]:.0f} us, peak {max(ts):.0f}, steady (last 15) {statistics.mean(ts[-15:]):.0f} us")
