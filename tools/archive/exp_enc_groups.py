#!/usr/bin/env python3
"""Experiment (round 4): does the encoder run faster when the 11 images go through the trunk in small groups, so that a
group's intermediate tensors (60.6 MB per image and half-resolution tensor) stay in the 256 MiB Infinity Cache between the
producing and the consuming kernel, instead of 11-image batches whose 667 MB tensors stream through HBM every pass?
Times the whole features() call (HIP events, 5 repetitions, median) for group sizes 11 (what ships), 6, 4, 3, 2, 1."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT
from cer_mvs_amd.encoder_hip import HipEncoder
from cer_mvs_amd.synthetic import fill_state_dict

dev = torch.device("cuda")
model = RAFT(test_mode=True); model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
eng = HipEncoder(model.fnet, dev)
N, H, W = 11, 1184, 1600
x = torch.rand(N, 3, H, W, device=dev) * 255
h, w = H // 4, W // 4
buf = torch.zeros(N - 1, (h + 4) * (w + 4), 64, device=dev)


def run(group):
    outs = []
    i = 0
    while i < N:
        j = min(N, i + group)
        if i == 0:
            ref, _, _, _ = eng.features(x[0:j], n_ref=1, border=2, scale=0.125, src_out=buf[0:j - 1] if j > 1 else None, raw=True)
            outs.append(ref)
        else:
            eng.features(x[i:j], n_ref=0, border=2, scale=0.125, src_out=buf[i - 1:j - 1], raw=True)
        i = j
    return outs[0]


base = None
for group in (11, 6, 4, 3, 2, 1, 11):
    ts = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ref = run(group); e1.record(); torch.cuda.synchronize()
        if rep:
            ts.append(e0.elapsed_time(e1))
    snap = (ref.clone(), buf.clone())
    if base is None:
        base = snap
    same = bool((snap[0] == base[0]).all() and (snap[1] == base[1]).all())
    print(f"group {group:2d}: features() median {statistics.median(ts):7.3f} ms  min {min(ts):7.3f}  (bit-identical to group 11: {same})", flush=True)
