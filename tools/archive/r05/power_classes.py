#!/usr/bin/env python3
"""Power and clock per kernel class (round 5): each kernel class of the forward is launched back to back for ~2.5 s while rocm-smi is polled
(average socket power / sclk over the second half of the window).  A class that runs at the board's power limit can only get faster by spending
fewer joules; one that runs far below it is latency-bound and has classic head-room.  usage: python tools/archive/r05/power_classes.py"""
import os, subprocess, sys, threading, time, re
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from cer_mvs_amd import RAFT, _lib as L, ops
from cer_mvs_amd.encoder_hip import HipEncoder
from cer_mvs_amd.projective import pij_matrices
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

dev = torch.device("cuda")
H, W, V = 1184, 1600, 10
h, w = H // 4, W // 4
P = h * w
g = torch.Generator().manual_seed(0)
rnd = lambda *s, lo=-1.0, hi=1.0: (lo + (hi - lo) * torch.rand(*s, generator=g))
U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
net = ops.to_frag16(torch.tanh(rnd(P, 64, lo=-2, hi=2)).to(dev), h, w, U)
c2 = ops.to_frag16(torch.relu(rnd(P, 64, lo=-1, hi=2)).to(dev), h, w, R)
disp = rnd(P, lo=0.0005, hi=0.0025).to(dev)
src = [(64, 2, U), (49, 1, Dp), (64, 2, R)]
pzr = ops.PackedConvS16(rnd(128, 177, 3, 3, lo=-0.05, hi=0.05), None, src, dev, corr_fp8=True)
pq = ops.PackedConvS16(rnd(64, 177, 3, 3, lo=-0.05, hi=0.05), None, src, dev, corr_fp8=True)
pc = ops.PackedConvS16(rnd(64, 64, 3, 3, lo=-0.1, hi=0.1), rnd(64, lo=-0.1, hi=0.1), [(64, 2, R)], dev, corr_fp8=True)
pd = ops.PackedConvS16(rnd(256, 64, 3, 3, lo=-0.08, hi=0.08), rnd(256, lo=-0.1, hi=0.1), [(64, 2, U)], dev, corr_fp8=True)
proj = ops.delta_proj_pack_s16(rnd(1, 256, 3, 3, lo=-0.05, hi=0.05), dev)
initzr = ops.s16_layout(rnd(P, 128, lo=-0.3, hi=0.3).to(dev), h, w, L.S16_ACC32)
initq = ops.s16_layout(rnd(P, 64, lo=-0.3, hi=0.3).to(dev), h, w, L.S16_ACC32)
PP = ops.s16_pixels(h, w)
z, rn, net2, c2o = torch.rand(PP, 64, device=dev), torch.empty(PP, 64, device=dev), torch.empty(PP, 64, device=dev), torch.empty(PP, 64, device=dev)
T = torch.empty(2, 9, P, device=dev)
vol = rnd(P, 64).to(dev)
w0t, b0 = rnd(33, 64).to(dev), rnd(64).to(dev)
c1o = torch.empty(PP, 64, device=dev)
org = disp.clone()
model = RAFT(test_mode=True, gru_precision="s16f8")
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
eng = HipEncoder(model.fnet, dev)
imgs = images[0].to(dev).float()
f1 = rnd(P, 64, lo=-4, hi=4).to(dev)
f2 = torch.zeros(V, (h + 4) * (w + 4), 64, device=dev)
f2.view(V, h + 4, w + 4, 64)[:, 2:-2, 2:-2] = rnd(V, h, w, 64, lo=-4, hi=4).to(dev)
intr4 = intr.clone(); intr4[:, :, :2] /= 4
Pij = pij_matrices(poses[0], intr4[0], [0] * V, list(range(1, V + 1))).to(dev)
split = (ops.feat_split(f1), ops.feat_split(f2))
d0 = torch.zeros(P, device=dev)
buf = torch.zeros(V, (h + 4) * (w + 4), 128, device=dev, dtype=torch.float16)
f1s = torch.empty(P, 128, device=dev, dtype=torch.float16)
inputs = (images.to(dev), poses.to(dev), intr.to(dev))

cases = {
    "z|r gates conv": lambda: ops.conv3x3_s16(pzr, [net, disp, c2], h, w, L.EPI_GATES, out=z, out2=rn, aux=net, init=initzr, log2s_out=U, log2s_aux=U),
    "q GRU conv": lambda: ops.conv3x3_s16(pq, [net, disp, c2], h, w, L.EPI_GRU, out=net2, aux=net, aux2=z, init=initq, log2s_out=U, log2s_aux=U),
    "delta head": lambda: ops.conv3x3_s16(pd, [net], h, w, L.EPI_DELTA, out=T, aux=proj),
    "corr2 conv": lambda: ops.conv3x3_s16(pc, [c2], h, w, L.EPI_RELU, out=c2o, out_split=True, log2s_out=R),
    "lookup": lambda: ops.lookup_encode(vol, org, disp.clone(), w0t, b0, 64, 0.0025 / 64, 3, 5, out=c1o, out_split=2, log2s=R, img_w=w),
    "cost volume stage 0": lambda: ops.cost_build(f1, f2, Pij, d0, 64, 0.0025 / 64, True, h, w, 3, fold=True, pyramid_scale=0.1, split=split, compact=True),
    "fnet (11 images)": lambda: eng.features_split(imgs, f1s, buf, n_ref=1, border=2, scale=0.125, raw=True),
    "whole forward": lambda: model(*inputs, scale=scale),
}
samples = []
stop = False
def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        pw = re.search(r"Power \(W\): ([0-9.]+)", out)
        ck = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        samples.append((time.time(), float(pw.group(1)) if pw else float("nan"), float(ck.group(1)) if ck else float("nan")))
th = threading.Thread(target=poll); th.start()
with torch.no_grad():
    for name, fn in cases.items():
        fn(); torch.cuda.synchronize()
        t0 = time.time(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < 2.5:
            for _ in range(20): fn()
            n += 20
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        t1 = time.time()
        sel = [(p, c) for (t, p, c) in samples if t0 + 1.2 <= t <= t1]
        pw = sum(p for p, _ in sel) / max(len(sel), 1); ck = sum(c for _, c in sel) / max(len(sel), 1)
        print(f"{name:22s} {1e3 * e0.elapsed_time(e1) / n:9.1f} us per launch   {pw:6.0f} W   {ck:5.0f} MHz   ({len(sel)} samples)", flush=True)
        time.sleep(1.0)
stop = True; th.join()
