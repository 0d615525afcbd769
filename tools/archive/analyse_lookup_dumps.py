#!/usr/bin/env python3
"""Offline analysis of the dumps tools/archive/repro_lookup_ab.py writes: for every failing 16-pixel run, which lookup tap is wrong and what it was
replaced by (the 1x1 conv is inverted on the channels whose ReLU is open in both outputs)."""
import glob, sys, numpy as np, torch
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/lkdump/dump_*.pt"
tot = {}
for f in sorted(glob.glob(pat)):
    d = torch.load(f)
    if "pix" not in d:
        continue
    pix, vol, disp, origin = d["pix"], d["vol"], d["disp"], d["origin"]
    W, b = d["w"].reshape(64, 33).numpy(), d["b"].numpy()
    D, incre, log2s, img_w = d["D"], d["incre"], d["log2s"], d["img_w"]

    def decode(buf):
        raw = buf.contiguous().view(torch.float16).reshape(-1, 4, 2, 2, 32, 8)
        val = (raw[:, :, 0].float() + raw[:, :, 1].float()) / (2.0 ** log2s)
        return val.permute(0, 3, 1, 2, 4).reshape(val.shape[0], 32, 64)
    spec, gen = decode(d["spec_all"]), decode(d["generic_all"])
    mtl = torch.unique(d["allrows"] // 32).tolist()
    mtx = (img_w + 15) // 16
    for j, p in enumerate(pix.tolist()):
        y, x = p // img_w, p % img_w
        mt, slot = (y >> 1) * mtx + (x >> 4), ((y & 1) << 4) | (x & 15)
        k = mtl.index(mt)
        s, g = spec[k, slot].numpy(), gen[k, slot].numpy()
        act = (s > 0) & (g > 0)
        if np.abs(s - g).max() == 0 or act.sum() < 34:
            continue
        df = np.linalg.lstsq(W[act], (s - g)[act], rcond=None)[0]
        t = int(np.argmax(np.abs(df)))
        lv, jj = t // 11, t % 11
        off, n = [0, 64, 96][lv], [64, 32, 16][lv]
        c = max((float(disp[j]) - float(origin[j])) / incre + D // 2, 0.0)
        xx = np.float32(c) / np.float32(1 << lv)
        fx, w = int(np.floor(xx)), float(xx - np.floor(xx))
        i0 = fx + jj - 5
        v0 = float(vol[j, off + i0]) if 0 <= i0 < n else 0.0
        v1 = float(vol[j, off + i0 + 1]) if 0 <= i0 + 1 < n else 0.0
        tol = lambda a, b_: abs(a - b_) < 3e-3 * max(1.0, abs(b_))
        kind = "v[j](1-w) missing" if tol(df[t], -(v0 * (1 - w))) else ("v[j+1]w missing" if tol(df[t], -(v1 * w)) else "other")
        tot[(jj, kind)] = tot.get((jj, kind), 0) + 1
print("failing pixels by (tap index j within its level, what is missing):")
for k in sorted(tot):
    print("  ", k, tot[k])
