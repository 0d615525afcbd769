#!/usr/bin/env python3
"""The lookup kernel alone, N launches on fixed inputs, every output compared with the first one on the device; mismatches are
characterised (how many elements, which 64-pixel tiles, which 16-channel groups, which launch).  Run several copies concurrently to
share the GPU between processes:  for i in 1 2 3; do CER_MVS_LIB=... python tools/archive/repro_lookup_kernel.py 3000 & done; wait"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
burst = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda")
h, w, D, L_, r = 296, 400, 64, 3, 5
P = h * w
rs = 112
g = torch.Generator(device="cpu").manual_seed(7)
vol = (torch.rand(P, rs, generator=g) * 16 - 8).to(dev)
origin = torch.full((P,), 0.00125).to(dev)
incre = 0.0025 / 64
disp = (torch.rand(P, generator=g) * 60 * incre).to(dev)
wt = (torch.rand(33, 64, generator=g) - 0.5).to(dev)
b = (torch.rand(64, generator=g) - 0.5).to(dev)
outs = [torch.zeros(ops.s16_pixels(h, w) if split == 2 else P, 64, device=dev) for _ in range(burst)]
first = ops.lookup_encode(vol, origin, disp, wt, b, D, incre, L_, r, out_split=split, log2s=4, img_w=w).clone()
torch.cuda.synchronize()
bad, shown = 0, 0
for it in range(0, n, burst):
    for o in outs:
        ops.lookup_encode(vol, origin, disp, wt, b, D, incre, L_, r, out=o, out_split=split, log2s=4, img_w=w)
    torch.cuda.synchronize()
    for j, o in enumerate(outs):
        if not torch.equal(o, first):
            bad += 1
            if shown < 6 and split == 0:
                shown += 1
                d = (o != first)
                px = d.any(1).nonzero().flatten()
                ch = d.any(0).nonzero().flatten()
                tiles = torch.unique(px // 64)
                grp = torch.unique(ch // 16)
                inpix = torch.unique(px % 64)
                mx = float((o - first).abs().max())
                print(f"launch {it + j}: {int(d.sum())} elements differ (max |diff| {mx:.3e}); pixels {px.numel()} in tiles {tiles.tolist()[:12]}{'...' if tiles.numel() > 12 else ''} "
                      f"(slots in tile: {inpix.tolist()[:16]}{'...' if inpix.numel() > 16 else ''}); channel groups {grp.tolist()}; channels {ch.tolist()[:20]}", flush=True)
print(f"pid {os.getpid()}: {bad} of {n} launches differ from the first")
