#!/usr/bin/env python3
"""Host (enqueue) time vs GPU time of one forward at cfg2: how far the CPU runs ahead of the device."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
import bench
H, W, V, cascade = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "dtu_1600x1184_v10_it32"]
dev = torch.device("cuda")
model = RAFT(cascade=cascade, test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
inp = (images.to(dev), poses.to(dev), intr.to(dev))
with torch.no_grad():
    for _ in range(3):
        model(*inp, scale=scale)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        out = model(*inp, scale=scale)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
