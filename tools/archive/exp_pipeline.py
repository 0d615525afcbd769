#!/usr/bin/env python3
"""A/B of RAFT.PIPELINE_BUILD (stage-0 cost volume built on a second stream under the encoders) at the bench workload."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cer_mvs_amd import RAFT
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

H, W, V, cascade = bench.WORKLOADS["dtu_1600x1184_v10_it32"]
dev = torch.device("cuda")
model = RAFT(cascade=cascade, test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
x = (images.to(dev), poses.to(dev), intr.to(dev))


def run(label, n=10):
    with torch.no_grad():
        for _ in range(2):
            model(*x, scale=scale)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            model(*x, scale=scale)
        e1.record()
        torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1) / n:.3f} ms / depth map", flush=True)


default_batches = RAFT._batches
for rep in range(2):
    RAFT.PIPELINE_BUILD = False
    run("no pipeline")
    RAFT.PIPELINE_BUILD = True
    for sizes in ([4, 3, 3], [5, 5], [7, 3], [10], [2, 2, 2, 2, 2]):
        RAFT._batches = staticmethod(lambda V, s=sizes: list(s))
        run(f"pipeline, batches {sizes}")
    RAFT._batches = default_batches
