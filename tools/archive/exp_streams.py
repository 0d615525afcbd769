#!/usr/bin/env python3
"""Throughput with S independent depth maps in flight on S HIP streams (one model copy per stream) against one at a time."""
import copy, os, sys, torch
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
n = 24
for S in (4, 6, 8, 4, 6, 8):
    models = [model] + [copy.deepcopy(model) for _ in range(S - 1)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [None] * S
    with torch.no_grad():
        for i in range(2 * S):
            with torch.cuda.stream(streams[i % S]):
                outs[i % S] = models[i % S](*x, scale=scale)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s_ in streams:
            s_.wait_event(e0)
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                outs[i % S] = models[i % S](*x, scale=scale)
        for s_ in streams:
            torch.cuda.current_stream().wait_stream(s_)
        e1.record()
        torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{S} stream(s): {e0.elapsed_time(e1) / n:.3f} ms per depth map ({1e3 * n / e0.elapsed_time(e1):.1f} maps/s), outputs identical: {same}", flush=True)
