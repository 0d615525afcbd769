#!/usr/bin/env python3
"""S depth maps in flight (S = 1..6) on one box, interleaved twice, plus the host-side enqueue time of one forward."""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from importlib import import_module
pkg = import_module("cer-mvs_amd")
RAFT = pkg.RAFT
syn = import_module("cer-mvs_amd.synthetic")

H, W, V, cascade = bench.WORKLOADS[bench.DEFAULT_WORKLOAD] if hasattr(bench, "DEFAULT_WORKLOAD") else bench.WORKLOADS["dtu_1600x1184_v10_it32"]
dev = torch.device("cuda")
model = RAFT(cascade=cascade, test_mode=True)
model.load_state_dict(syn.fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = syn.synthetic_scene(H, W, V, seed=0)
x = (images.to(dev), poses.to(dev), intr.to(dev))
with torch.no_grad():
    for _ in range(3):
        model(*x, scale=scale)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(*x, scale=scale)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    print("host enqueue ms / total ms per forward:", [f"{a:.2f}/{b:.2f}" for a, b in ts], flush=True)
n = 24
models = [model] + [copy.deepcopy(model) for _ in range(5)]
allstreams = [torch.cuda.Stream() for _ in range(6)]
for S in (1, 2, 3, 4, 6, 1, 2, 3, 4, 6):
    streams = allstreams[:S]
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
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                outs[i % S] = models[i % S](*x, scale=scale)
        t1 = time.perf_counter()
        for s_ in streams:
            torch.cuda.current_stream().wait_stream(s_)
        e1.record()
        torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{S} stream(s): {e0.elapsed_time(e1) / n:.3f} ms per depth map ({1e3 * n / e0.elapsed_time(e1):.1f} maps/s), host enqueue {1e3 * (t1 - t0) / n:.2f} ms per map, identical: {same}", flush=True)
