#!/usr/bin/env python3
"""Two depth maps in flight: does it matter whether the two forwards run in phase (both in the encoders, then both in the GRU loop) or
staggered by half a forward (one in its HBM-bound encoders while the other is in its MFMA-bound loop)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cer_mvs_amd import RAFT
from cer_mvs_amd.pipeline import DepthMapPipeline
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
from cer_mvs_amd.update import UpdateBlock

H, W, V, cascade = bench.WORKLOADS["dtu_1600x1184_v10_it32"]
dev = torch.device("cuda")
model = RAFT(cascade=cascade, test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
x = (images.to(dev), poses.to(dev), intr.to(dev))
mid = {}
real_run = UpdateBlock.run
def run_hook(self, iters, vol, origin, net_l, disp, hoisted, stage, *a, **k):
    if stage == 1 and "want" in mid:                       # start of the second stage's loop: ~60 % into a forward
        ev = torch.cuda.Event(); ev.record(); mid["ev"] = ev; mid.pop("want")
    return real_run(self, iters, vol, origin, net_l, disp, hoisted, stage, *a, **k)
UpdateBlock.run = run_hook


def measure(stagger, n=16):
    pipe = DepthMapPipeline(model, streams=2)
    with torch.no_grad():
        for _ in range(4):
            pipe.result(pipe.submit(*x, scale), wait_on_host=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in pipe.streams:
            st.wait_event(e0)
        hs = []
        for i in range(n):
            if stagger and i == 0:
                mid["want"] = True
            hs.append(pipe.submit(*x, scale))
            if stagger and i == 0:
                pipe.streams[1].wait_event(mid["ev"])      # the second stream starts when the first is ~60 % through its forward
        for st in pipe.streams:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
    print(f"stagger={stagger}: {e0.elapsed_time(e1) / n:.3f} ms per depth map", flush=True)


for s_ in (False, True, False, True):
    measure(s_)
