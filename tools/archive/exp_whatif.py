#!/usr/bin/env python3
"""What-if under two depth maps in flight: throughput with one component of the forward made free (cached result), i.e. the most an
optimisation of that component could buy.  Components: the cost volume (both stages), the encoders."""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cer_mvs_amd import RAFT, ops
from cer_mvs_amd.pipeline import DepthMapPipeline
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

H, W, V, cascade = bench.WORKLOADS["dtu_1600x1184_v10_it32"]
dev = torch.device("cuda")
model = RAFT(cascade=cascade, test_mode=True)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
x = (images.to(dev), poses.to(dev), intr.to(dev))


def run(label, S=2, n=12):
    pipe = DepthMapPipeline(model, streams=S)
    with torch.no_grad():
        for _ in range(2 * S):
            pipe.result(pipe.submit(*x, scale), wait_on_host=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in pipe.streams:
            st.wait_event(e0)
        hs = [pipe.submit(*x, scale) for _ in range(n)]
        for st in pipe.streams:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
    print(f"{label:34s} S={S}: {e0.elapsed_time(e1) / n:.3f} ms per depth map", flush=True)


run("baseline", 1); run("baseline", 2)
real_build = ops.cost_build
cache = {}
def fake_build(f1, f2, Pij, disp, D, *a, **k):
    key = (D, torch.cuda.current_stream().cuda_stream)
    if key not in cache:
        cache[key] = real_build(f1, f2, Pij, disp, D, *a, **k)
    return cache[key]
ops.cost_build = fake_build
real_split = ops.feat_split
scache = {}
def fake_split(t, out=None):
    key = (tuple(t.shape), torch.cuda.current_stream().cuda_stream)
    if key not in scache:
        scache[key] = real_split(t)
    return scache[key]
ops.feat_split = fake_split
run("cost volume + split free", 1); run("cost volume + split free", 2)
ops.cost_build, ops.feat_split = real_build, real_split
real_encode = RAFT.encode
ecache = {}
def fake_encode(self, images, views, raw=False, parts="all"):
    key = (id(self), torch.cuda.current_stream().cuda_stream, parts)
    if key not in ecache:
        ecache[key] = real_encode(self, images, views, raw=raw, parts=parts)
    net, inp, f1, f2 = ecache[key]
    return net.clone(), inp, f1, f2
RAFT.encode = fake_encode
run("encoders free", 1); run("encoders free", 2)

RAFT.encode = real_encode
# ---- inner-loop kernels made free one class at a time (raw C-ABI level: replayed launch plans call the library directly)
from cer_mvs_amd import _lib as L
import ctypes
lib = L.load()
def free_symbol(name):
    real = getattr(lib, name)
    class Fake:
        def __call__(self, *a):
            return 0
    setattr(lib, name, Fake())
    return real
for names, label in ((("cer_lookup_encode_f32",), "lookup free"), (("cer_delta_sum_f32",), "delta_sum free")):
    reals = {n_: free_symbol(n_) for n_ in names}
    run(label, 1); run(label, 2)
    for n_, r_ in reals.items():
        setattr(lib, n_, r_)
