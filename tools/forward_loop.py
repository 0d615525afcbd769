#!/usr/bin/env python3
"""N identical test-mode forwards of a bench workload and nothing else - the profiling subject of tools/archive/prof_r05.sh.

bench.py's process also runs a calibration forward, an instrumented forward and extra encode() calls; under rocprofv3 those make "per forward"
columns wrong (VERDICT r4 "weak" 7).  Here every forward in the process is the same forward (gru_precision pinned, no calibration): kernel time
per forward = window total / forwards, exactly.  usage: forward_loop.py [--workload W] [--streams S] [--forwards N] [--gru-precision P]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                               # noqa: E402  (the workload table)
from cer_mvs_amd import RAFT                                               # noqa: E402
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dtu_1600x1184_v10_it32", choices=sorted(bench.WORKLOADS))
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--forwards", type=int, default=12)
ap.add_argument("--gru-precision", default="s16f8")
ap.add_argument("--enc-precision", default="f6", help="the shipped auto form on the bench weights is s16f8+e6: encoders in the FP6-correction form")
args = ap.parse_args()
H, W, V, cascade = bench.WORKLOADS[args.workload]
dev = torch.device("cuda")
model = RAFT(cascade=cascade, test_mode=True, gru_precision=args.gru_precision, enc_precision=args.enc_precision, cost_precision="x2")   # (the shipped auto form on the bench weights: s16f8+e6+c2)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
model = model.to(dev).eval()
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
inputs = (images.to(dev), poses.to(dev), intr.to(dev))
with torch.no_grad():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.streams == 1:
        for _ in range(args.forwards):
            out = model(*inputs, scale=scale)
    else:
        from cer_mvs_amd.pipeline import DepthMapPipeline
        pipe = DepthMapPipeline(model, streams=args.streams)
        hs = [pipe.submit(*inputs, scale) for _ in range(args.forwards)]
        out = pipe.result(hs[-1], wait_on_host=False)
    torch.cuda.synchronize()
print(f"{args.forwards} forwards, {args.streams} in flight: {1e3 * (time.perf_counter() - t0) / args.forwards:.3f} ms per forward (first ones cold)")
