#!/usr/bin/env python3
"""Geometric-consistency fusion at BASELINE configs[1] size (1600x1184 depth maps, 10 source views per reference view):
fused vote kernel (HIP events), the whole ten-round loop over 11 views, HBM roofline fraction, and the CPU oracle
(oracle/fusion_oracle.py = the reference's op sequence) timed on the host cores for one vote.  One JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cer_mvs_amd import fusion                                            # noqa: E402
from cer_mvs_amd.synthetic import synthetic_depth_maps, synthetic_scene   # noqa: E402


def main():
    H, W, V = 1184, 1600, 10
    dev = torch.device("cuda")
    _, poses, intr, _ = synthetic_scene(32, 32, V, seed=1)
    K, E = intr[0].clone(), poses[0]
    K[:, 0, 0] = K[:, 1, 1] = 1.8 * W
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    depths_cpu = synthetic_depth_maps(H, W, V, seed=1)
    depths = depths_cpu.to(dev)
    N = V + 1
    src = list(range(1, N))
    cams = fusion.compose_cams(K[0], E[0], K[src], E[src]).to(dev)
    dsrc = depths[src].contiguous()
    cnt = torch.zeros(fusion.COUNTERS, device=dev, dtype=torch.int32)
    f = lambda: fusion.vote(depths[0], K[0], E[0], dsrc, K[src], E[src], 4.0, 1300.0, cams=cams, count=cnt)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    P = H * W
    # compulsory traffic of one vote: reference depth (4 B) + every source depth map once (4 B x S; the 2x2 footprints of
    # neighbouring pixels overlap, so a source map is read about once) + mask (1 B) + averaged depth (4 B) per pixel
    alg_bytes = P * (4 + 4 * V + 1 + 4)
    pairs = [(i, [j for j in range(N) if j != i]) for i in range(N)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    masks, est, thre, hist = fusion.fuse_depth_maps(depths, K, E, pairs, glb=0.25)
    torch.cuda.synchronize(); loop_s = time.perf_counter() - t0
    # CPU oracle: one vote, the host's torch threads as they come
    from oracle import fusion_oracle as FO
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t0 = time.perf_counter()
    FO.vote(depths_cpu[0], K[0], E[0], depths_cpu[src], K[src], E[src], 4.0, 1300.0)
    cpu_s = time.perf_counter() - t0
    print(json.dumps({
        "metric": "geometric-consistency votes/s (1 reference depth map vs 10 source views, 1600x1184)", "value": 1e6 / us, "unit": "votes/s",
        "vote_us": us, "ten_round_loop_11_views_s": loop_s, "final_mean_mask_area": hist[-1][1],
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg_bytes / (us * 1e-6) / 1e9 / 8000.0, "algorithmic_bytes": alg_bytes},
        "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "votes/s", "cores": torch.get_num_threads(), "kind": "port",
                         "sample": "one vote (1 reference x 10 source views) through oracle/fusion_oracle.py", "seconds": cpu_s}}))


if __name__ == "__main__":
    main()
