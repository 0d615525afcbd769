#!/usr/bin/env python3
"""bench.py - depth-maps/s of the CER-MVS depth-inference hot path on MI355X.

A "step" is one test-mode RAFT.forward (one depth map: 1 reference + V source views already resident in
HBM -> disparity in HBM) at BASELINE.json configs[1]: DTU 1600x1184, 10 source views, 32 GRU iterations
(cascade (64,64,16),(-1,320,16)), synthetic images + closed-form weights, fp32 end to end.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1 (configs[3]): ONE reference frame is sharded over the ranks (`--mode shard`, default): rank g encodes the
source views v with (v-1) % N == g and the feature maps are all-gathered (RCCL over xGMI); image rows are sharded for
the cost volume and the GRU loop with a 7-row halo all-gather of (net, disp) per iteration (cer-mvs_amd/slab.py).
Total work is fixed as N grows => "scaling": "strong".  `--mode views` is the simpler scheme (views sharded, view-sum
volume all-reduced once per stage, GRU replicated); `--mode replica` runs N independent depth maps (weak scaling).

Rank 0 prints ONE JSON line (see the README of this file's contract in DESIGN.md §Measurement): besides
the driver's keys it carries
  "roofline":     the dominant kernel (the z|r gate convolution, MFMA-bound) - algorithmic FLOPs per launch
                  divided by its average launch duration measured with HIP events on its stream,
  "cpu_baseline": the oracle (a CPU port of the reference's torch op sequence) timed on this box's host
                  cores on a bounded sample of the same workload, extrapolated to depth-maps/s.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {
    # name: (H, W, V, cascade)
    "dtu_1600x1184_v10_it32": (1184, 1600, 10, [(64, 64, 16), (-1, 320, 16)]),
    "dtu_640x480_v2_it4": (480, 640, 2, [(64, 64, 2), (-1, 320, 2)]),
    "blended_2048x1536_v7_it16": (1536, 2048, 7, [(64, 64, 8), (-1, 320, 8)]),
    "tnt_3840x2160_v15_it16": (2160, 3840, 15, [(64, 64, 8), (-1, 320, 8)]),
}
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dtu_1600x1184_v10_it32", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="shard", choices=["shard", "views", "replica"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1: nccl (= RCCL over xGMI, the product path) or gloo (validation "
                         "of the multi-rank code path on a box with fewer GPUs than ranks: ranks share devices)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "amp"])
    ap.add_argument("--encoder", default="hip", choices=["hip", "miopen"], help="encoder backend: channels-last HIP engine or PyTorch-ROCm (MIOpen)")
    ap.add_argument("--gru-precision", default="s16", choices=["s16", "f16x3", "fp32"],
                    help="arithmetic of the update block's 3x3 convs: split-f16 MFMA with fp32-class accuracy (s16: round-2 kernels, one "
                         "accumulator; f16x3: round-1 kernels), or exact fp32 MFMA")
    return ap.parse_args()


def kernel_timing(model, inputs, scale):
    """One extra, instrumented forward: HIP events (on the launch stream = torch's current stream) around every
    kernel class of the inner loop.  Returns {name: (launches, total_ms)}."""
    from cer_mvs_amd import ops, update
    records = {}
    pending = []
    originals = {}
    plans = update.USE_PLANS
    update.USE_PLANS = False             # every iteration through the (wrapped) ops, so that each launch gets its events

    def wrap(name, fn, label=None):
        def inner(*a, **k):
            key = label(*a, **k) if label else name
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            pending.append((key, e0, e1))
            return out
        return inner

    def conv_label(pc, srcs, h, w, epi, **k):
        return {0: "conv3x3_linear", 1: f"conv3x3_relu_{pc.cout}", 2: "conv3x3_gates_zr", 3: "conv3x3_gru_q", 4: "conv3x3_delta_fused"}[epi]

    for name in ("cost_build", "pyramid", "lookup_encode", "delta_tail", "delta_sum"):
        originals[name] = getattr(ops, name)
        setattr(ops, name, wrap(name, originals[name]))
    originals["conv3x3"] = ops.conv3x3
    ops.conv3x3 = wrap("conv3x3", originals["conv3x3"], conv_label)
    originals["conv3x3_s16"] = ops.conv3x3_s16
    ops.conv3x3_s16 = wrap("conv3x3_s16", originals["conv3x3_s16"], conv_label)
    try:
        with torch.no_grad():
            model(*inputs, scale=scale)
        torch.cuda.synchronize()
    finally:
        update.USE_PLANS = plans
        for name, fn in originals.items():
            setattr(ops, name, fn)
    per = {}
    for key, e0, e1 in pending:
        per.setdefault(key, []).append(e0.elapsed_time(e1))
    # per class: launches and launches x MEDIAN launch time - a single stalled launch in this one instrumented pass
    # (observed: one 20 ms outlier among 32 launches) must not move the reported per-launch duration; classes with two
    # launches of different shapes (cost_build: the two cascade stages) keep their plain sum
    for key, ts in per.items():
        ts_sorted = sorted(ts)
        total = sum(ts) if len(ts) < 4 else ts_sorted[len(ts) // 2] * len(ts)
        records[key] = (len(ts), total)
    return records


def cpu_baseline(H, W, V, cascade, sd):
    """Oracle (CPU port of the reference's torch op sequence) on a bounded sample, extrapolated to one depth map."""
    from oracle import cer_oracle as O
    from cer_mvs_amd.synthetic import synthetic_scene
    cores = os.cpu_count() or 1
    # torch's CPU kernels stop scaling (and the 528 small grid_samples per lookup get much slower) with hundreds of
    # threads: calibrate the thread count on one conv + one grid_sample and keep the fastest.
    import torch.nn.functional as F
    cand = sorted({c for c in (8, 16, 32, 64, cores) if c <= cores})
    xcal = torch.randn(1, 64, H // 4, W // 4)
    wcal = torch.randn(64, 64, 3, 3)
    gcal = torch.rand(H // 4 * W // 4 // 16, 1, 1, 2) * 2 - 1
    rcal = torch.randn(H // 4 * W // 4 // 16, 1, 1, 64)
    best, best_t = cand[0], float("inf")
    for c in cand:
        torch.set_num_threads(c)
        F.conv2d(xcal, wcal, padding=1)
        t0 = time.perf_counter()
        F.conv2d(xcal, wcal, padding=1)
        for _ in range(16):
            F.grid_sample(rcal, gcal, align_corners=True)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = c, dt
    threads = best
    torch.set_num_threads(threads)
    h, w = H // 4, W // 4
    P = h * w
    images, poses, intr, _ = synthetic_scene(H, W, 1, seed=0)
    imgs = images[0].float() * (2 / 255.0) - 1
    intr_f = intr[0].clone()
    intr_f[:, :2] /= 4
    stages = O.resolve_cascade(cascade)
    t = {}
    with torch.no_grad():
        O.encoder(imgs[0:1, :, :64, :64], sd, "fnet.", "instance")            # warm the thread pool
        t0 = time.perf_counter()
        fm = torch.cat([O.encoder(imgs[i:i + 1], sd, "fnet.", "instance") for i in range(2)], 0)
        t["enc_per_image"] = (time.perf_counter() - t0) / 2
        disp = torch.full((h, w), 0.0015)
        t["build_per_view"] = []
        for s, (D, incre, T) in enumerate(stages):
            t0 = time.perf_counter()
            vol, origin = O.cost_volume(fm, poses[0], intr_f, D, incre, disp, shift=(s == 0))
            levels = O.pyramid(vol, 3)
            t["build_per_view"].append(time.perf_counter() - t0)
        # one lookup (all V views: cost is value-independent, so tile the 1-view pyramid) + one update iteration
        D, incre, _ = stages[-1]
        lv = [l.expand(V, -1, -1).contiguous() for l in levels]
        net = torch.zeros(1, 64, h, w)
        inp = torch.zeros(1, 64, h, w)
        d4 = disp.view(1, 1, h, w)
        t0 = time.perf_counter()
        feats = O.lookup(lv, origin, disp, D, incre, 5)
        t["lookup_per_iter"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.update_block(sd, net, inp, d4, feats, 1)
        t["update_per_iter"] = time.perf_counter() - t0
    iters = sum(T for _, _, T in stages)
    total = (V + 2) * t["enc_per_image"] + V * sum(t["build_per_view"]) + iters * (t["lookup_per_iter"] + t["update_per_iter"])
    sample_s = 2 * t["enc_per_image"] + sum(t["build_per_view"]) + t["lookup_per_iter"] + t["update_per_iter"]
    return {
        "value": 1.0 / total, "unit": "depth-maps/s", "cores": threads, "host_cores": cores, "kind": "port",
        "sample": (f"oracle/cer_oracle.py at {W}x{H}: 2 fnet passes, 1-view cost volume + pyramid for both stages, "
                   f"1 lookup over {V} views + 1 update-block iteration ({sample_s:.1f} s measured); extrapolated to "
                   f"{V + 2} encoder passes, {V} views x 2 stages, {iters} iterations = {total:.1f} s per depth map"),
        "seconds_per_depth_map": total,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({ndev} visible); RCCL needs one GPU per rank")
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        group = dist.group.WORLD

    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

    H, W, V, cascade = WORKLOADS[args.workload]
    shard = world > 1 and args.mode in ("shard", "views")
    model = RAFT(cascade=cascade, test_mode=True, precision=args.precision, view_group=group if shard else None,
                 gru_precision=args.gru_precision, encoder_backend=args.encoder, shard="slab" if args.mode == "shard" else "views")
    sd = fill_state_dict(model.state_dict(), seed=5)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    seed = 0 if (shard or world == 1) else rank          # replicas: a different reference frame per rank
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=seed)
    inputs = (images.to(dev), poses.to(dev), intr.to(dev))

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = model(*inputs, scale=scale)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model(*inputs, scale=scale)
        sync()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    maps = args.steps * (world if (world > 1 and not shard) else 1)

    result = None
    rec = None
    if rank == 0 or shard:               # a sharded forward is collective: every rank runs the instrumented pass
        rec = kernel_timing(model, inputs, scale)
    if rank == 0:
        P = (H // 4) * (W // 4)
        if shard and args.mode == "shard":
            # row-slab sharding: rank 0's kernels process its owned rows plus the halo, not the whole image - the per-launch
            # roofline figures below are per rank and must use that pixel count
            from cer_mvs_amd import slab as _slab
            if _slab.can_shard(H // 4, world):
                _, _, e0, e1 = _slab.slab_bounds(H // 4, world, 0)
                P = (e1 - e0) * (W // 4)
        kern = {k: {"launches": n, "avg_us": 1e3 * t / n, "total_ms": t} for k, (n, t) in sorted(rec.items())}
        n_zr, t_zr = rec["conv3x3_gates_zr"]
        flops_zr = 2.0 * 9 * (64 + 49 + 64) * 128 * P            # algorithmic (unpadded K = net|disp49|corr), DESIGN.md
        achieved = flops_zr / (t_zr / n_zr * 1e-3) / 1e12
        if args.gru_precision in ("s16", "f16x3"):
            # every fp32 product costs 3 f16 MFMA products -> ceiling = f16 dense peak / 3 in fp32-equivalent flops
            peak, kname = F16_MFMA_PEAK_TFLOPS / 3, "conv3x3_f16x3_kernel<4,2,1,2,3,4,GATES> (z|r gates, 3x3, K=177, N=128; 3 f16 MFMAs per fp32 product)"
        else:
            peak, kname = FP32_MFMA_PEAK_TFLOPS, "conv3x3_kernel<2,2,4,4,GATES> (z|r gates, 3x3, K=177, N=128; exact fp32 MFMA)"
        traffic = None                                            # HBM bytes per launch from the committed PMC passes
        try:
            with open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = json.load(f)["conv3x3_gates_zr"]["traffic_bytes"] if args.gru_precision in ("s16", "f16x3") else None
        except Exception:
            traffic = None
        roofline = {"kernel": kname, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic, "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2x FETCH correction)", "avg_launch_us": 1e3 * t_zr / n_zr, "launches": n_zr, "flops_per_launch": flops_zr,
                    "peak_note": ("fp32-equivalent ceiling = 2500 TF dense f16 MFMA / 3" if args.gru_precision in ("s16", "f16x3")
                                  else "fp32 MFMA dense peak"),
                    "frac_of_raw_f16_peak": (3 * achieved / F16_MFMA_PEAK_TFLOPS) if args.gru_precision in ("s16", "f16x3") else None}
        hbm = None
        if "lookup_encode" in rec:
            n_lk, t_lk = rec["lookup_encode"]
            D0 = cascade[0][0]
            rowf = (D0 + D0 // 2 + D0 // 4 + 3) // 4 * 4
            bytes_lk = 4.0 * P * (rowf + 2 + 64)                  # stage-0 row + origin + disp read, 64 floats written
            gbs = bytes_lk / (t_lk / n_lk * 1e-3) / 1e9
            hbm = {"kernel": "lookup_encode_kernel (multi-level lookup + view mean + 1x1 conv)", "bound": "hbm", "achieved": gbs,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "avg_launch_us": 1e3 * t_lk / n_lk,
                   "bytes_per_launch": bytes_lk, "note": "stage-0 row length used for all launches (stage-1 rows are 80 floats)"}
        result = {
            "metric": "depth-maps/sec (ref+N src views) at DTU 1600x1184; HBM GB/s vs roofline",
            "value": maps / elapsed, "unit": "depth-maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if (shard or world == 1) else "weak", "vs_baseline": None,
            "dtype": ("f32" if args.precision == "fp32" else "f32 (encoders f16 autocast)")
                     + (" [GRU convs: f32 operands split into 2 x f16, 3 MFMAs per product, f32 accumulate - fp32-equivalent]"
                        if args.gru_precision == "f16x3" else ""), "data": "synthetic",
            "config": {"workload": args.workload, "image": f"{W}x{H}", "src_views": V, "cascade": cascade,
                       "gru_iters": sum(c[2] for c in cascade),
                       "parallelism": "single" if world == 1 else (
                           (f"row-slab x{world}: feature all-gather + 7-row halo all-gather per GRU iteration" if args.mode == "shard"
                            else f"view-shard x{world} + all-reduce/stage") if shard else f"replica x{world}"),
                       **({"backend": "gloo (validation run, ranks may share a GPU)"} if (world > 1 and args.backend == "gloo") else {})},
            "roofline": roofline, "roofline_hbm_kernel": hbm, "kernels": kern,
            "peak_device_memory_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
        }
        # parity of THIS run's output: the default workload with the default seeds is exactly the configuration the reference
        # was captured on (tests/golden/e2e_cfg2.npz, tools/gen_golden.py --only e2e_cfg2); committed fixture, no reference needed
        gpath = os.path.join(REPO, "tests", "golden", "e2e_cfg2.npz")
        if args.workload == "dtu_1600x1184_v10_it32" and (world == 1 or shard) and os.path.exists(gpath):
            import numpy as np
            ref = torch.from_numpy(np.load(gpath)["disp"]).double()
            got = out.detach().cpu().double()
            if got.shape == ref.shape:
                result["parity"] = {"rel_l1_disparity_vs_reference_capture": float((got - ref).abs().sum() / ref.abs().sum()),
                                    "tolerance": 1e-4, "fixture": "tests/golden/e2e_cfg2.npz"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(H, W, V, cascade, {k: v.cpu() for k, v in sd.items()})
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
