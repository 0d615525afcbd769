#!/usr/bin/env python3
"""bench.py - depth-maps/s of the CER-MVS depth-inference hot path on MI355X.

A "step" is one test-mode RAFT.forward (one depth map: 1 reference + V source views already resident in
HBM -> disparity in HBM) at BASELINE.json configs[1]: DTU 1600x1184, 10 source views, 32 GRU iterations
(cascade (64,64,16),(-1,320,16)), synthetic images + closed-form weights.  Tensors are fp32 end to end; the dense convolutions
split their fp32 operands into f16 halves on the matrix cores (the JSON line's `dtype` and `gru_precision` say exactly how).
With one GPU the timed steps are submitted to pipeline.DepthMapPipeline (`--streams`, default 3 depth maps in flight: the product's
inference() default); `one_at_a_time` in the JSON line is the same forward with a single depth map in flight.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1 (configs[3]): ONE reference frame is sharded over the ranks.  The default `--mode both` times the two schemes one
after the other and reports both under "modes" (plus, as context, `replica`: N independent depth maps, weak scaling); the headline
(`value`, `ms_per_step`) is a fresh timed run of the faster of the two strong-scaling schemes:
  shard   rank g encodes the source views v with (v-1) % N == g; their feature maps are all-gathered (RCCL over xGMI) while
          the rank encodes the reference view; image rows are sharded for the cost volume and the GRU loop, with the 7-row
          halo of (net, disp) exchanged point-to-point with the two neighbours per iteration (cer-mvs_amd/slab.py);
  views   north_star's scheme: views sharded, the view-sum volume all-reduced once per stage, GRU replicated.
Total work is fixed as N grows => "scaling": "strong".  `--mode replica` runs N independent depth maps (weak scaling).

Rank 0 prints ONE JSON line (see the README of this file's contract in DESIGN.md §Measurement): besides
the driver's keys it carries
  "roofline":     the dominant kernel (the z|r gate convolution, MFMA-bound) - algorithmic FLOPs per launch
                  divided by its average launch duration measured with HIP events on its stream,
  "cpu_baseline": the oracle (a CPU port of the reference's torch op sequence) timed on this box's host
                  cores on ONE WHOLE depth map of the same workload (median of 3 runs, thread count chosen by a small timed forward).
N > 1 lines also carry "n1_one_at_a_time": rank 0's own single-GPU forward (one depth map at a time, unsharded), so that the
strong-scaling ratio is taken against the regime the sharded modes run in (they keep one depth map in flight).
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
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dtu_1600x1184_v10_it32", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="both", choices=["both", "shard", "views", "replica"],
                    help="N > 1: both = time the view-shard scheme (north_star's) AND the row-slab scheme, headline = row-slab")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent depth maps kept in flight per GPU (cer-mvs_amd/pipeline.py: the product's inference() default is 3); "
                         "1 = one at a time.  Sharded N > 1 modes always run one depth map at a time")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1: nccl (= RCCL over xGMI, the product path) or gloo (validation "
                         "of the multi-rank code path on a box with fewer GPUs than ranks: ranks share devices)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "amp"])
    ap.add_argument("--encoder", default="hip", choices=["hip", "miopen"], help="encoder backend: channels-last HIP engine or PyTorch-ROCm (MIOpen)")
    ap.add_argument("--enc-precision", default="auto", choices=["auto", "f16x3", "f6"],
                    help="arithmetic of the encoders' convolutions: three f16 MFMA terms, or the correction terms in FP6 (RAFT enc_precision)")
    ap.add_argument("--cost-precision", default="auto", choices=["auto", "x3", "x2"],
                    help="dots of the cost volume: three split-f16 terms, or without the source texels' lo planes (RAFT cost_precision)")
    ap.add_argument("--gru-precision", default="auto", choices=["auto", "s16f6", "s16f8", "s16", "f16x3", "fp32"],
                    help="arithmetic of the update block's 3x3 convs: split-f16 MFMA (s16f8: correction terms on the fp8 matrix instruction, "
                         "4e-6 from fp32; s16: all-f16, fp32-class, one accumulator; f16x3: round-1 kernels), or exact fp32 MFMA.  auto (the "
                         "product's default): s16f8 if the model's first forward agrees with s16 within 2.5e-5 relative L1, else s16 - the "
                         "calibration forward runs in the warm-up; the JSON line says which form the timed region used")
    return ap.parse_args()


def kernel_timing(model, inputs, scale):
    """One extra, instrumented forward: HIP events (on the launch stream = torch's current stream) around EVERY launch the
    library makes - encoders, cost volume, lookups, update-block convs (cer-mvs_amd/_lib.timing).  Launch plans are off for
    this pass, so every iteration goes through the Python wrappers: it is slower than the timed region (stated in the JSON).
    Returns ({class: (launches, total_ms)}, wall_ms of the pass)."""
    from cer_mvs_amd import _lib as L, update
    plans = update.USE_PLANS
    update.USE_PLANS = False
    recs = []
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad(), L.timing(recs):
            model(*inputs, scale=scale)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
    finally:
        update.USE_PLANS = plans
    epi_names = {0: "linear", 1: "relu", 2: "gates_zr", 3: "gru_q", 4: "delta_fused"}
    per = {}
    nbuild = 0
    for name, args, e0, e1 in recs:
        key = name[4:] if name.startswith("cer_") else name
        if name == "cer_conv3x3_s16":
            key = "conv3x3_" + epi_names[args[15] & 0xFF] + (f"_{args[14]}" if (args[15] & 0xFF) == 1 else "")
        elif name == "cer_conv3x3_f16x3":
            key = "conv3x3_" + epi_names[args[12] & 0xFF] + (f"_{args[11]}" if (args[12] & 0xFF) == 1 else "")
        elif name == "cer_conv3x3_f32":
            key = "conv3x3_" + epi_names[args[11] & 0xFF] + (f"_{args[10]}" if (args[11] & 0xFF) == 1 else "")
        elif name in ("cer_cost_build_f32", "cer_cost_lines_f32"):      # (lines: setup + tile kernel + view reduction = one call)
            key = f"cost_build_stage{nbuild}"
            nbuild += 1
        elif name == "cer_enc_conv_f16x3":
            key = f"enc_conv_{args[11]}to{args[12]}_taps{args[13]}_s{args[14]}"
        per.setdefault(key, []).append(e0.elapsed_time(e1))
    out = {}
    # per class: launches and launches x MEDIAN launch time (a single stalled launch in this one pass must not move the reported
    # per-launch duration); classes with < 4 launches keep their plain sum.  The REAL sum of the class's event times is kept beside it
    # (``kernel_timing.sums``): a kernel SUM must not be built from medians (21 encoder launches of 28-613 us are not 21 x 63 us).
    sums = {}
    for key, ts in per.items():
        ts_sorted = sorted(ts)
        out[key] = (len(ts), sum(ts) if len(ts) < 4 else ts_sorted[len(ts) // 2] * len(ts))
        sums[key] = sum(ts)
    kernel_timing.sums = sums
    return out, wall


def _cpu_baseline_child(H, W, V, cascade, sd, budget_s, q):
    try:
        q.put(cpu_baseline(H, W, V, cascade, sd, budget_s))
    except Exception as e:                                   # (reported, not raised: the GPU line must not depend on the host side)
        q.put({"value": None, "unit": "depth-maps/s", "kind": "port", "sample": f"the CPU baseline failed: {type(e).__name__}: {e}"})


def cpu_baseline_guarded(H, W, V, cascade, sd):
    """``cpu_baseline`` in a child process under a wall-clock limit (CER_BENCH_CPU_BUDGET seconds of CPU work, default 300, + start-up): the boxes of the
    pool differ in the cores a process may really use, and a baseline that oversubscribes a small quota has taken 50 minutes before (round 6) - the GPU
    line of this script must not hang on the host side.  The child also adapts to its budget (fewer calibration candidates, fewer timed runs)."""
    import multiprocessing as mp
    import queue as _queue
    budget = float(os.environ.get("CER_BENCH_CPU_BUDGET", "300"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_baseline_child, args=(H, W, V, [list(c) for c in cascade], sd, budget, q), daemon=True)
    p.start()
    try:
        res = q.get(timeout=budget + 150.0)
    except _queue.Empty:
        res = {"value": None, "unit": "depth-maps/s", "kind": "port", "cores": None,
               "sample": f"the CPU baseline did not finish within {budget + 150.0:.0f} s on this box (host cores {os.cpu_count()}): stopped"}
    p.join(5.0)
    if p.is_alive():
        p.kill()
    return res


def cpu_baseline(H, W, V, cascade, sd, budget_s=300.0):
    """The oracle (oracle/cer_oracle.py: the reference's torch op sequence on CPU, fp32) on the bench workload itself - one
    whole depth map, same images / weights, nothing extrapolated - on this box's host cores.  Thread count (round 6, VERDICT r5
    "weak" 7): calibrated on the workload's OWN op sizes - one full-resolution image through the feature encoder, one source view's
    cost volume and one GRU iteration (lookup + update block) at the full 1/4-resolution grid, weighted by how often a depth map runs
    each - not on a small stand-in forward.  Candidates are 16 / 32 / 64 threads: torch's CPU kernels stop scaling, and the 528 small
    grid_samples per lookup get slower, beyond that, and ``os.cpu_count()`` (reported as ``host_cores``; BASELINE.md's stated setting)
    can exceed the cores the process may actually use (``usable_cores``)."""
    from oracle import cer_oracle as O
    from cer_mvs_amd.synthetic import synthetic_scene
    cores = os.cpu_count() or 1
    t_begin = time.perf_counter()
    runs = max(1, int(os.environ.get("CER_BENCH_CPU_RUNS", "3")))
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
    sdp = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    iters = sum(c[2] for c in cascade)
    with torch.no_grad():
        # ---- full-size pieces (inputs made once; what is timed is the op, with a warm-up call per thread count)
        h, w = H // 4, W // 4
        img1 = images[0, 1:2].float() * (2 / 255.0) - 1
        pz = poses[0].clone().float()
        pz[:, :3, 3] *= float(torch.as_tensor(scale).reshape(-1)[0])
        kz = intr[0].clone().float()
        kz[:, :2] /= 4
        torch.set_num_threads(min(cores, 32))
        fm2 = torch.cat([O.encoder(images[0, i:i + 1].float() * (2 / 255.0) - 1, sdp, "fnet.", "instance") for i in (0, 1)], 0)
        (D0, inc0, _), = O.resolve_cascade(cascade[:1])
        disp0 = torch.zeros(h, w)
        net0 = torch.tanh(fm2[:1]) * 0.5
        inp0 = torch.relu(fm2[1:2])
        vol, origin = O.cost_volume(fm2, pz[:2], kz[:2], D0, inc0, disp0, shift=True)
        levels = O.pyramid(vol, 3)
        pieces = {
            "encoder, 1 image": (lambda: O.encoder(img1, sdp, "fnet.", "instance"), V + 2),
            "cost volume, 1 view": (lambda: O.cost_volume(fm2, pz[:2], kz[:2], D0, inc0, disp0, shift=True), V * len(cascade)),
            "GRU iteration": (lambda: O.update_block(sdp, net0, inp0, disp0.view(1, 1, h, w), O.lookup(levels, origin, disp0, D0, inc0, 5), 0), iters),
        }
        cal, detail = {}, {}
        # candidates: 16 / 32 / 64 threads (more have only ever been slower for these ops, and os.cpu_count() may exceed the cores this process
        # may use: hundreds of OpenMP threads spinning on a smaller cgroup quota took > 50 minutes in a round-6 run).  A candidate whose warm-up
        # call of a piece takes more than 4 x the best time seen for that piece is dropped on the spot.
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
        best_piece = {}
        for c in sorted({c for c in (16, 32, 64) if c <= min(cores, usable)} or {min(cores, usable)}):
            if cal and time.perf_counter() - t_begin > 0.3 * budget_s:          # (a slow box: the candidates measured so far have to do)
                break
            torch.set_num_threads(c)
            tot, dropped = 0.0, False
            for name, (fn, weight) in pieces.items():
                t0 = time.perf_counter()
                fn()
                warm = time.perf_counter() - t0
                if name in best_piece and warm > 4.0 * best_piece[name] + 1.0:
                    dropped = True
                    break
                t0 = time.perf_counter()
                fn()
                dt = time.perf_counter() - t0
                best_piece[name] = min(best_piece.get(name, dt), dt)
                detail.setdefault(name, {})[c] = dt
                tot += weight * dt
            if not dropped:
                cal[c] = tot
        threads = min(cal, key=cal.get)

        def whole(nthreads, n):
            torch.set_num_threads(nthreads)
            ts = []
            for _ in range(n):
                if ts and (time.perf_counter() - t_begin) + max(ts) > budget_s:  # (the next run would not fit the budget: fewer runs, said in `sample`)
                    break
                t0 = time.perf_counter()
                O.raft_forward(sd, images, poses, intr, scale, cascade=cascade)
                ts.append(time.perf_counter() - t0)
            return ts
        times = whole(threads, runs)
        total = sorted(times)[len(times) // 2]
    best_total, best_threads = total, threads
    cal_txt = "; ".join(f"{name}: " + ", ".join(f"{c} thr {t:.2f} s" for c, t in sorted(d.items())) for name, d in detail.items())
    return {
        "value": 1.0 / best_total, "unit": "depth-maps/s", "cores": best_threads, "host_cores": cores, "kind": "port",
        "sample": (f"oracle/cer_oracle.py, ONE WHOLE depth map of the bench workload ({W}x{H}, {V} source views, "
                   f"{iters} GRU iterations): {len(times)} timed run(s), median {total:.1f} s, on {threads} of {cores} host cores; thread count "
                   f"calibrated on full-size pieces of this workload, weighted by their count per depth map ({cal_txt}; projected "
                   + ", ".join(f"{c} thr {t:.1f} s" for c, t in sorted(cal.items())) + "); CER_BENCH_CPU_RUNS sets the number of runs"),
        "seconds_per_depth_map": best_total, "runs_s": times, "usable_cores": usable,
        "thread_calibration_s": {str(c): t for c, t in sorted(cal.items())},
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
    if world == 1:
        args.mode = "shard"                               # (single GPU: the modes coincide)

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(mode):
        """W warm-up + K timed forwards in one sharding mode -> (seconds: max over ranks, output, model, inputs, weights)."""
        shard = world > 1 and mode in ("shard", "views")
        model = RAFT(cascade=cascade, test_mode=True, precision=args.precision, view_group=group if shard else None,
                     gru_precision=args.gru_precision, encoder_backend=args.encoder, shard="slab" if mode == "shard" else "views",
                     enc_precision=args.enc_precision, cost_precision=args.cost_precision)
        sd = fill_state_dict(model.state_dict(), seed=5)
        model.load_state_dict(sd)
        model = model.to(dev).eval()
        seed = 0 if (shard or world == 1) else rank          # replicas: a different reference frame per rank
        images, poses, intr, scale = synthetic_scene(H, W, V, seed=seed)
        inputs = (images.to(dev), poses.to(dev), intr.to(dev))
        S = 1 if shard else max(1, args.streams)
        lat = None
        with torch.no_grad():
            if S == 1:
                for _ in range(args.warmup):
                    out = model(*inputs, scale=scale)
                while model.gru_precision == "auto" and model._auto_pending():      # (the calibration forwards - RAFT.AUTO_INPUTS of them - stay out of the timed region)
                    out = model(*inputs, scale=scale)
                sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    out = model(*inputs, scale=scale)
                sync()
                elapsed = time.perf_counter() - t0
            else:
                # S depth maps in flight (DepthMapPipeline: one model replica + one HIP stream each, submitted round-robin): every
                # step is a complete forward; the steps overlap on the GPU.  One-at-a-time time first, for the record.
                from cer_mvs_amd.pipeline import DepthMapPipeline
                pipe = DepthMapPipeline(model, streams=S)
                for _ in range(max(args.warmup, S)):
                    out = pipe.result(pipe.submit(*inputs, scale), wait_on_host=False)
                while model.gru_precision == "auto" and any(m._auto_pending() for m in pipe.models):   # (calibration forwards + the replicas' adoption: untimed)
                    out = pipe.result(pipe.submit(*inputs, scale), wait_on_host=False)
                sync()
                out = model(*inputs, scale=scale)        # untimed: the caller's stream has its own allocator pool (first use = hipMalloc)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(min(args.steps, 3)):
                    out = model(*inputs, scale=scale)
                torch.cuda.synchronize()
                lat = (time.perf_counter() - t0) / min(args.steps, 3)
                sync()
                t0 = time.perf_counter()
                handles = [pipe.submit(*inputs, scale) for _ in range(args.steps)]
                out = pipe.result(handles[-1], wait_on_host=False)
                sync()
                elapsed = time.perf_counter() - t0
                assert all(torch.equal(pipe.result(h_), out) for h_ in handles)      # (same input every step: same output every step)
        assert torch.isfinite(out).all()
        timed_run.one_at_a_time_ms = None if lat is None else 1e3 * lat
        timed_run.streams = S
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt[0])
        return elapsed, out, model, inputs, scale, sd

    modes = None
    if args.mode == "both":
        # north_star's scheme (views sharded, view-sum volume all-reduced once per stage, GRU replicated) and the row-slab scheme,
        # one after the other; the headline is the faster one (decided on the max-over-ranks times, so every rank agrees) and the
        # rest of the line - kernels, roofline - describes that mode
        modes = {}
        for m, label in (("views", f"view-shard x{world} + all-reduce/stage"), ("shard", f"row-slab x{world}")):
            try:
                e_m, out_m, model_m, _, _, _ = timed_run(m)
                modes[m] = {"value": args.steps / e_m, "ms_per_step": 1e3 * e_m / args.steps, "parallelism": label}
                del model_m, out_m
            except Exception as exc:                       # (a scheme the transport refuses must not cost the other scheme's line)
                modes[m] = {"error": f"{type(exc).__name__}: {exc}"[:300], "parallelism": label}
            torch.cuda.empty_cache()
        try:
            # context, never the headline: N independent depth maps, one per GPU, no communication (weak scaling)
            e_r, out_r, model_r, _, _, _ = timed_run("replica")
            modes["replica"] = {"value": world * args.steps / e_r, "ms_per_step": 1e3 * e_r / args.steps, "scaling": "weak",
                                "parallelism": f"replica x{world}: {world} independent depth maps per step"}
            del model_r, out_r
        except Exception as exc:
            modes["replica"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        torch.cuda.empty_cache()
        ok = [m for m in ("shard", "views") if "ms_per_step" in modes[m]]
        if not ok:
            raise SystemExit(f"bench.py: both sharding schemes failed: {modes}")
        args.mode = min(ok, key=lambda m: modes[m]["ms_per_step"])
    shard = world > 1 and args.mode in ("shard", "views")
    elapsed, out, model, inputs, scale, sd = timed_run(args.mode)
    requested_precision = args.gru_precision
    if args.gru_precision == "auto":     # the form the calibration kept (cer-mvs_amd/raft.py: RAFT._forward_calibrating) is what was timed
        assert model.auto_choice is not None and model.auto_choice.partition("+")[0] in ("s16f6", "s16f8", "s16"), "the warm-up did not calibrate gru_precision='auto'"
        args.gru_precision = model.auto_choice.partition("+")[0]
    enc_f6 = bool(model._enc_f6)          # what the timed forwards ran the encoders in
    cost_x2 = bool(model._cost_x2)        # ... and the cost volume's dots
    if modes is not None:
        modes[args.mode].update(value=args.steps / elapsed, ms_per_step=1e3 * elapsed / args.steps, headline=True)
    maps = args.steps * (world if (world > 1 and not shard) else 1)
    # N > 1: the sharded modes keep ONE depth map in flight, the N = 1 headline three - the speed-up of N GPUs has to be taken against
    # the single-GPU one-at-a-time forward (VERDICT r3 "weak" 7).  Rank 0 times it on its own GPU while the other ranks wait.
    n1_ref = None
    if world > 1:
        if rank == 0:
            m1 = RAFT(cascade=cascade, test_mode=True, precision=args.precision, gru_precision=args.gru_precision, encoder_backend=args.encoder,
                      enc_precision="f6" if enc_f6 else "f16x3", cost_precision="x2" if cost_x2 else "x3")
            m1.load_state_dict(sd)
            m1 = m1.to(dev).eval()
            with torch.no_grad():
                for _ in range(3):
                    m1(*inputs, scale=scale)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                k1 = max(3, min(args.steps, 10))
                for _ in range(k1):
                    m1(*inputs, scale=scale)
                torch.cuda.synchronize()
            t1 = (time.perf_counter() - t0) / k1
            n1_ref = {"ms_per_depth_map": 1e3 * t1, "value": 1.0 / t1,
                      "speedup_of_this_line": (maps / elapsed) * t1 if shard else None,
                      "note": "rank 0's single-GPU forward of the same workload, one depth map at a time, unsharded: the base of the strong-scaling ratio"}
            del m1
        sync()

    result = None
    rec = None
    if rank == 0 or shard:               # a sharded forward is collective: every rank runs the instrumented pass
        rec, inst_wall_ms = kernel_timing(model, inputs, scale)
    if rank == 0:
        h4, w4 = H // 4, W // 4
        P = h4 * w4
        if shard and args.mode == "shard":
            # row-slab sharding: rank 0's kernels process its owned rows plus the halo, not the whole image - the per-launch
            # roofline figures below are per rank and must use that pixel count
            from cer_mvs_amd import slab as _slab
            if _slab.can_shard(h4, world):
                _, _, e0, e1 = _slab.slab_bounds(h4, world, 0)
                P = (e1 - e0) * w4
        Vloc = V if not (shard and args.mode == "views") else (V + world - 1) // world
        C, L_, r_ = 64, 3, 5
        stages = model.stages()
        split = args.gru_precision in ("s16f6", "s16f8", "s16", "f16x3")
        mfma_peak = F16_MFMA_PEAK_TFLOPS / 3 if split else FP32_MFMA_PEAK_TFLOPS      # encoders, and the 3-term update-block convs
        # update-block convs in the fp8-correction form: per 16-channel tap step one f16 MFMA (32 cycles) + half of one fp8 K = 64
        # MFMA (64 cycles for two steps) = 2 f16-MFMA times of matrix-pipe occupancy per fp32 product instead of 3
        # ... in the FP6-correction form (round 6) the K = 64 MFMA takes 32 cycles: 1.5 f16-MFMA times per fp32 product
        gru_peak = F16_MFMA_PEAK_TFLOPS / 1.5 if args.gru_precision == "s16f6" else F16_MFMA_PEAK_TFLOPS / 2 if args.gru_precision == "s16f8" else mfma_peak
        # ---- algorithmic work per launch, SURVEY.md 8(d) (fp32, view-mean folded):
        #   build(s): bytes 4[P C (V+1) + P] + 4 P 1.75 D_s, flops 512 V P D_s;  lookup: 284 P bytes per iteration;
        #   GRU convs: 2 * 9 * K * N * P flops with the hoisted `inp` slice and the collapsed encoder NOT credited (K = the
        #   reference's channel counts: 64 + 49 + 64);  encoders: 71 GFLOP per image.
        alg = {}
        for si, (D, _, _) in enumerate(stages):
            alg[f"cost_build_stage{si}"] = dict(bytes=4.0 * (P * C * (Vloc + 1) + P) + 4.0 * P * 1.75 * D, flops=512.0 * Vloc * P * D,
                                                bound="hbm (gather-dot, ~100 FLOP/B: also shown against the fp32 vector peak)")
        alg["lookup_encode_f32"] = dict(bytes=284.0 * P, bound="hbm",
                                    note="8(d) bytes: 3 windows of 12 taps + origin + disp in, 33 floats out; the kernel writes the fused "
                                         "1x1 conv's 64 floats instead (bytes_fused = 408 P)", bytes_fused=408.0 * P)
        alg["conv3x3_gates_zr"] = dict(flops=2.0 * 9 * 177 * 128 * P, bound="mfma")
        alg["conv3x3_gru_q"] = dict(flops=2.0 * 9 * 177 * 64 * P, bound="mfma")
        alg["conv3x3_relu_64"] = dict(flops=2.0 * 9 * 64 * 64 * P, bound="mfma")
        alg["conv3x3_delta_fused"] = dict(flops=2.0 * 9 * 64 * 256 * P + 2.0 * 9 * 256 * P, bound="mfma")
        alg["conv3x3_linear"] = dict(flops=2.0 * 9 * 64 * 96 * P, bound="mfma", note="hoisted inp slice: z|r (128) and q (64) launches, mean")
        kern = {}
        for k, (n, t) in sorted(rec.items()):
            e = {"launches": n, "avg_us": 1e3 * t / n, "total_ms": t}
            if k in alg:
                a_ = alg[k]
                sec = t / n * 1e-3
                if "bytes" in a_:
                    e.update(alg_bytes=a_["bytes"], GBps=a_["bytes"] / sec / 1e9, frac_hbm=a_["bytes"] / sec / 1e9 / HBM_PEAK_GBS)
                    if "bytes_fused" in a_:
                        e.update(GBps_fused=a_["bytes_fused"] / sec / 1e9, frac_hbm_fused=a_["bytes_fused"] / sec / 1e9 / HBM_PEAK_GBS)
                if "flops" in a_:
                    pk = (gru_peak if k.startswith("conv3x3") else mfma_peak) if a_["bound"] == "mfma" else FP32_MFMA_PEAK_TFLOPS
                    e.update(alg_flops=a_["flops"], TFLOPs=a_["flops"] / sec / 1e12, frac_flops=a_["flops"] / sec / 1e12 / pk, flops_peak=pk)
                e["bound"] = a_["bound"]
                if "note" in a_:
                    e["note"] = a_["note"]
            kern[k] = e
        cl_key = "cost_lines_kernel_x2" if cost_x2 else "cost_lines_kernel"            # (round 6: the two-term form has its own PMC row)
        for cand in ("r03_pmc_traffic.json", "r04_pmc_traffic.json", "r05_pmc_traffic.json", "r06_pmc_traffic.json"):   # HBM traffic of the cost-volume kernel from the
            try:                                                  # committed PMC passes (the latest file holding the kernel wins)
                with open(os.path.join(REPO, "profiles", cand)) as f:
                    pm = json.load(f).get(cl_key, {})
                for k in kern:
                    if k.startswith("cost_build_stage") and "traffic_bytes" in pm:
                        kern[k]["pmc_traffic_bytes_avg_of_both_stages"] = pm["traffic_bytes"]
                        kern[k]["pmc_source"] = f"profiles/{cand} ({cl_key}, mean of the stage-0 and stage-1 launches)"
            except Exception:
                pass
        enc = [(k, v) for k, v in rec.items() if k.startswith("enc_")]
        enc_ms = 0.0
        if enc and world == 1:
            # the encoder section as a whole, GPU-paced: three back-to-back encode() calls between two events (the per-launch event
            # times of the instrumented pass are host-paced for these short launches and over-count: VERDICT r2)
            views_all = list(range(1, V + 1))
            with torch.no_grad():
                imgs_raw = inputs[0].float()
                model.encode(imgs_raw, views_all, raw=True)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(3):
                    model.encode(imgs_raw, views_all, raw=True)
                ev1.record()
                torch.cuda.synchronize()
            enc_ms = ev0.elapsed_time(ev1) / 3
        if enc_ms > 0:
            nimg = (Vloc + 1) + 1                              # fnet on the reference + source views, cnet on the reference
            enc_flops = 71.0e9 * nimg
            kern["encoders_total"] = {"launches": sum(n for _, (n, _) in enc), "total_ms": enc_ms, "alg_flops": enc_flops,
                                      "TFLOPs": enc_flops / (enc_ms * 1e-3) / 1e12, "frac_flops": enc_flops / (enc_ms * 1e-3) / 1e12 / mfma_peak,
                                      "flops_peak": mfma_peak, "bound": "mfma",
                                      "note": f"{nimg} encoder passes x 71 GFLOP (SURVEY.md 8(a) row 2); total_ms = the whole encode() "
                                              "section timed GPU-paced (3 back-to-back calls between two HIP events), not the sum of the "
                                              "host-paced per-launch events listed under enc_*"}
        n_zr, t_zr = rec["conv3x3_gates_zr"]
        flops_zr = alg["conv3x3_gates_zr"]["flops"]
        achieved = flops_zr / (t_zr / n_zr * 1e-3) / 1e12
        if args.gru_precision == "s16f6":
            kname = ("conv3x3_s16_kernel<1,4,4,GATES,F6> (z|r gates, 3x3, K=177, N=128; per fp32 product one f16 MFMA + the two 2^-11 "
                     "correction terms on the FP6 (e2m3, one E8M0 scale per 16-channel block) form of the block-scaled MFMA, one accumulator)")
        elif args.gru_precision == "s16f8":
            kname = ("conv3x3_s16_kernel<1,4,4,GATES,F8> (z|r gates, 3x3, K=177, N=128; per fp32 product one f16 MFMA + the two 2^-11 "
                     "correction terms on the block-scaled fp8 MFMA, one accumulator)")
        elif args.gru_precision == "s16":
            kname = "conv3x3_s16_kernel<1,4,4,GATES> (z|r gates, 3x3, K=177, N=128; 3 f16 MFMAs per fp32 product, one accumulator)"
        elif args.gru_precision == "f16x3":
            kname = "conv3x3_f16x3_kernel<4,2,1,2,3,4,GATES> (z|r gates, 3x3, K=177, N=128; 3 f16 MFMAs per fp32 product)"
        else:
            kname = "conv3x3_kernel<2,2,4,4,GATES> (z|r gates, 3x3, K=177, N=128; exact fp32 MFMA)"
        traffic, traffic_src = None, None                        # HBM bytes per launch from the committed PMC passes
        for cand in ("r02_pmc_traffic.json", "r03_pmc_traffic.json", "r04_pmc_traffic.json", "r05_pmc_traffic.json", "r06_pmc_traffic.json"):   # (the latest file holding the kernel wins)
            try:
                with open(os.path.join(REPO, "profiles", cand)) as f:
                    pm_ = json.load(f)
                    traffic = (pm_["conv3x3_gates_zr_f6"]["traffic_bytes"] if args.gru_precision == "s16f6" else
                               pm_["conv3x3_gates_zr_f8"]["traffic_bytes"] if args.gru_precision == "s16f8" else
                               pm_["conv3x3_gates_zr"]["traffic_bytes"] if args.gru_precision == "s16" else None)
                traffic_src = f"profiles/{cand} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2x FETCH correction)"
            except Exception:
                pass
        roofline = {"kernel": kname, "bound": "mfma", "achieved": achieved, "peak": gru_peak, "unit": "TFLOP/s", "frac": achieved / gru_peak,
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": 1e3 * t_zr / n_zr, "launches": n_zr,
                    "flops_per_launch": flops_zr,
                    "peak_note": ("fp32-equivalent ceiling = matrix-pipe time of one f16 MFMA (2500 TF dense) + half an FP6 K=64 MFMA (10 PF dense) "
                                  "per 16-channel tap step = 2500 / 1.5 = 1667; frac_of_fp8_form_ceiling (2500 / 2, rounds 3-5) and "
                                  "frac_of_three_term_ceiling (2500 / 3, rounds 2-3) hold the same launch against the earlier forms' ceilings"
                                  if args.gru_precision == "s16f6" else
                                  "fp32-equivalent ceiling = matrix-pipe time of one f16 MFMA (2500 TF dense) + half an fp8 K=64 MFMA (5000 TF "
                                  "dense) per 16-channel tap step = 2500 / 2; against the all-f16 form's ceiling (2500 / 3, rounds 2-3) the "
                                  "same launch is at frac_of_three_term_ceiling" if args.gru_precision == "s16f8" else
                                  "fp32-equivalent ceiling = 2500 TF dense f16 MFMA / 3" if split else "fp32 MFMA dense peak"),
                    "frac_of_three_term_ceiling": (achieved / (F16_MFMA_PEAK_TFLOPS / 3)) if split else None,
                    "frac_of_fp8_form_ceiling": (achieved / (F16_MFMA_PEAK_TFLOPS / 2)) if args.gru_precision in ("s16f6", "s16f8") else None,
                    "frac_of_raw_f16_peak": ((1.5 if args.gru_precision == "s16f6" else 2 if args.gru_precision == "s16f8" else 3) * achieved / F16_MFMA_PEAK_TFLOPS) if split else None}
        hbm = None
        if "lookup_encode_f32" in kern:
            lk = kern["lookup_encode_f32"]
            hbm = {"kernel": "lookup_encode_kernel (multi-level lookup + view mean + 1x1 conv)", "bound": "hbm", "achieved": lk["GBps"],
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lk["frac_hbm"], "avg_launch_us": lk["avg_us"],
                   "bytes_per_launch": lk["alg_bytes"], "bytes_formula": "SURVEY.md 8(d): 284 * P per iteration",
                   "frac_with_fused_output_bytes": lk["frac_hbm_fused"], "bytes_fused_output": 408.0 * P}
        loop_keys = ("lookup_encode_f32", "conv3x3_relu_64", "conv3x3_gates_zr", "conv3x3_gru_q", "conv3x3_delta_fused", "delta_sum_f32")
        loop_ms = sum(rec[k][1] for k in loop_keys if k in rec)
        iters = sum(c[2] for c in cascade)
        inner = {"ms_per_depth_map": loop_ms, "us_per_iteration": 1e3 * loop_ms / iters,
                 "alg_bytes_per_iteration": (284.0 + 908.0) * P, "alg_flops_per_iteration": 1210368.0 * P,
                 "frac_hbm": (284.0 + 908.0) * P * iters / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "frac_mfma": 1210368.0 * P * iters / (loop_ms * 1e-3) / 1e12 / gru_peak,
                 "note": "correlation + GRU inner loop against both rooflines with SURVEY.md 8(d)'s unreduced per-iteration figures "
                         "(1.21 MFLOP and 1192 B per pixel): the loop is MFMA-bound, not HBM-bound (SURVEY.md 7, hard part 1)"}
        result = {
            "metric": "depth-maps/sec (ref+N src views) at DTU 1600x1184; HBM GB/s vs roofline",
            "value": maps / elapsed, "unit": "depth-maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if (shard or world == 1) else "weak", "vs_baseline": None,
            "dtype": ("f32" if args.precision == "fp32" else "f32 (encoders f16 autocast)")
                     + ((" [dense convs: f32 operands split into 2 x f16, f32 accumulate; encoders 3 f16 MFMAs per product (fp32-class), "
                        "update-block convs f16 main term + the two 2^-11 correction terms in e4m3 on the fp8 MFMA: 4e-6 relative L1 from fp32 "
                        "end to end, bar 1e-4]" if args.gru_precision == "s16f8" else
                        " [dense convs: f32 operands split into 2 x f16, f32 accumulate; encoders 3 f16 MFMAs per product (fp32-class), "
                        "update-block convs f16 main term + the two 2^-11 correction terms in e2m3 (FP6, one power-of-two scale per 16-channel "
                        "block) on the block-scaled MFMA: ~5e-6 relative L1 from fp32 end to end, bar 1e-4]" if args.gru_precision == "s16f6" else
                        " [dense convs: f32 operands split into 2 x f16, 3 MFMAs per product, f32 accumulate - fp32-class]" if split else "")
                        .replace("encoders 3 f16 MFMAs per product (fp32-class), ",
                                 "encoders f16 main term + the two 2^-11 correction terms in e2m3 (FP6, one power-of-two scale per 16-channel block: features "
                                 "4e-5 from the three-term form), " if enc_f6 else "encoders 3 f16 MFMAs per product (fp32-class), ")),
            "data": "synthetic",
            "enc_precision": {"requested": args.enc_precision, "timed": "f6" if enc_f6 else "f16x3"},
            "cost_precision": {"requested": args.cost_precision, "timed": "x2" if cost_x2 else "x3",
                               "note": "x2: the cost volume's 64-channel dots take the source features as f16 (their lo planes are not read; the reference rows keep "
                                       "both halves): 1.2e-5 relative L1 on the volume"},
            "gru_precision": {"requested": requested_precision, "timed": args.gru_precision, "auto_form": model.auto_choice,
                              **({"calibration_rel_l1_vs_s16": model.auto_error, "candidates": list(model._auto_forms()), "tolerance": model.AUTO_TOL,
                                  "calibration_inputs": model.AUTO_INPUTS,
                                  "note": "gru_precision='auto': the first forwards of a set of weights (calibration_inputs of them, inside the "
                                          "warm-up) run in the fp32-class form 's16' and in the cheapest candidate still standing ('+c2': two-term cost-volume dots; '+e6': "
                                          "the encoders' correction terms in FP6; both on top of the update block's fp8 corrections, then those alone); a candidate is kept only if it agrees with 's16' within the tolerance on every one of "
                                          "them (the figure is the kept form's worst)"}
                                 if requested_precision == "auto" else {})},
            "config": {"workload": args.workload, "image": f"{W}x{H}", "src_views": V, "cascade": cascade,
                       "gru_iters": sum(c[2] for c in cascade),
                       "depth_maps_in_flight": getattr(timed_run, "streams", 1),
                       "parallelism": "single" if world == 1 else (
                           (f"row-slab x{world}: source-feature all-gather (overlapped with the reference encode) + 7-row halo exchanged "
                            f"point-to-point with the 2 neighbours per GRU iteration" if args.mode == "shard"
                            else f"view-shard x{world} + all-reduce/stage") if shard else f"replica x{world}"),
                       **({"backend": "gloo (validation run, ranks may share a GPU)"} if (world > 1 and args.backend == "gloo") else {})},
            **({"modes": modes} if modes is not None else {}),
            **({"n1_one_at_a_time": n1_ref} if n1_ref is not None else {}),
            **({"one_at_a_time": {"ms_per_depth_map": timed_run.one_at_a_time_ms, "value": 1e3 / timed_run.one_at_a_time_ms,
                                  "note": "the same forward with ONE depth map in flight (--streams 1): the latency of a depth map, and what "
                                          "`value` was in rounds 1-2.  `value` / `ms_per_step` above are the throughput with "
                                          f"{timed_run.streams} independent depth maps in flight on {timed_run.streams} HIP streams "
                                          "(cer-mvs_amd/pipeline.py, the product's inference() default): every step is a complete "
                                          "forward, the steps overlap and fill each other's partially filled last waves of tiles.  The "
                                          "roofline / kernels entries below come from a one-at-a-time instrumented pass"}}
               if getattr(timed_run, "one_at_a_time_ms", None) else {}),
            "roofline": roofline, "roofline_hbm_kernel": hbm, "inner_loop": inner, "kernels": kern,
            "instrumented_pass": (lambda sums, enc_t: {
                "wall_ms": inst_wall_ms,
                # a REAL sum: every class's event times added up; the encoder classes - short launches whose host-paced
                # events over-count - replaced by the GPU-paced time of the whole encode() section where it was measured
                "sum_of_kernels_ms": sum(t for k, t in sums.items() if not (enc_t and k.startswith("enc_"))) + (enc_t or 0.0),
                "sum_of_kernel_events_ms": sum(sums.values()),
                "note": "one extra forward after the timed region with HIP events around every library launch and launch plans disabled; "
                        "its per-launch MEDIANS feed `kernels[*].avg_us` / `roofline`; sum_of_kernels_ms adds the event times themselves "
                        "(encoder launches: the GPU-paced encode() section, kernels.encoders_total) and is comparable with the rocprofv3 kernel "
                        "sum of profiles/r05_kernel_stats_s1.md; its wall time is NOT ms_per_step"})(
                    getattr(kernel_timing, "sums", {}), kern.get("encoders_total", {}).get("total_ms")),
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
            result["cpu_baseline"] = cpu_baseline_guarded(H, W, V, cascade, {k: v.detach().cpu() for k, v in sd.items()})
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
