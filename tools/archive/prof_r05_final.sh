#!/bin/bash
# Round-5 evidence at the shipped defaults, ONE box, one call: bench lines (default = three depth maps in flight + CPU baseline; --streams 1),
# rocprofv3 kernel statistics of identical forwards (tools/archive/prof_r05.sh), PMC passes (tools/archive/pmc_r05.sh), the other BASELINE workloads, the 2-rank
# gloo bench path, one rank's share of the sharded forwards (tools/rank_share.py), e2e parity.  Run on the GPU box: gpurun -- tools/archive/prof_r05_final.sh
out=gpurun_out/r05
mkdir -p $out
python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $out/bench.json
python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $out/streams1_bench.json
tools/archive/prof_r05.sh $out > /dev/null 2>&1
[ -n "$SKIP_PMC" ] || tools/archive/pmc_r05.sh $out/pmc > /dev/null 2>&1
python tools/rank_share.py --json $out/rank_share.json > $out/rank_share.log 2>&1
for wl in blended_2048x1536_v7_it16 tnt_3840x2160_v15_it16 dtu_640x480_v2_it4; do
  for s in 3 1; do
    python bench.py --workload $wl --streams $s --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'workload': d['config']['workload'], 'depth_maps_in_flight': d['config']['depth_maps_in_flight'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'peak_device_memory_gb': d['peak_device_memory_gb'], 'gru_precision': d['gru_precision']['timed']}))"
  done
done > $out/other_workloads.jsonl
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 --no-cpu-baseline 2> $out/bench_2rank_gloo.err | grep "^{" > $out/bench_2rank_gloo.json
python -m pytest tests/test_hip_parity.py -q -m gpu -k "end_to_end_cfg1 or end_to_end_cfg2" -s 2>&1 | grep -E "rel-L1|passed|failed" > $out/e2e.log
python - <<'P'
import json
for f in ("bench.json", "streams1_bench.json"):
    d = json.load(open("gpurun_out/r05/" + f)); print(f, round(d["value"], 2), round(d["ms_per_step"], 3), d.get("one_at_a_time", {}).get("value"), d["roofline"]["frac"], d["instrumented_pass"].get("sum_of_kernels_ms"))
P
