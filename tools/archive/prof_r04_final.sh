#!/bin/bash
# End-of-round-4 evidence at the shipped defaults: tools/archive/prof_r04.sh (bench lines, rocprofv3 kernel statistics with three depth maps in flight
# and one at a time, encoder timeline, PMC passes, e2e parity log) + the other BASELINE workloads + the 2-rank gloo bench path + the
# schedule-fuzz determinism run.  Run on the GPU box: gpurun -- tools/archive/prof_r04_final.sh
out=gpurun_out/r04
tools/archive/prof_r04.sh
for wl in blended_2048x1536_v7_it16 tnt_3840x2160_v15_it16 dtu_640x480_v2_it4; do
  for s in 3 1; do
    python bench.py --workload $wl --streams $s --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'workload': d['config']['workload'], 'depth_maps_in_flight': d['config']['depth_maps_in_flight'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'peak_device_memory_gb': d['peak_device_memory_gb'], 'gru_precision': d['gru_precision']['timed']}))"
  done
done > $out/other_workloads.jsonl
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 --no-cpu-baseline 2> $out/bench_2rank_gloo.err | grep "^{" > $out/bench_2rank_gloo.json
