#!/bin/bash
# Round-6 evidence at the shipped defaults, ONE box, one call: bench lines (default = three depth maps in flight + CPU baseline; --streams 1; the
# opt-in FP6 form), rocprofv3 kernel statistics of identical forwards (tools/prof_r06.sh), what each component costs in either regime
# (tools/r06/whatif.py), PMC passes solo and with three in flight (tools/pmc_r06.sh), the other BASELINE workloads, the 2-rank gloo bench path,
# one rank's share of the sharded forwards (tools/rank_share.py), e2e parity.  Run on the GPU box: gpurun -- tools/prof_r06_final.sh
# usage: tools/prof_r06_final.sh [sections]   sections: any of  bench prof pmc rest  (default: all; one gpurun call per section keeps a call short)
out=gpurun_out/r06
mkdir -p $out
sec=${1:-"bench prof pmc rest"}
if [[ $sec == *bench* ]]; then
timeout 900 python bench.py --steps 20 --warmup 5 2>$out/bench.err | grep "^{" > $out/bench.json
fi
if [[ $sec == *prof* ]]; then
timeout 600 python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $out/streams1_bench.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gru-precision s16f6 2>/dev/null | grep "^{" > $out/s16f6_bench.json
tools/prof_r06.sh $out > /dev/null 2>&1
timeout 600 python tools/r06/whatif.py > $out/whatif.txt 2>&1
python tools/r06/bench_conv_forms.py > $out/conv_forms.txt 2>&1
bash tools/r06/cost_ab.sh default > /dev/null 2>&1; cp gpurun_out/r06e/cost_ab.txt $out/cost_lines.txt
fi
if [[ $sec == *pmc* ]]; then
tools/pmc_r06.sh $out/pmc > /dev/null 2>&1
fi
if [[ $sec == *rest* ]]; then
timeout 900 python tools/rank_share.py --json $out/rank_share.json > $out/rank_share.log 2>&1
for wl in blended_2048x1536_v7_it16 tnt_3840x2160_v15_it16 dtu_640x480_v2_it4; do
  for s in 3 1; do
    timeout 600 python bench.py --workload $wl --streams $s --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'workload': d['config']['workload'], 'depth_maps_in_flight': d['config']['depth_maps_in_flight'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'peak_device_memory_gb': d['peak_device_memory_gb'], 'gru_precision': d['gru_precision']['timed'], 'auto_form': d['gru_precision'].get('auto_form'), 'calibration_rel_l1_vs_s16': d['gru_precision'].get('calibration_rel_l1_vs_s16')}))"
  done
done > $out/other_workloads.jsonl
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 --no-cpu-baseline 2> $out/bench_2rank_gloo.err | grep "^{" > $out/bench_2rank_gloo.json
python -m pytest tests/test_hip_parity.py -q -m gpu -k "end_to_end_cfg1 or end_to_end_cfg2" -s 2>&1 | grep -E "rel-L1|passed|failed" > $out/e2e.log
fi
python - <<'P'
import json
for f in ("bench.json", "streams1_bench.json", "s16f6_bench.json"):
    d = json.load(open("gpurun_out/r06/" + f)); print(f, round(d["value"], 2), round(d["ms_per_step"], 3), d.get("one_at_a_time", {}).get("value"), d["roofline"]["frac"], d["instrumented_pass"].get("sum_of_kernels_ms"), d["gru_precision"])
P
