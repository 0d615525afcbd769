#!/bin/bash
out=gpurun_out/r05; mkdir -p $out
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $out/gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $out/bench.json
python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $out/streams1_bench.json
tools/archive/prof_r05.sh $out > /dev/null 2>&1
python - <<'P'
import json
for f in ("bench.json", "streams1_bench.json"):
    d = json.load(open("gpurun_out/r05/" + f)); print(f, round(d["value"], 2), round(d["ms_per_step"], 3), d.get("one_at_a_time", {}).get("value"), round(d["roofline"]["frac"], 3), d["roofline"]["traffic_source"][:28], d["instrumented_pass"].get("sum_of_kernels_ms"), d.get("parity"))
P
tail -3 $out/kernel_stats_s1.md
