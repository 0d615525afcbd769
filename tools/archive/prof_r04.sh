#!/bin/bash
# Round-4 profiles at the shipped defaults (gru_precision="auto" -> s16f8 on the bench weights; encoder engine "pc"; three depth maps in
# flight): bench lines, rocprofv3 kernel statistics (default command = 3 in flight, and --streams 1), the encoder timeline, PMC passes
# for the encoder's producer / consumer kernels and the z|r conv.  Run on the GPU box: gpurun -- tools/archive/prof_r04.sh
out=gpurun_out/r04
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $out/bench.json
python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $out/streams1_bench.json
for s in 3 1; do
  rocprofv3 --kernel-trace --stats -d $out/prof_s$s -o bench -- python bench.py --streams $s --no-cpu-baseline --steps 12 --warmup 5 > $out/prof_s$s.log 2>&1
  db=$(find $out/prof_s$s -name "*.db" | head -1)
  python tools/rocpd_summary.py "$db" $out/kernel_stats_s$s.md --per "conv3x3_s16_kernel<1, 4, 4, 2, 1>:32" > /dev/null 2>&1
  find $out/prof_s$s -name "*.db" -delete; find $out/prof_s$s -name "*.csv" -size +2M -delete
done
tools/archive/prof_enc_rocprof.sh $out/enc pc > /dev/null 2>&1
[ -n "$SKIP_PMC" ] || tools/archive/pmc_r04.sh $out/pmc > /dev/null 2>&1
python -m pytest tests/test_hip_parity.py -q -m gpu -k "end_to_end_cfg1 or end_to_end_cfg2" -s 2>&1 | grep -E "rel-L1|passed|failed" > $out/e2e.log
