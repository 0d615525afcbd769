#!/bin/bash
# Round-3 profiles of the default configuration (gru_precision="s16f8"): bench lines, rocprofv3 kernel statistics (three depth maps in
# flight - the default - and one at a time) and the PMC passes of the z|r gate convolution.  Run on the GPU box: gpurun -- tools/archive/prof_r03c.sh
out=gpurun_out/r03c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py 2>/dev/null | grep "^{" > $out/bench.json
python bench.py --streams 1 2>/dev/null | grep "^{" > $out/streams1_bench.json
for s in 3 1; do
  rocprofv3 --kernel-trace --stats -d $out/prof_s$s -o bench -- python bench.py --streams $s --no-cpu-baseline > $out/prof_s$s.log 2>&1
  db=$(find $out/prof_s$s -name "*.db" | head -1)
  python tools/rocpd_summary.py "$db" $out/kernel_stats_s$s.md --per "conv3x3_s16_kernel<1, 4, 4, 2, 1>:32" > /dev/null 2>&1
  find $out/prof_s$s -name "*.db" -delete; find $out/prof_s$s -name "*.csv" -size +2M -delete
done
[ -n "$SKIP_PMC" ] || tools/archive/pmc_r03b.sh $out/pmc > /dev/null 2>&1
python -m pytest tests/test_hip_parity.py -q -m gpu -k "end_to_end_cfg1 or odd_image" -s 2>&1 | grep -E "rel-L1|passed|failed" > $out/e2e.log
