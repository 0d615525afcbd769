#!/bin/bash
# Round-5 kernel statistics: rocprofv3 --kernel-trace --stats of tools/forward_loop.py (identical forwards only: "per forward" = total / forwards),
# one at a time and three in flight; optional env for A/B (CER_DELTA_MERGED=0 ...).  usage: tools/archive/prof_r05.sh [outdir] [tag]
out=${1:-gpurun_out/r05}; tag=${2:-}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 1 3; do
  rocprofv3 --kernel-trace --stats -d $out/prof${tag}_s$s -o fw -- python tools/forward_loop.py --streams $s --forwards 14 > $out/prof${tag}_s$s.log 2>&1
  db=$(find $out/prof${tag}_s$s -name "*.db" | head -1)
  python tools/rocpd_summary.py "$db" $out/kernel_stats${tag}_s$s.md --per "conv3x3_s16_kernel<1, 4, 4, 2, 1:32" > /dev/null 2>&1
  find $out/prof${tag}_s$s -name "*.db" -delete; find $out/prof${tag}_s$s -name "*.csv" -size +2M -delete
done
head -30 $out/kernel_stats${tag}_s1.md
