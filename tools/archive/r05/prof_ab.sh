#!/bin/bash
CER_DELTA_MERGED=0 tools/archive/prof_r05.sh gpurun_out/r05g _split > /dev/null 2>&1
tools/archive/prof_r05.sh gpurun_out/r05g _merged > /dev/null 2>&1
for t in _split _merged; do echo "== $t"; grep "conv3x3_s16\|lookup\|delta_sum\|forwards in window" gpurun_out/r05g/kernel_stats${t}_s1.md | cut -c1-130; done
