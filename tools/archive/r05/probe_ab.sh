#!/bin/bash
# usage: probe_ab.sh <nameA> <nameB> ...   timings of corr2 / q on probe libraries (interleaved, twice), then traces of <name>t libraries if present
out=gpurun_out/r05p; mkdir -p $out
V=$PWD/cer-mvs_amd/csrc/variants
for rep in 1 2; do for n in "$@"; do echo "== $n"; CER_MVS_LIB=$V/libcermvs_probe_$n.so timeout 300 python tools/bench_conv_s16.py --f8 --rounds 3 --reps 10 --only "corr2,q gru" 2>&1 | grep -v amdgpu.ids | cut -c1-60; done; done | tee $out/ab_$1_$2.txt
for n in "$@"; do if [ -f $V/libcermvs_probe_${n}t.so ]; then echo "== trace $n"; CER_MVS_LIB=$V/libcermvs_probe_${n}t.so timeout 300 python tools/trace_s16.py --f8 --conv q --mt 2 2>&1 | grep "lifetime\|prologue\|main loop \|epilogue  \|9-tap group \|by group index\|disparity section\|runs alone"; fi; done | tee $out/trace_$1_$2.txt
