#!/bin/bash
out=gpurun_out/r05e; mkdir -p $out
V=$PWD/cer-mvs_amd/csrc/variants
for rep in 1 2; do for n in "$@"; do if [ $n = default ]; then lib=$PWD/cer-mvs_amd/csrc/libcermvs.so; else lib=$V/libcermvs_$n.so; fi; CER_MVS_LIB=$lib timeout 300 python tools/archive/r05/bench_cost.py $n 2>&1 | grep "stage"; done; done | tee $out/cost_ab.txt
