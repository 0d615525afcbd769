#!/bin/bash
# Reproduction of the round-2 lookup specialisation's intermittent failure: N concurrent processes on one GPU, each running the bench
# workload M times and comparing every output with its first one and with the reference capture (tools/stress_parity.py).
# usage: tools/archive/repro_lookup_spec.sh [lib] [processes] [forwards]     (run on the GPU box)
lib=${1:-cer-mvs_amd/csrc/variants/libcermvs_lkspec.so}
np=${2:-2}
m=${3:-100}
for i in $(seq 1 $np); do
  CER_MVS_LIB=$lib python tools/stress_parity.py $m > /tmp/repro_$i.log 2>&1 &
done
wait
for i in $(seq 1 $np); do grep -v amdgpu.ids /tmp/repro_$i.log | tail -4; done
