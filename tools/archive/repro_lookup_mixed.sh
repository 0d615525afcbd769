#!/bin/bash
# the lookup kernel alone (library $1) in one process while two other processes run whole forwards (shipped library): interference
# from kernels of every kind.  usage: tools/archive/repro_lookup_mixed.sh <lib> <launches> [split]
lib=$1; m=$2; sp=${3:-0}
python tools/stress_parity.py 150 > /tmp/rm_b.log 2>&1 &
python tools/stress_parity.py 150 > /tmp/rm_c.log 2>&1 &
sleep 20
CER_MVS_LIB=$lib python tools/archive/repro_lookup_kernel.py $m $sp > /tmp/rm_a.log 2>&1
wait
grep -v amdgpu.ids /tmp/rm_a.log | tail -8; tail -1 /tmp/rm_b.log; tail -1 /tmp/rm_c.log
