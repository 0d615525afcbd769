#!/bin/bash
# usage: tools/archive/repro_lookup_multi.sh <lib> <processes> <launches> [split]
lib=$1; np=$2; m=$3; sp=${4:-0}
for i in $(seq 1 $np); do CER_MVS_LIB=$lib python tools/archive/repro_lookup_kernel.py $m $sp > /tmp/rk_$i.log 2>&1 & done
wait
for i in $(seq 1 $np); do grep -v amdgpu.ids /tmp/rk_$i.log | tail -8; done
