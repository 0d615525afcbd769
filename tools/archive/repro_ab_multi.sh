#!/bin/bash
# usage: tools/archive/repro_ab_multi.sh <processes> <forwards> [env assignments...]
np=$1; m=$2; shift 2
for i in $(seq 1 $np); do env "$@" CER_MVS_LIB=${LIB:-cer-mvs_amd/csrc/variants/libcermvs_lkspec.so} python tools/archive/repro_lookup_ab.py $m p$i > /tmp/ab_$i.log 2>&1 & done
wait
for i in $(seq 1 $np); do grep -v amdgpu.ids /tmp/ab_$i.log | tail -7; done
