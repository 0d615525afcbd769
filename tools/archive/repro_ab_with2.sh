#!/bin/bash
# usage: tools/archive/repro_ab_with2.sh <harness processes> <forwards> <companions> <companion lib>
nh=$1; m=$2; nc=$3; clib=$4
for i in $(seq 1 $nc); do CER_MVS_LIB=$clib python tools/stress_parity.py 200 > /tmp/comp_$i.log 2>&1 & done
for i in $(seq 1 $nh); do CER_MVS_LIB=cer-mvs_amd/csrc/variants/libcermvs_lkspec.so python tools/archive/repro_lookup_ab.py $m h$i > /tmp/h_$i.log 2>&1 & done
wait
for i in $(seq 1 $nh); do grep "forwards:" /tmp/h_$i.log; done
for i in $(seq 1 $nc); do tail -1 /tmp/comp_$i.log; done
