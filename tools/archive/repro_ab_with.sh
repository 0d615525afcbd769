#!/bin/bash
# one A/B harness process (experimental library) + a companion: usage tools/archive/repro_ab_with.sh <forwards> <companion command...>
m=$1; shift
"$@" > /tmp/companion.log 2>&1 &
sleep 8
CER_MVS_LIB=${LIB:-cer-mvs_amd/csrc/variants/libcermvs_lkspec.so} python tools/archive/repro_lookup_ab.py $m solo 2>&1 | grep -v amdgpu.ids | grep "forwards:"
wait
tail -1 /tmp/companion.log
