#!/bin/bash
# usage: tools/archive/r05/mkprobe.sh <name> <flags...>   -> cer-mvs_amd/csrc/variants/libcermvs_probe_<name>.so
# conv_s16.hip compiled with -DSX_PROBE=<0|1> (ONE kernel configuration: the 64-output-channel fp8-correction form, KS = the value) + the flags given,
# linked with the other objects of the product library.  Seconds instead of minutes; serves only the q / corr2 launches (tools/bench_conv_s16.py --f8 --only "corr2,q gru").
set -e
name=$1; shift
cd "$(dirname "$0")/../../../cer-mvs_amd/csrc"
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wall -Wno-unused-function -Wno-unused-variable "$@" -c conv_s16.hip -o variants/probe_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name.*conv3x3.*Li3ELi1E" | grep "VGPRs\|Scratch" | sed "s/.*remark: *//; s/ \[-R.*//" | tr '\n' ' '; echo
objs=$(ls *.o | grep -v "^conv_s16.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $objs variants/probe_$name.o -o variants/libcermvs_probe_$name.so
