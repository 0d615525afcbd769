#!/bin/bash
out=gpurun_out/r05c; mkdir -p $out
V=$PWD/cer-mvs_amd/csrc/variants
CER_MVS_LIB=$V/libcermvs_ksprobe.so timeout 300 python tools/trace_s16.py --f8 --conv q --mt 2 2>&1 | grep -v "amdgpu.ids\|^  File\|^    " > $out/trace_q_ks.txt
CER_S16_KS=0 CER_MVS_LIB=$V/libcermvs_sxtrace.so timeout 300 python tools/trace_s16.py --f8 --conv q --mt 2 2>&1 | grep -v "amdgpu.ids\|^  File\|^    "> $out/trace_q_base.txt
head -20 $out/trace_q_ks.txt; echo; head -20 $out/trace_q_base.txt
