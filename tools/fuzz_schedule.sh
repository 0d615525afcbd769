#!/bin/bash
# the determinism tests on the schedule-fuzz build (common.hpp CER_FUZZ: per-wave pseudo-random s_sleep of 0-7.7 k cycles
# behind every barrier and in front of every LDS write phase of conv3x3_s16_kernel and lookup_encode_kernel).  Any launch that differs
# from the first is a real race with a reproducer.  usage (GPU box): tools/fuzz_schedule.sh [launches per case, default 500]
n=${1:-500}
make -C cer-mvs_amd/csrc variants/libcermvs_fuzz.so > /dev/null 2>&1
CER_MVS_LIB=$PWD/cer-mvs_amd/csrc/variants/libcermvs_fuzz.so CER_DET_LAUNCHES=$n python -m pytest tests/test_determinism_gpu.py -q -k "update_block or lookup" 2>&1 | tail -5
