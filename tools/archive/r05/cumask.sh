#!/bin/bash
CER_MVS_LIB=$PWD/cer-mvs_amd/csrc/variants/libcermvs_sxtrace.so timeout 300 python tools/archive/r05/cumask_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
