#!/bin/bash
# the tests of round 4's opt-in kernel forms (skipped on the product library) against the variant library that carries them
make -C cer-mvs_amd/csrc variants/libcermvs_optin.so > /dev/null 2>&1
export CER_MVS_LIB=$PWD/cer-mvs_amd/csrc/variants/libcermvs_optin.so
python -m pytest tests/test_host_cpu.py -q -k "exports or variant or argument_errors" 2>&1 | tail -2
python -m pytest tests/test_conv_s16_gpu.py tests/test_hip_parity.py -q -m gpu -k "producer_consumer or cost_lines_matches_walk" 2>&1 | tail -3
