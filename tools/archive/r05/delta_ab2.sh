#!/bin/bash
out=gpurun_out/r05f; mkdir -p $out
timeout 900 python -m pytest tests/test_conv_s16_gpu.py tests/test_hip_parity.py tests/test_determinism_gpu.py -x -q -k "conv_s16 or delta or end_to_end or update_block or determin or slab" 2>&1 | tail -6 | tee $out/tests.txt
tools/archive/r05/ab_bench.sh "split:CER_DELTA_MERGED=0" "merged:"
