#!/bin/bash
out=gpurun_out/r05f; mkdir -p $out
timeout 900 python -m pytest tests/test_conv_s16_gpu.py tests/test_hip_parity.py -x -q -k "conv_s16 or delta or end_to_end_cfg1 or end_to_end_tiny or update_block" 2>&1 | tail -6 | tee $out/tests.txt
for rep in 1 2; do for m in 0 1; do echo "== CER_DELTA_MERGED=$m"; CER_DELTA_MERGED=$m timeout 300 python tools/bench_conv_s16.py --f8 --rounds 3 --reps 10 --only "delta" 2>&1 | grep -v amdgpu.ids | cut -c1-64; done; done | tee $out/bench.txt
