#!/bin/bash
# term-split (KS) form of the 64-channel convs against the round-3 form: kernel tests, then timings in both forms
out=gpurun_out/r05b; mkdir -p $out
timeout 900 python -m pytest tests/test_conv_s16_gpu.py -x -q 2>&1 | tail -5 > $out/test_conv.txt
for ks in 0 1 0 1; do echo "== CER_S16_KS=$ks"; CER_S16_KS=$ks timeout 300 python tools/bench_conv_s16.py --f8 --rounds 3 --reps 10 --only "corr2,q gru" 2>&1 | grep -v amdgpu.ids; done > $out/bench_conv.txt
cat $out/test_conv.txt $out/bench_conv.txt
