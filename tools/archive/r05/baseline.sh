#!/bin/bash
# round 5, first call: where the round starts on today's box (conv traces, conv A/B timings, bench lines)
out=gpurun_out/r05a; mkdir -p $out
V=$PWD/cer-mvs_amd/csrc/variants
for c in zr q; do CER_MVS_LIB=$V/libcermvs_sxtrace.so timeout 300 python tools/trace_s16.py --f8 --conv $c > $out/trace_$c.txt 2>&1; done
timeout 300 python tools/bench_conv_s16.py --f8 --rounds 3 --reps 10 > $out/bench_conv.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --streams 1 > $out/bench_s1.json 2> $out/bench_s1.err
timeout 600 python bench.py --no-cpu-baseline > $out/bench_s3.json 2> $out/bench_s3.err
tail -5 $out/trace_q.txt; tail -12 $out/bench_conv.txt
