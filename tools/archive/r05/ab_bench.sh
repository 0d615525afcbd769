#!/bin/bash
# same-box A/B of whole-forward throughput: each argument is "label:ENV=val,ENV=val[,LIB=variantname]"; every configuration runs one at a time and
# with three depth maps in flight, the whole set twice (interleaved)
out=gpurun_out/r05ab; mkdir -p $out; : > $out/ab.txt
V=$PWD/cer-mvs_amd/csrc/variants
for rep in 1 2; do
  for cfg in "$@"; do
    label=${cfg%%:*}; envs=${cfg#*:}
    cmd="env"
    IFS=',' read -ra kv <<< "$envs"
    for e in "${kv[@]}"; do
      if [[ $e == LIB=* ]]; then cmd="$cmd CER_MVS_LIB=$V/libcermvs_${e#LIB=}.so"; elif [ -n "$e" ]; then cmd="$cmd $e"; fi
    done
    for s in 1 3; do
      $cmd timeout 600 python bench.py --no-cpu-baseline --streams $s 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('$label', 'streams=$s', 'maps/s=%.2f' % d['value'], 'ms=%.3f' % d['ms_per_step'], {n.replace('conv3x3_',''): round(v.get('avg_us', 0), 1) for n, v in k.items() if any(t in n for t in ('lookup', 'delta_sum', 'cost_build', 'gates', 'gru_q', 'relu_64', 'delta_fused'))})" | tee -a $out/ab.txt
    done
  done
done
