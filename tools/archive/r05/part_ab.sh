#!/bin/bash
tools/archive/r05/ab_bench.sh "shared:CER_PIPE_PARTITION=0" "partitioned:" 2>&1 | grep "streams=3" | cut -c1-64
for s in 2 4; do python bench.py --no-cpu-baseline --streams $s --steps 18 --warmup 5 2>/dev/null | grep "^{" > /tmp/b.json; python -c "
import json
d = json.load(open('/tmp/b.json')); print('partitioned streams', d['config']['depth_maps_in_flight'], 'maps/s %.2f' % d['value'], 'ms %.3f' % d['ms_per_step'], d.get('parity', {}).get('rel_l1_disparity_vs_reference_capture'))"; done
