#!/bin/bash
for rep in 1 2; do for s in 2 3 4; do python bench.py --no-cpu-baseline --streams $s --steps 18 --warmup 5 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('streams', $s, 'maps/s %.2f' % d['value'], 'ms %.3f' % d['ms_per_step'], 'one-at-a-time', round(d['one_at_a_time']['value'], 2), 'parity', d.get('parity'))"; done; done
