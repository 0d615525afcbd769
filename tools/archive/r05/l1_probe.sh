#!/bin/bash
out=gpurun_out/r05j; mkdir -p $out
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  tag=$(echo $set | tr ' ' '_')
  tools/pmc.sh "$out/$tag" "cost_lines_kernel" "$set" -- python tools/prof_build.py | sed "s/^/cost_lines /"
done
