#!/bin/bash
# HBM traffic of single kernels from rocprofv3 PMC counters, one counter set per pass (FETCH_SIZE and WRITE_SIZE
# do not fit one pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Run on the GPU box:
#   gpurun -- tools/archive/pmc_traffic.sh gpurun_out/pmc_r01
out=${1:-gpurun_out/pmc_traffic}
mkdir -p "$out"
run() {  # name regex what
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
    tag=$(echo $set | tr ' ' '_')
    tools/pmc.sh "$out/$1/$tag" "$2" "$set" -- python tools/prof_conv.py $3 --reps 1 | sed "s/^/$1 /"
  done
}
run conv_zr "conv3x3_f16x3" zr
run conv_q "conv3x3_f16x3" q
run conv_delta "conv3x3_f16x3" d1f
run lookup_encode "lookup_encode" lookup
run cost_build0 "cost_build" build0
run cost_build1 "cost_build" build1
