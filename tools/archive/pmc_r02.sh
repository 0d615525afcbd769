#!/bin/bash
# Round-2 PMC evidence for the shipped kernels, one counter set per rocprofv3 pass (no tracing flags), run on the GPU box:
#   gpurun -- tools/archive/pmc_r02.sh gpurun_out/pmc_r02        then  python tools/pmc_summary.py gpurun_out/pmc_r02 profiles/r02_pmc_traffic.json
# Kernels are launched alone at cfg2 shapes (296 x 400 features, 10 views) on random data.
out=${1:-gpurun_out/pmc_r02}
mkdir -p "$out"
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"
      "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
run() {  # name regex command...
  name=$1; re=$2; shift 2
  for set in "${SETS[@]}"; do
    tag=$(echo $set | tr ' ' '_')
    tools/pmc.sh "$out/$name/$tag" "$re" "$set" -- "$@" | sed "s/^/$name /"
  done
}
run conv3x3_gates_zr    "conv3x3_s16_kernel<1, 4, 4, 2>" python tools/bench_conv_s16.py --only "z|r" --rounds 1 --reps 1
run conv3x3_gru_q       "conv3x3_s16_kernel<2, 2, 4, 3>" python tools/bench_conv_s16.py --only "q gru" --rounds 1 --reps 1
run conv3x3_delta_fused "conv3x3_s16_kernel<1, 4, 4, 4>" python tools/bench_conv_s16.py --only "delta" --rounds 1 --reps 1
run conv3x3_relu_64     "conv3x3_s16_kernel<2, 2, 4, 1>" python tools/bench_conv_s16.py --only "corr2" --rounds 1 --reps 1
run lookup_encode       "lookup_encode"                  python tools/prof_conv.py lookup --reps 1
run enc_stem_s16        "enc_stem_s16"                   python tools/prof_conv.py stem --reps 1
run cost_build_stage0   "cost_build"                     python tools/prof_conv.py build0 --reps 1
run cost_build_stage1   "cost_build"                     python tools/prof_conv.py build1 --reps 1
