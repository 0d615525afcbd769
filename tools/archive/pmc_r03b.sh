#!/bin/bash
# Round-3 PMC passes for the z|r gate convolution in the fp8-correction form (the default since gru_precision="s16f8"), launched
# alone at the bench shapes; one counter set per rocprofv3 pass, no tracing flags.  Run on the GPU box:
#   gpurun -- tools/archive/pmc_r03b.sh gpurun_out/pmc_r03b ; then append counters.txt to profiles/r03_pmc_counters.txt and re-run tools/pmc_summary.py
out=${1:-gpurun_out/pmc_r03b}
mkdir -p "$out"
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"
      "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA")
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_')
  tools/pmc.sh "$out/conv3x3_gates_zr_f8/$tag" "conv3x3_s16_kernel<1, 4, 4, 2, 1>" "$set" -- python tools/bench_conv_s16.py --f8 --only "z|r" --rounds 1 --reps 1 | sed "s/^/conv3x3_gates_zr_f8 /"
done | tee "$out/counters.txt"
