#!/bin/bash
# Round-3 PMC evidence for the cost-volume kernel (csrc/cost_lines.hip), one counter set per rocprofv3 pass (no tracing flags), run
# on the GPU box:   gpurun -- tools/archive/pmc_r03.sh gpurun_out/pmc_r03     then   python tools/pmc_summary.py gpurun_out/pmc_r03 profiles/r03_pmc_traffic.json
# The kernel runs on the bench scene (tools/prof_build.py: encoded features, true epipolar geometry), both stages.
out=${1:-gpurun_out/pmc_r03}
mkdir -p "$out"
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"
      "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA")
for re in "cost_lines_kernel" "cost_lines_reduce_kernel" "feat_split_kernel"; do
  for set in "${SETS[@]}"; do
    tag=$(echo $set | tr ' ' '_')
    tools/pmc.sh "$out/$re/$tag" "$re" "$set" -- python tools/prof_build.py | sed "s/^/$re /"
  done
done | tee "$out/counters.txt"
# the roofline kernel (z|r gate convolution, unchanged since round 2) and the lookup, launched alone at the bench shapes
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_')
  tools/pmc.sh "$out/conv3x3_gates_zr/$tag" "conv3x3_s16_kernel<1, 4, 4, 2>" "$set" -- python tools/bench_conv_s16.py --only "z|r" --rounds 1 --reps 1 | sed "s/^/conv3x3_gates_zr /"
  tools/pmc.sh "$out/lookup_encode/$tag" "lookup_encode" "$set" -- python tools/prof_conv.py lookup --reps 1 | sed "s/^/lookup_encode /"
done | tee -a "$out/counters.txt"
