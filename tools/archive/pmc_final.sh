#!/bin/bash
# Final round-1 PMC evidence (run on the GPU box): MFMA / VALU / LDS activity of the dominant kernels, one counter set per pass.
out=${1:-gpurun_out/pmc_final}
mkdir -p "$out"
run() {  # name regex command...
  name=$1; re=$2; shift 2
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $set | tr ' ' '_')
    tools/pmc.sh "$out/$name/$tag" "$re" "$set" -- "$@" | sed "s/^/$name /"
  done
}
run conv_zr "conv3x3_f16x3_kernel<4, 2, 1, 2, 3, 4, 2>" python tools/prof_conv.py zr --reps 1
run enc_stream "enc_conv32_stream" python tools/archive/prof_enc.py
run geo_fusion "geo_consistency" python tools/bench_fusion.py
