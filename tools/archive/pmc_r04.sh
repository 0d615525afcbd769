#!/bin/bash
# Round-4 PMC passes (one counter set per rocprofv3 pass, no tracing flags): the encoder's producer / consumer kernels (32 -> 32
# single and dual-source, 64 -> 64, the stem) launched alone at the bench shapes (tools/bench_pc.py, tools/archive/prof_enc.py), the z|r gate conv
# and the lookup.  Run on the GPU box: gpurun -- tools/archive/pmc_r04.sh gpurun_out/r04/pmc ; summary: tools/pmc_summary.py <dir>/counters.txt out.json
out=${1:-gpurun_out/r04/pmc}
mkdir -p "$out"
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"
      "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA")
run() {   # name, kernel regex, command...
  local name=$1 re=$2; shift 2
  for set in "${SETS[@]}"; do
    tag=$(echo $set | tr ' ' '_')
    tools/pmc.sh "$out/$name/$tag" "$re" "$set" -- "$@" | sed "s/^/$name /"
  done
}
{
  if [ -z "$PMC_ONLY_NEW" ]; then
  run enc_pc_32to32 "enc_pc_kernel<32, 32, 1, 9, 0, false>" python tools/bench_pc.py 32
  run enc_pc_32to32_dual "enc_pc_kernel<32, 32, 1, 9, 0, true>" python tools/bench_pc.py 32 dual
  run enc_pc_64to64 "enc_pc_kernel<64, 64, 1, 9, 0, false>" python tools/bench_pc.py 64
  run enc_pc_32to64_s2_dual "enc_pc_kernel<32, 64, 2, 9, 0, true>" python tools/bench_pc.py s2 dual
  run enc_stem_pc "enc_stem_pc_kernel" python tools/archive/prof_enc.py pc fnet
  run conv3x3_gates_zr_f8 "conv3x3_s16_kernel<1, 4, 4, 2, 1>" python tools/bench_conv_s16.py --f8 --only "z|r" --rounds 1 --reps 1
  fi
  run conv3x3_gru_q_f8 "conv3x3_s16_kernel<2, 2, ., 3, 1>" python tools/bench_conv_s16.py --f8 --only "gru" --rounds 1 --reps 1
  run conv3x3_delta_f8 "conv3x3_s16_kernel<1, 4, 4, 4, 1>" python tools/bench_conv_s16.py --f8 --only "delta" --rounds 1 --reps 1
  run conv3x3_corr2_f8 "conv3x3_s16_kernel<2, 2, ., 1, 1>" python tools/bench_conv_s16.py --f8 --only "corr2" --rounds 1 --reps 1
  run cost_lines_kernel "cost_lines_kernel" python tools/prof_build.py
  run lookup_encode "lookup_encode" python tools/prof_conv.py lookup --reps 1
} | tee "$out/counters.txt"
python tools/pmc_summary.py "$out/counters.txt" "$out/pmc_traffic.json" > /dev/null
find "$out" -name "*.csv" -size +1M -delete
