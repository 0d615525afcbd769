#!/bin/bash
# Round-6 PMC passes (one counter set per rocprofv3 pass, no tracing flags).  (1) every kernel launched alone at the bench shapes, as in round 5
# (+ the FP6-correction z|r kernel, + the cost volume's band pass); (2) VERDICT r5 item 6: the same counters over tools/forward_loop.py with THREE
# depth maps in flight - the regime that ships - per kernel name, so that the solo figures can be held against it (rocprofv3 may serialise
# kernels while it collects counters: the per-launch GRBM_GUI_ACTIVE of (2) against (1) says whether it did).  Run on the GPU box: gpurun -- tools/pmc_r06.sh
out=${1:-gpurun_out/r06/pmc}
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
  run conv3x3_gates_zr_f8 "conv3x3_s16_kernel<1, 4, 4, 2, 1>" python tools/bench_conv_s16.py --f8 --only "z|r" --rounds 1 --reps 1
  run conv3x3_gates_zr_f6 "conv3x3_s16_kernel<1, 4, 4, 2, 2>" python tools/r06/bench_conv_forms.py --rounds 2 --reps 4
  run conv3x3_gru_q_f8 "conv3x3_s16_kernel<2, 2, ., 3, 1>" python tools/bench_conv_s16.py --f8 --only "gru" --rounds 1 --reps 1
  run conv3x3_delta_f8 "conv3x3_s16_kernel<1, 4, 4, 4, 1>" python tools/bench_conv_s16.py --f8 --only "delta" --rounds 1 --reps 1
  run conv3x3_corr2_f8 "conv3x3_s16_kernel<2, 2, ., 1, 1>" python tools/bench_conv_s16.py --f8 --only "corr2" --rounds 1 --reps 1
  run cost_lines_kernel "cost_lines_kernel<3, true>" python tools/prof_build.py
  run cost_lines_bands_kernel "cost_lines_bands_kernel" python tools/prof_build.py
  run cost_lines_kernel_x2 "cost_lines_kernel<3, false>" env CER_COST_X2=1 python tools/prof_build.py
  run lookup_encode "lookup_encode" python tools/prof_conv.py lookup --reps 1
  run enc_pc_32to32 "enc_pc_kernel<32, 32, 1, 9, 0, false>" python tools/bench_pc.py 32
  run enc_pc_32to32_f6 "enc_pc_kernel<32, 32, 1, 9, 0, false, true>" env CER_ENC_F6=1 python tools/bench_pc.py 32
  run enc_pc_32to32_dual_f6 "enc_pc_kernel<32, 32, 1, 9, 0, true, true>" env CER_ENC_F6=1 python tools/bench_pc.py 32 dual
  # (2) three depth maps in flight, whole forwards
  for s in 1 3; do
    run inflight${s}_gates_zr_f8 "conv3x3_s16_kernel<1, 4, 4, 2, 1>" python tools/forward_loop.py --streams $s --forwards 6
    run inflight${s}_lookup "lookup_encode" python tools/forward_loop.py --streams $s --forwards 6
    run inflight${s}_cost_lines "cost_lines_kernel" python tools/forward_loop.py --streams $s --forwards 6
  done
} | tee "$out/counters.txt"
python tools/pmc_summary.py "$out/counters.txt" "$out/pmc_traffic.json" > /dev/null
find "$out" -name "*.csv" -size +1M -delete
