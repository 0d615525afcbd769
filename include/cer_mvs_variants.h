/*
 * cer_mvs_variants.h - entry points that exist ONLY in cer-mvs_amd/csrc/variants/libcermvs_optin.so
 * (make -C cer-mvs_amd/csrc variants/libcermvs_optin.so; loaded with CER_MVS_LIB=...), on top of everything in cer_mvs.h.
 *
 * Round 4 built two alternative kernel forms that are correct, tested against the default forms and measured SLOWER at the
 * bench workload.  Since round 5 they are not linked into the product library; the variant library keeps
 * them buildable for the tests (tests/test_conv_s16_gpu.py, tests/test_hip_parity.py: skipped unless this library is loaded)
 * and the measurement tools (tools/archive/trace_sxpc.py, tools/archive/bench_cost_lines.py, tools/archive/stats_cost_lines.py).
 */
#ifndef CER_MVS_VARIANTS_H
#define CER_MVS_VARIANTS_H
#include "cer_mvs.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Producer / consumer form of the GRU loop's fp8-correction convolutions (csrc/experimental/conv_s16pc.hip, round 4; replaces the
 * launches of cer_conv3x3_s16 with CER_EPI_CORR_FP8 - reference: core/update.py:17-25,63-71): process-wide switch, 0 = off (default;
 * environment CER_S16_PC=1 turns it on at first use), 1 = on, anything else = query; returns the previous setting.  Same operands
 * and results (the hoisted `init` term is added in the epilogue instead of seeding the accumulators: last-bit differences). */
int cer_conv3x3_s16_pc(int on);

/* Which kernel builds the per-view partial volumes of cer_cost_lines_f32 (reference: core/corr.py:84-91): 0 (default) one line per
 * 256-thread block (round 3); 1 = several neighbouring lines of a segment per block sharing ONE band fetched through LDS (round 4:
 * 2.8 x fewer texel bytes through the CU's vector-memory path, more vector instructions per sample - measured no faster; lines whose
 * bands do not fit its window are listed in the workspace and finished by the one-line kernel in the same call).  Same cells,
 * weights and dot products: the two forms agree bit for bit.  CER_COST_LINES_FORM in the environment sets the initial value
 * (CER_COST_LINES_NW, CER_COST_LINES_OUTD pick the variant); returns the previous setting, < 0 only queries. */
int cer_cost_lines_form(int form);
/* Diagnostics: 32 counters of the multi-line kernel (all zero unless built with -DC8_STATS=1) to host memory. */
int cer_cost_lines_stats(unsigned long long* out, int reset);

#ifdef __cplusplus
}
#endif
#endif
