// Shared pieces of the update block's s16 convolutions: csrc/conv_s16.hip (round 2/3: every wave runs every phase of a tile) and
// csrc/experimental/conv_s16pc.hip (round 4: producer / consumer wave roles; variant library only).  See conv_s16.hip for the arithmetic and the layouts.
#pragma once
#include "common.hpp"
#include <math.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx8 __attribute__((ext_vector_type(8)));

#define SX_TW 16                           // tile width in pixels
#define SX_HW 18                           // halo columns
#define SX_PITCH 20                        // LDS pixels per halo row (pitch % 4 == 0 keeps (q % 4) == (col % 4))
#define SX_ROWB (SX_PITCH * 64)            // LDS bytes per halo row
#define SX_ROWB8 (SX_PITCH * 128)          // ... of a 32-channel chunk in the fp8-correction form
#define SX_DTW 24                          // disparity tile columns (halo + 3 each side)
#define SX_EPI_DELTA 4
#define SX_HID_LOG2 4                      // scale of the delta head's hidden activations (relu outputs)
// Wait states (and a compiler memory barrier) behind the 16-byte LDS stores of the disparity generators: an EMPIRICAL margin of round 3.
// One generator variant produced intermittently wrong last tile rows and became clean with them; two candidate hardware hazards were
// excluded by micro-tests then.  Round 4 found what fails on this chip - packed-fp32 instructions that take their low result from src1's
// high half, next to f16 MFMA waves (tools/check_isa.py) - and the statement most likely only changed what hipcc emitted
// around it; the variant is gone and cannot be re-checked.  The form that ships contains no such instruction with or without the
// statement (-DSX_NO_STORE_WAIT: same scan result) and never failed either way.
#ifndef SX_NO_STORE_WAIT
#define SX_LDS_STORE_WAIT() asm volatile("s_nop 3" ::: "memory")
#else
#define SX_LDS_STORE_WAIT() do { } while (0)
#endif
#ifndef SX_OCC3
#define SX_OCC3 0    // 1: the 64-channel fp8-correction kernels aim at three blocks per CU (167 VGPRs, 8 spilled; fits with 8-row tiles, tile_mt 2:
                     // 52.7 KB of LDS per block).  Measured at 296 x 400: q 70.3 us against 65.2 (default, 12-row tiles, two blocks), corr2 32.1 / 32.3 - off
#endif
#ifndef SX_WRING3
#define SX_WRING3 1  // the 64-output-channel fp8-correction kernels request their weight slices two chunk steps ahead (ring of three register slots)
#endif
#ifndef SX_WSPLIT
#define SX_WSPLIT 0  // 1: the fp8- / FP6-correction kernels with two weight slots refill a slot part by part behind the MFMA group that used the part
                     // (every part gets 1 2/3 taps between request and use instead of 1 .. 1 2/3).  Round 6, same-box A/B (profiles/r06_wsplit_ab.txt):
                     // z|r stand-alone 98.4-99.4 against 101.8-102.7 us in the fp8 form but 95.6 against 92.6 in the FP6 form, delta 78.7 against
                     // 77.0; inside the forward 95.5-96.1 against 96.2-96.4 us and the same depth maps per second - and 16 spilled registers
                     // (outside the chunk loop) where the default has none.  Off.
#endif
#ifndef SX_STAGGER
#define SX_STAGGER 4       // conv_s16.hip: delay of a CU's second workgroup in the first round of blocks, in 64-cycle units per 16-channel step of the launch (0: off)
#endif
#ifndef SX_STAGGER_BIT
#define SX_STAGGER_BIT 16  // HW_ID bit that tells the two workgroups of a CU apart: 16 = thread-group id parity (0 = wave slot parity: measured equal)
#endif
#ifndef SX_STAGGER_WM1
#define SX_STAGGER_WM1 0   // 1: only the 128-output-channel launches (1 x 4 wave layout: z|r gates, delta head)
#endif
#ifndef SX_STAGGER_CUS
#define SX_STAGGER_CUS 256
#endif
#ifndef SX_TRACE
#define SX_TRACE 0   // variant builds only (tools/trace_s16.py): per wave cycle stamps + HW_ID written to `aux2` (GATES: unused there)
#endif

struct S16Args {
    const char* src[CER_CONV_MAX_SRC];     // tensors (frag16) first, the disparity source (fp32 [P]) last
    int ch[CER_CONV_MAX_SRC];
    int kind[CER_CONV_MAX_SRC];            // 2 = frag16 tensor, 1 = disparity
    int nsrc;
    const _Float16* wpk;                   // literal packing
    const _Float16* wpk_c;                 // collapsed packing (interior tiles) or null
    const float* bias;
    const float* init;                     // acc32 layout
    float* out;
    float* out2;
    const float* aux;
    const float* aux2;
    const void* edge;                      // rim-correction filters of the collapsed disparity form (cer_conv3x3_s16_edge_pack) or null
    int h, w, cout, tiles_x, ntiles, ny, mtx, mty;
    int border_first;                      // tiles on the image rim are given to the first blocks launched (see the kernel)
    float S, invS;                         // accumulator = S * conv
    float out_scale;                       // frag16 outputs
    float aux_inv;                         // 1 / scale of the frag16 hidden state read by GATES / GRU
    float disp_scale;                      // generated disparity features
    float proj_inv;                        // DELTA: 1 / (hidden scale * w2 scale)
    int out_split;
    int* flag;                             // sticky overflow flag (cer_overflow_flag) or null
};

// ---- operand split: 8 fp32 -> hi | lo halves of v * scale (packed conversions)
__device__ __forceinline__ void sx_split8(const float (&v)[8], float scale, half8& hi, half8& lo) {
    cer_h2 h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cer_f2 x = (cer_f2){v[2 * i], v[2 * i + 1]} * scale;
        x = __builtin_elementwise_min(__builtin_elementwise_max(x, (cer_f2){-65504.0f, -65504.0f}), (cer_f2){65504.0f, 65504.0f});
        h[i] = __builtin_convertvector(x, cer_h2);
        l[i] = __builtin_convertvector(x - __builtin_convertvector(h[i], cer_f2), cer_h2);
    }
    hi = (half8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
    lo = (half8){l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
}
__device__ __forceinline__ void sx_join8(const half8 hi, const half8 lo, float inv, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((float)hi[e] + (float)lo[e]) * inv;
}
// gate non-linearities on the hardware exp2 / rcp (1 ulp each; ~5 instructions instead of ~25 for expf + IEEE division: the
// epilogue was VALU-bound).  Absolute error <= 3e-7 - the class of the fp32 accumulation that feeds them.
__device__ __forceinline__ float sx_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float sx_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }


#ifndef CER_WITH_SXPC
#define CER_WITH_SXPC 0
#endif
// round 4 (experimental/conv_s16pc.hip, variants/libcermvs_optin.so only): the producer / consumer form of the fp8-correction convolutions of
// the GRU loop.  Returns CER_ESHAPE when the launch is not one it serves (the caller then takes the kernels of conv_s16.hip).
int sxpc_dispatch(S16Args& a, int epi, int tile_mt, hipStream_t st);
