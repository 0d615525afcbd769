// Shared device helpers for libcermvs (gfx950 / CDNA4 only: wave64, DPP row ops, MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cer_mvs.h"

#define CER_RETURN_IF_LAUNCH_FAILED()                  \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)

// ---- split-f16 operand preparation, two elements at a time -------------------------------------------------------
// x = hi + 2^-11 * lo with hi = f16(x), lo = f16((x - hi) * 2^11) (gru_f16x3.hip).  Written on 2-vectors so that hipcc emits
// v_pk_add_f32 / v_pk_mul_f32 / v_cvt_pk_f16_f32: 5 instead of 12 VALU operations per element, bit-identical results.
typedef float cer_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 cer_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 cer_h8 __attribute__((ext_vector_type(8)));
#if defined(__HIPCC__)
__device__ __forceinline__ void cer_split2(cer_f2 x, cer_h2& hi, cer_h2& lo) {
    x = __builtin_elementwise_min(__builtin_elementwise_max(x, (cer_f2){-65504.0f, -65504.0f}), (cer_f2){65504.0f, 65504.0f});
    hi = __builtin_convertvector(x, cer_h2);
    lo = __builtin_convertvector((x - __builtin_convertvector(hi, cer_f2)) * 2048.0f, cer_h2);
}
// eight fp32 values -> (hi, lo) half8 vectors
__device__ __forceinline__ void cer_split8(const float (&v)[8], cer_h8& hi, cer_h8& lo) {
    cer_h2 h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cer_split2((cer_f2){v[2 * i], v[2 * i + 1]}, h[i], l[i]);
    hi = (cer_h8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
    lo = (cer_h8){l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
}
#endif

// ---- schedule fuzzing (variant build -DCER_FUZZ=1, make variants/libcermvs_fuzz.so): a per-wave pseudo-random
// s_sleep (0 .. 15 x 512 cycles) at every marked point - behind every barrier and in front of every LDS write phase of the kernels
// that have shown timing-dependent output in some variant (conv3x3_s16_kernel, lookup_encode_kernel).  It shifts the waves of a block
// against each other by up to several thousand cycles, far beyond what co-resident blocks or a second process do: a missing
// happens-before between one wave's ds_read and another's ds_write then shows as run-to-run nondeterminism with a reproducer
// (tests/test_determinism_gpu.py under CER_MVS_LIB=.../libcermvs_fuzz.so).  In the product build the macros expand to nothing.
#ifndef CER_FUZZ
#define CER_FUZZ 0
#endif
#if CER_FUZZ && defined(__HIPCC__)
#define CER_FUZZ_INIT() unsigned cer_fuzz_state = __builtin_amdgcn_readfirstlane((unsigned)((blockIdx.x * 8u + (threadIdx.x >> 6)) * 2654435761u) ^ (unsigned)__builtin_readcyclecounter())
#define CER_FUZZ_POINT() do { cer_fuzz_state = cer_fuzz_state * 1664525u + 1013904223u;                                   \
                              const unsigned n_ = __builtin_amdgcn_readfirstlane((cer_fuzz_state >> 24) & 15u);          \
                              for (unsigned i_ = 0; i_ < n_; ++i_) __builtin_amdgcn_s_sleep(8); } while (0)
#else
#define CER_FUZZ_INIT() do { } while (0)
#define CER_FUZZ_POINT() do { } while (0)
#endif

// process-wide sticky overflow flag (cer_overflow_flag in capi.hip): a device int that saturating conversions or into, or null
int* cer_overflow_flag_get();
// CUs of the current device (capi.hip; cached per device)
int cer_num_cus();
// LDS bytes per CU of the current device (capi.hip; cached per device)
int cer_lds_per_cu();

static inline bool cer_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- cross-lane sum over one 16-lane DPP row (4 v_add_f32_dpp, no LDS) -------------------
template <int CTRL>
__device__ __forceinline__ float cer_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// after the call every lane of the row holds the row total
__device__ __forceinline__ float cer_row16_sum(float v) {
    v += cer_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    v += cer_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    v += cer_dpp<0x141>(v);   // row_half_mirror
    v += cer_dpp<0x140>(v);   // row_mirror
    return v;
}

__device__ __forceinline__ float4 cer_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float cer_dot4(float4 a, float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}

// Bilinear sample of an NHWC map at (x, y) dotted with this lane's channel quads of f1.
// 16 lanes cooperate on one sample: lane `sub` owns channels 4*sub + 64*q .. +3, q < NQ, so one
// wave instruction fetches 4 whole texels (4 x 256 B) with 16-B loads.  Texels outside the map
// read as zero and a non-finite coordinate samples nothing (the reference would propagate NaN
// through 0*NaN, correlation_kernel.cu:97-100; see DESIGN.md "deliberate deviations").
// Returns the lane-partial dot (reduce with cer_row16_sum).  `tx0 = floor(x)`, `ty0 = floor(y)`.
template <int NQ>
__device__ __forceinline__ float cer_bilerp_dot(const float* __restrict__ f2, int H2, int W2, int C, int sub,
                                                float fx, float fy, float dx, float dy, const float4 (&f1q)[NQ]) {
    // everything outside [-1, W2] x [-1, H2] has all four corners out of bounds
    const bool any = (fx >= -1.0f) && (fx <= (float)W2) && (fy >= -1.0f) && (fy <= (float)H2);
    if (!any) return 0.0f;
    const int ix = (int)fx, iy = (int)fy;
    const bool x0ok = ix >= 0 && ix < W2, x1ok = ix + 1 >= 0 && ix + 1 < W2;
    const bool y0ok = iy >= 0 && iy < H2, y1ok = iy + 1 >= 0 && iy + 1 < H2;
    const float* base = f2 + ((long)iy * W2 + ix) * C + 4 * sub;
    float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float* bq = base + 64 * q;
        if (y0ok && x0ok) s00 = cer_dot4(f1q[q], cer_ld4(bq), s00);
        if (y0ok && x1ok) s01 = cer_dot4(f1q[q], cer_ld4(bq + C), s01);
        if (y1ok && x0ok) s10 = cer_dot4(f1q[q], cer_ld4(bq + (long)W2 * C), s10);
        if (y1ok && x1ok) s11 = cer_dot4(f1q[q], cer_ld4(bq + (long)W2 * C + C), s11);
    }
    const float wx1 = dx, wx0 = 1.0f - dx, wy1 = dy, wy0 = 1.0f - dy;
    return s00 * wy0 * wx0 + s01 * wy0 * wx1 + s10 * wy1 * wx0 + s11 * wy1 * wx1;
}
