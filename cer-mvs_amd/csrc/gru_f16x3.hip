// K3 (fast path): the update block's 3x3 convolutions as implicit GEMMs on the f16 matrix cores with
// fp32-equivalent accuracy ("f16x3" split) - reference: core/update.py:13-25,61-71,80-85,87-120.
//
// Why: v_mfma_f32_16x16x4_f32 (exact fp32) runs at the fp32 VECTOR rate (157 TF); v_mfma_f32_32x32x16_f16
// runs 16x faster.  Every fp32 operand is split into two halves,  x = x_hi + 2^-11 * x_lo'  with
// x_hi = f16(x), x_lo' = f16((x - x_hi) * 2^11)  (both exactly representable steps; |x - x_hi - 2^-11 x_lo'|
// <= 2^-22 |x|), and   x*w ~= x_hi*w_hi + 2^-11 (x_hi*w_lo' + x_lo'*w_hi):  three f16 MFMAs whose products
// are exact in the fp32 accumulator (11b x 11b mantissas).  The dropped x_lo*w_lo term is 2^-22 relative, the
// same class as fp32 rounding, so results track the exact-fp32 kernel to ~1e-7 (measured: disparity rel-L1
// 1.4e-7 after 16 GRU iterations) at a net 16/3 = 5.3x the fp32-MFMA rate.  The two scaled partial sums
// live in separate fp32 accumulators and are combined once in the epilogue.
//
// Structure (per block: 8 waves = 4 pixel rows x 2 channel halves, 4 x 32 pixel tile x NB = 128 | 64 output channels, two
// blocks per CU):
//   * K loop over 32-channel chunks of the concatenated sources; per chunk the 6 x 34 halo is read once (fp32, coalesced),
//     split to hi|lo f16 (two elements per instruction: common.hpp) and stored to LDS with a 144-B pixel stride
//     (conflict-free ds_read_b128 A fragments for every tap);
//   * the disparity encoder's 49 channels are generated from an LDS disparity tile; tiles whose pixels all have their 3x3
//     neighbourhood inside the image evaluate that source as ONE 81-tap filter on the raw disparity (3 single-tap steps
//     instead of 18; weights pre-summed by cer_conv3x3_f16x3_pack_collapsed);
//   * weights are pre-split and pre-packed on the host in B-fragment order; per (chunk, tap) step the block's NB/8 KiB slice
//     is DMA'd global->LDS with global_load_lds_dwordx4 (lane-linear image == fragment order) through a 3-slot ring two
//     steps ahead of the multiply, with counted vmcnt waits; the wave index lives in an SGPR (readfirstlane) so that the
//     piece addresses are uniform + lane offset (no spills in the tap loop);
//   * per step and k16-step a wave reads its A hi/lo and B hi/lo fragments (ds_read_b128) and issues 3 MFMAs per 32x32
//     output tile; the 64-channel configuration skews the MFMA stream half a tap against the LDS reads;
//   * epilogue: the wave's tile goes through a private LDS transpose, so that the hoisted `init` term, the GRU operands and
//     the outputs move as 16-byte accesses; gate math as in the fp32 kernel; CER_EPI_DELTA instead projects the hidden tile
//     onto the nine taps of the 256->1 conv and writes only those planes.
// Compile-time experiment switches (default off; DESIGN.md roadmap): HX_ABL (phase ablations), HX_TRACE (cycle stamps),
// HX_CFG / HX_TH (other tilings), HX_EPI2=0 (direct 4-byte epilogue).
#include "common.hpp"
#include <stdlib.h>
#include <string.h>
#ifndef HX_ABL
#define HX_ABL 0      // profiling ablations, compile time: 1 no MFMA, 2 no barriers, 4 no staging, 8 no init/epilogue, 16 no weight DMA
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifndef HX_EPI2
#define HX_EPI2 1     // epilogue through LDS with 16-byte global accesses (0: the direct 4-byte form, kept for A/B runs)
#endif
#ifndef HX_TH
#define HX_TH 4                        // tile rows (8: experiment, one 8-wave block of 256 px per CU)
#endif
#define HX_TW 32
#define HX_HH (HX_TH + 2)
#define HX_HW (HX_TW + 2)
#define HX_ROWS (HX_HH * HX_HW)        // 204 halo pixels
#define HX_KC 32                       // channels per chunk
#define HX_AS 144                      // LDS bytes per halo pixel: 32 hi | 32 lo | 16 pad
#define HX_A_BYTES (HX_ROWS * HX_AS)   // 29376
#define HX_EPI_DELTA 4                 // conv + ReLU + projection onto the 9 taps of the 256->1 delta conv (cer_mvs.h: CER_EPI_DELTA)
#define HX_HS 272                      // LDS bytes per pixel row of the hidden tile (128 f16 + 16 pad: conflict-free b128 reads)

// HX_TRACE (debug builds only, tools/archive/trace_conv.py): per (block, wave, step) cycle stamps written to `aux2`, which the traced
// epilogue (GATES) does not use: [0] loop top, [1] after the barrier, [2] after the first k16-step, [3] end of the step; slot
// 63 of each wave holds (kernel entry, main loop start, main loop end, kernel end) and slot 62 the HW_ID register.
#ifndef HX_TRACE
#define HX_TRACE 0
#endif
#if HX_TRACE
#define HX_STAMP(k) do { if (HX_TRACE) trace_v[k] = __builtin_readcyclecounter(); } while (0)
#else
#define HX_STAMP(k) do { } while (0)
#endif

struct ConvArgsX {
    const float* src[CER_CONV_MAX_SRC];
    int ch[CER_CONV_MAX_SRC];
    int chpad[CER_CONV_MAX_SRC];
    int kind[CER_CONV_MAX_SRC];
    int nsrc;
    const _Float16* wpk;
    const _Float16* wpk_c;                 // collapsed-disparity packing for interior tiles (or null)
    const float* bias;
    const float* init;
    float* out;
    float* out2;
    const float* aux;
    const float* aux2;
    int h, w, cout;
    int tiles_x;
    int out_split;                         // RELU / LINEAR: out, GATES: out2 (r*h), GRU: out are written in the split32 layout
    int aux_split;                         // GATES / GRU: aux (the previous hidden state) is read in the split32 layout
};

__device__ __forceinline__ float hx_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void hx_split(float v, _Float16& hi, _Float16& lo) {
    const float x = fminf(fmaxf(v, -65504.0f), 65504.0f);
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * 2048.0f);
}

#define HX_DT_H (HX_HH + 6)            // disparity tile rows (halo + 3 each side for the 7x7 unfold)
#define HX_DT_W (HX_HW + 6)
#define HX_D_BYTES (HX_DT_H * HX_DT_W * 4)

__device__ __forceinline__ void hx_write8(char* __restrict__ ldsA, int row, int g, const float (&v)[8]) {
    half8 hi, lo;
    cer_split8(v, hi, lo);                                 // packed conversions (common.hpp)
    *reinterpret_cast<half8*>(ldsA + row * HX_AS + g * 16) = hi;
    *reinterpret_cast<half8*>(ldsA + row * HX_AS + 64 + g * 16) = lo;
}

// kind-0 chunk: every thread first issues ALL its loads (addresses clamped into the image so the loads are
// unconditional and independent), then splits fp32 -> hi|lo f16 and writes the LDS tile (zero padding here).
template <int NTHR>
__device__ __forceinline__ void hx_stage_tensor(char* __restrict__ ldsA, const ConvArgsX& a, int s, int c0, int ty0, int tx0) {
    constexpr int ITEMS = (HX_ROWS * 4 + NTHR - 1) / NTHR;
    float4 raw[ITEMS][2];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = min((int)threadIdx.x + NTHR * i, HX_ROWS * 4 - 1);
        const int row = idx >> 2, g = idx & 3;
        const int hy = row / HX_HW, hx = row - hy * HX_HW;
        const int gy = min(max(ty0 + hy - 1, 0), a.h - 1), gx = min(max(tx0 + hx - 1, 0), a.w - 1);
        const float* p = a.src[s] + ((long)gy * a.w + gx) * a.ch[s] + c0 + 8 * g;
        raw[i][0] = cer_ld4(p);
        raw[i][1] = cer_ld4(p + 4);
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = threadIdx.x + NTHR * i;
        if (idx < HX_ROWS * 4) {
            const int row = idx >> 2;
            const int hy = row / HX_HW, hx = row - hy * HX_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const float m = (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) ? 1.0f : 0.0f;
            const float v[8] = {raw[i][0].x * m, raw[i][0].y * m, raw[i][0].z * m, raw[i][0].w * m,
                                raw[i][1].x * m, raw[i][1].y * m, raw[i][1].z * m, raw[i][1].w * m};
            hx_write8(ldsA, row, idx & 3, v);
        }
    }
}

// kind-3 chunk: the tensor already holds hi|lo f16 pairs ("split32": per pixel and 32-channel chunk 128 bytes = 32 hi halves |
// 32 lo halves, written by a producer epilogue with out_split) - staging is two 16-byte loads and two 16-byte LDS writes per
// (pixel, 8-channel group), no conversion arithmetic.
template <int NTHR>
__device__ __forceinline__ void hx_stage_presplit(char* __restrict__ ldsA, const ConvArgsX& a, int s, int c0, int ty0, int tx0) {
    constexpr int ITEMS = (HX_ROWS * 4 + NTHR - 1) / NTHR;
    float4 raw[ITEMS][2];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = min((int)threadIdx.x + NTHR * i, HX_ROWS * 4 - 1);
        const int row = idx >> 2, g = idx & 3;
        const int hy = row / HX_HW, hx = row - hy * HX_HW;
        const int gy = min(max(ty0 + hy - 1, 0), a.h - 1), gx = min(max(tx0 + hx - 1, 0), a.w - 1);
        const float* p = a.src[s] + ((long)gy * a.w + gx) * a.ch[s] + c0;      // 32 floats = 128 bytes per (pixel, chunk)
        raw[i][0] = cer_ld4(p + 4 * g);                    // 8 hi halves
        raw[i][1] = cer_ld4(p + 16 + 4 * g);               // 8 lo halves
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = threadIdx.x + NTHR * i;
        if (idx < HX_ROWS * 4) {
            const int row = idx >> 2, g = idx & 3;
            const int hy = row / HX_HW, hx = row - hy * HX_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            // zero padding by a bit mask on the f16 pairs (a select between two float4 aggregates makes hipcc park both in scratch
            // and select a pointer)
            const unsigned m = (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) ? 0xFFFFFFFFu : 0u;
            uint4 hi4 = *reinterpret_cast<const uint4*>(&raw[i][0]), lo4 = *reinterpret_cast<const uint4*>(&raw[i][1]);
            hi4.x &= m; hi4.y &= m; hi4.z &= m; hi4.w &= m;
            lo4.x &= m; lo4.y &= m; lo4.z &= m; lo4.w &= m;
            *reinterpret_cast<uint4*>(ldsA + row * HX_AS + g * 16) = hi4;
            *reinterpret_cast<uint4*>(ldsA + row * HX_AS + 64 + g * 16) = lo4;
        }
    }
}

// 8 consecutive channels (first one co, a multiple of 8) of pixel `pix` in a split32 tensor with C channels
__device__ __forceinline__ void hx_store_split8(float* __restrict__ base, long pix, int C, int co, const float (&v)[8]) {
    half8 hi, lo;
    cer_split8(v, hi, lo);
    char* p = reinterpret_cast<char*>(base) + (pix * C + (co & ~31)) * 4 + (co & 31) * 2;
    *reinterpret_cast<half8*>(p) = hi;
    *reinterpret_cast<half8*>(p + 64) = lo;
}
__device__ __forceinline__ void hx_load_split8(const float* __restrict__ base, long pix, int C, int co, float (&v)[8]) {
    const char* p = reinterpret_cast<const char*>(base) + (pix * C + (co & ~31)) * 4 + (co & 31) * 2;
    const half8 hi = *reinterpret_cast<const half8*>(p), lo = *reinterpret_cast<const half8*>(p + 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf((float)lo[e], 1.0f / 2048.0f, (float)hi[e]);
}

// disparity tile (zero outside the image) -> LDS, once per block; feeds the on-the-fly disparity encoder
template <int NTHR>
__device__ __forceinline__ void hx_load_disp_tile(float* __restrict__ ldsD, const float* __restrict__ d, int h, int w, int ty0, int tx0) {
    for (int idx = threadIdx.x; idx < HX_DT_H * HX_DT_W; idx += NTHR) {
        const int r = idx / HX_DT_W, c = idx - r * HX_DT_W;
        const int gy = ty0 + r - 4, gx = tx0 + c - 4;
        ldsD[idx] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? d[(long)gy * w + gx] : 0.f;
    }
}

// kind-1 chunk: 100 * (unfold7x7(disp) - disp) (core/update.py:80-85,97) generated from the LDS disparity tile
template <int NTHR>
__device__ __forceinline__ void hx_stage_disp(char* __restrict__ ldsA, const float* __restrict__ ldsD, const ConvArgsX& a, int c0, int ty0,
                                              int tx0) {
    for (int idx = threadIdx.x; idx < HX_ROWS * 4; idx += NTHR) {
        const int row = idx >> 2, g = idx & 3;
        const int hy = row / HX_HW, hx = row - hy * HX_HW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool inside = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;      // the 3x3 conv zero-pads the FEATURE map
        const float ctr = ldsD[(hy + 3) * HX_DT_W + hx + 3];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + 8 * g + e;
            const int uy = c / 7, ux = c - uy * 7;
            v[e] = (inside && c < 49) ? 100.0f * (ldsD[(hy + uy) * HX_DT_W + hx + ux] - ctr) : 0.f;
        }
        hx_write8(ldsA, row, g, v);
    }
}

// collapsed disparity chunk (interior tiles only): channel s = (sy, sx) of the 9x9 window holds 100 * (disp[p + s - 4] - disp[p]).
// The 3x3 conv over the 49 unfold features is linear in the disparity, so for a pixel whose 3x3 neighbours are all inside
// the image it equals ONE 81-tap filter on the (zero-padded) disparity around p; the weights are pre-summed on the host
// (cer_conv3x3_f16x3_pack_collapsed).  Only the centre tap reads this chunk, so only the tile's own 128 pixels are staged.
template <int NTHR>
__device__ __forceinline__ void hx_stage_disp9(char* __restrict__ ldsA, const float* __restrict__ ldsD, int c0) {
    for (int idx = threadIdx.x; idx < HX_TH * HX_TW * 4; idx += NTHR) {
        const int px = idx >> 2, g = idx & 3;
        const int hy = 1 + px / HX_TW, hx = 1 + px % HX_TW;
        const float ctr = ldsD[(hy + 3) * HX_DT_W + hx + 3];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + 8 * g + e;
            const int sy = c / 9, sx = c - sy * 9;
            v[e] = (c < 81) ? 100.0f * (ldsD[(hy - 1 + sy) * HX_DT_W + hx - 1 + sx] - ctr) : 0.f;
        }
        hx_write8(ldsA, hy * HX_HW + hx, g, v);
    }
}

// DMA the block's weight slice of one (chunk, tap) step into an LDS ring slot: NB/8 pieces of 1 KiB
template <int NB, int NWAVES>
__device__ __forceinline__ void hx_issue_B(char* __restrict__ ldsB, const _Float16* __restrict__ slice) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
#pragma unroll
    for (int i = 0; i < NB / 8 / NWAVES; ++i) {
        const int piece = wave + NWAVES * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slice + piece * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(ldsB + piece * 1024), 16, 0, 0);
    }
}

// WAVES_M x WAVES_N waves, each owning WM x WN MFMA tiles of 32 pixels x 32 channels; NBUF = weight ring depth
// PS: the tensor sources are pre-split (kind 3) - a compile-time choice so that only ONE tensor staging routine is inlined into
// the chunk loop (with both, the 128-channel kernels spill 16 VGPRs instead of 2)
template <int WAVES_M, int WAVES_N, int WM, int WN, int NBUF, int MINW, int EPI, bool PS>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINW) void conv3x3_f16x3_kernel(const ConvArgsX a) {
    static_assert(WAVES_M * WM == HX_TH, "tile config");
    constexpr int NWAVES = WAVES_M * WAVES_N, NTHR = 64 * NWAVES;
    constexpr int NB = WAVES_N * WN * 32;                  // output channels per block
    constexpr int B_BYTES = NB * 128;                      // per (chunk, tap) step: NB/32 n-tiles x 4 KiB
    constexpr int DMA_PER_WAVE = NB / 8 / NWAVES;          // global_load_lds instructions per wave per step
    extern __shared__ __attribute__((aligned(16))) char hx_smem[];
    char* ldsA = hx_smem;
    char* ldsB = hx_smem + HX_A_BYTES;                     // NBUF ring slots of B_BYTES
    float* ldsD = reinterpret_cast<float*>(hx_smem + HX_A_BYTES + NBUF * B_BYTES);

    const int tile = blockIdx.x;
    const int ty0 = (tile / a.tiles_x) * HX_TH, tx0 = (tile % a.tiles_x) * HX_TW;
    const int nb0 = blockIdx.y * NB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, kg = lane >> 5;
    const int NT = a.cout / 32;

#if HX_TRACE
    unsigned long long trace_v[4];
    unsigned long long* trace_base = (unsigned long long*)a.aux2 + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * NWAVES + wave) * 64 * 4;
    const unsigned long long trace_t0 = __builtin_readcyclecounter();
#endif
    // ---- pipeline prologue: the weight DMA ring runs NBUF-1 steps ahead of the multiply, across chunk boundaries
    // tiles whose pixels all have their 3x3 neighbourhood inside the image take the collapsed disparity chunks
    const bool coll = a.wpk_c && ty0 >= 1 && ty0 + HX_TH <= a.h - 1 && tx0 >= 1 && tx0 + HX_TW <= a.w - 1;
    int nsteps = 0;
    for (int s = 0; s < a.nsrc; ++s) nsteps += (coll && a.kind[s] == 1) ? 3 : (a.chpad[s] / HX_KC) * 9;
    const _Float16* wbase = (coll ? a.wpk_c : a.wpk) + (long)(nb0 / 32) * 2048;      // + step * NT * 2048
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < nsteps && !(HX_ABL & 16)) hx_issue_B<NB, NWAVES>(ldsB + i * B_BYTES, wbase + (long)i * NT * 2048);
    for (int s = 0; s < a.nsrc; ++s)
        if (a.kind[s] == 1) hx_load_disp_tile<NTHR>(ldsD, a.src[s], a.h, a.w, ty0, tx0);

    floatx16 accm[WM][WN], accl[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int gy = ty0 + wm * WM + m;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int co = nb0 + (wn * WN + n) * 32 + li;
            floatx16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.f;
            if (a.init && !(HX_ABL & 8)) {
                // (with the LDS epilogue the hoisted term is added there, through 16-byte loads)
                if (!HX_EPI2 || EPI == HX_EPI_DELTA) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gx = tx0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        if (gy < a.h && gx < a.w) v[r] = a.init[((long)gy * a.w + gx) * a.cout + co];
                    }
                }
            } else if (a.bias) {
                const float b = a.bias[co];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = b;
            }
            accm[m][n] = v;
#pragma unroll
            for (int r = 0; r < 16; ++r) accl[m][n][r] = 0.f;
        }
    }

    // MFMA operands of one k16-step: A hi/lo for the wave's WM pixel tiles, B hi/lo for its WN channel tiles
    struct Frag {
        half8 ah[WM], al[WM], bh[WN], bl[WN];
    };
    auto load_frags = [&](Frag& f, const char* B, int dy, int dx, int ks) {
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int row = (wm * WM + m + dy) * HX_HW + li + dx;
            const char* p = ldsA + row * HX_AS + ks * 32 + kg * 16;
            f.ah[m] = *reinterpret_cast<const half8*>(p);
            f.al[m] = *reinterpret_cast<const half8*>(p + 64);
        }
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const char* p = B + (((wn * WN + n) * 2 + ks) * 2) * 1024 + lane * 16;
            f.bh[n] = *reinterpret_cast<const half8*>(p);
            f.bl[n] = *reinterpret_cast<const half8*>(p + 1024);
        }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                if (HX_ABL & 1) {
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" ::"v"(f.ah[m]), "v"(f.al[m]), "v"(f.bh[n]), "v"(f.bl[n]));
#endif
                } else {
                    accm[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bh[n], accm[m][n], 0, 0, 0);
                    accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bl[n], accl[m][n], 0, 0, 0);
                    accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[m], f.bh[n], accl[m][n], 0, 0, 0);
                }
            }
    };
    // (64-channel blocks only: with 128 channels per block the extra live fragment set does not fit 128 VGPRs.)
    constexpr bool SKEW = (WM * WN == 1) || (MINW <= 2);   // needs a second fragment set in registers
    // The MFMA stream is skewed by half a tap against the LDS reads: the second k16-step of tap t-1 (already in registers)
    // is multiplied right after barrier(t) while the first operands of tap t travel from LDS, so the matrix pipe has work
    // during the post-barrier LDS latency.  `pend` starts as zeros (a harmless first multiply) and is drained at the end.
    Frag pend;
    if constexpr (SKEW)
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) { pend.ah[m][e] = (_Float16)0; pend.al[m][e] = (_Float16)0; }
    if constexpr (SKEW)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int e = 0; e < 8; ++e) { pend.bh[n][e] = (_Float16)0; pend.bl[n][e] = (_Float16)0; }

#if HX_TRACE
    const unsigned long long trace_t1 = __builtin_readcyclecounter();
#endif
    int step = 0;                                          // chunk * 9 + tap; ring slot = step % NBUF
    for (int s = 0; s < a.nsrc; ++s) {
        const bool c9 = coll && a.kind[s] == 1;
        const int chunk_end = c9 ? 96 : a.chpad[s], ntaps = c9 ? 1 : 9;
        for (int c0 = 0; c0 < chunk_end; c0 += HX_KC) {
            __syncthreads();                               // every wave has finished reading A (previous chunk, tap 8); ldsD visible
            if (!(HX_ABL & 4)) {
                if (a.kind[s] != 1) {
                    if constexpr (PS) hx_stage_presplit<NTHR>(ldsA, a, s, c0, ty0, tx0);
                    else hx_stage_tensor<NTHR>(ldsA, a, s, c0, ty0, tx0);
                } else if (c9) hx_stage_disp9<NTHR>(ldsA, ldsD, c0);
                else hx_stage_disp<NTHR>(ldsA, ldsD, a, c0, ty0, tx0);
            }
#pragma unroll 1
            for (int t = 0; t < ntaps; ++t, ++step) {
                const int tap = c9 ? 4 : t;
                // B[step] was DMA'd NBUF-1 steps ago.  VMEM ops retire in order, so leaving the DMAs of the NBUF-2 younger
                // steps in flight still guarantees B[step].  (Builtin waits: hipcc's own scoreboard must see them.)
                HX_STAMP(0);
                const int younger = min(NBUF - 2, nsteps - 1 - step);
                if (NBUF >= 3 && younger >= 1) __builtin_amdgcn_s_waitcnt(0x0F70 | DMA_PER_WAVE);
                else __builtin_amdgcn_s_waitcnt(0x0F70);
                __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this wave's LDS writes (A tile) are done
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");    // (the bare builtin does not order LDS accesses for the compiler)
                if (!(HX_ABL & 2)) __builtin_amdgcn_s_barrier();   // all shares of B[step] + A visible; slot of B[step-1] free
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                HX_STAMP(1);
                if (step + NBUF - 1 < nsteps && !(HX_ABL & 16))
                    hx_issue_B<NB, NWAVES>(ldsB + ((step + NBUF - 1) % NBUF) * B_BYTES, wbase + (long)(step + NBUF - 1) * NT * 2048);
                const char* B = ldsB + (step % NBUF) * B_BYTES;
                const int dy = tap / 3, dx = tap - dy * 3;
                if constexpr (SKEW) {
                    Frag cur;
                    load_frags(cur, B, dy, dx, 0);
                    mma(pend);                             // previous tap, second k16-step: covers the LDS latency of `cur`
                    __builtin_amdgcn_sched_barrier(0);
                    load_frags(pend, B, dy, dx, 1);
                    mma(cur);                              // covers the LDS latency of the new `pend`
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        half8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
                        for (int m = 0; m < WM; ++m) {
                            const int row = (wm * WM + m + dy) * HX_HW + li + dx;
                            const char* p = ldsA + row * HX_AS + ks * 32 + kg * 16;
                            ah[m] = *reinterpret_cast<const half8*>(p);
                            al[m] = *reinterpret_cast<const half8*>(p + 64);
                        }
#pragma unroll
                        for (int n = 0; n < WN; ++n) {
                            const char* p = B + (((wn * WN + n) * 2 + ks) * 2) * 1024 + lane * 16;
                            bh[n] = *reinterpret_cast<const half8*>(p);
                            bl[n] = *reinterpret_cast<const half8*>(p + 1024);
                        }
#pragma unroll
                        for (int m = 0; m < WM; ++m)
#pragma unroll
                            for (int n = 0; n < WN; ++n) {
                                if (HX_ABL & 1) {
#if defined(__HIP_DEVICE_COMPILE__)
                                    asm volatile("" ::"v"(ah[m]), "v"(al[m]), "v"(bh[n]), "v"(bl[n]));
#endif
                                } else {
                                    accm[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], accm[m][n], 0, 0, 0);
                                    accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], accl[m][n], 0, 0, 0);
                                    accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], accl[m][n], 0, 0, 0);
                                }
                            }
                        if (ks == 0) HX_STAMP(2);
                    }
                }
#if HX_TRACE
                trace_v[3] = __builtin_readcyclecounter();
                if (lane == 0 && step < 62)
                    for (int k = 0; k < 4; ++k) trace_base[step * 4 + k] = trace_v[k];
#endif
            }
        }
    }
    if constexpr (SKEW) mma(pend);
#if HX_TRACE
    const unsigned long long trace_t2 = __builtin_readcyclecounter();
#endif

    // ---- epilogue: lane holds channel co, pixels x = tx0 + (r&3) + 8*(r>>2) + 4*kg of row gy
    const int half = a.cout / 2;
    if (HX_ABL & 8) {
#if defined(__HIP_DEVICE_COMPILE__)
        for (int m = 0; m < WM; ++m) for (int n = 0; n < WN; ++n) asm volatile("" ::"v"(accm[m][n]), "v"(accl[m][n]));
#endif
        return;
    }
    if constexpr (EPI == HX_EPI_DELTA) {
        // delta head, fused: hid = relu(conv) never leaves the CU.  The block's 128 x 128 (pixel x channel) hidden tile is
        // split to hi|lo f16 into LDS (A-operand order) and multiplied by the 256->1 conv's weights arranged as a
        // [channel x 9 taps] matrix: T[tap][p] = sum_c w2[tap][c] * hid[p][c] over this block's 128 channels, again with
        // 3 f16 MFMAs per product.  cer_delta_sum_f32 then gathers the 9 tap planes of both channel halves.
        static_assert(EPI != HX_EPI_DELTA || (NB == 128 && WM == 1), "DELTA epilogue is built for the 128-channel config");
        char* Hhi = hx_smem;
        char* Hlo = hx_smem + 128 * HX_HS;
        __syncthreads();                                   // main loop finished everywhere: A/B LDS can be reused
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int col = (wn * WN + n) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const float v = fmaxf(fmaf(accl[0][n][r], 1.0f / 2048.0f, accm[0][n][r]), 0.f);
                _Float16 h, l;
                hx_split(v, h, l);
                *reinterpret_cast<_Float16*>(Hhi + prow * HX_HS + col * 2) = h;
                *reinterpret_cast<_Float16*>(Hlo + prow * HX_HS + col * 2) = l;
            }
        }
        __syncthreads();
        if (wave < 4) {                                    // one M-tile (image row of the tile) per wave
            const _Float16* w2 = reinterpret_cast<const _Float16*>(a.aux) + (long)blockIdx.y * 8 * 2 * 512;
            floatx16 tm, tl;
#pragma unroll
            for (int r = 0; r < 16; ++r) { tm[r] = 0.f; tl[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const half8 ah = *reinterpret_cast<const half8*>(Hhi + (wave * 32 + li) * HX_HS + ks * 32 + kg * 16);
                const half8 al = *reinterpret_cast<const half8*>(Hlo + (wave * 32 + li) * HX_HS + ks * 32 + kg * 16);
                const half8 bh = *reinterpret_cast<const half8*>(w2 + (ks * 2 + 0) * 512 + lane * 8);
                const half8 bl = *reinterpret_cast<const half8*>(w2 + (ks * 2 + 1) * 512 + lane * 8);
                tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, tm, 0, 0, 0);
                tl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, tl, 0, 0, 0);
                tl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, tl, 0, 0, 0);
            }
            // tap planes go through LDS so that each plane row leaves as one contiguous 128-B store (the direct
            // 4-B scatter from the MFMA layout cost 6.5x write amplification: 55.7 MB written for 8.5 MB of payload)
            float* Tl = reinterpret_cast<float*>(hx_smem + 2 * 128 * HX_HS) + wave * 9 * 32;      // [9 taps][32 px] per wave
            if (li < 9) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Tl[li * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg] = fmaf(tl[r], 1.0f / 2048.0f, tm[r]);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0); the exchange is wave-local
            const int gy = ty0 + wave;
            if (gy < a.h) {
                for (int idx = lane; idx < 9 * 32; idx += 64) {
                    const int tap = idx >> 5, gx = tx0 + (idx & 31);
                    if (gx < a.w) a.out[((long)blockIdx.y * 9 + tap) * a.h * a.w + (long)gy * a.w + gx] = Tl[idx];
                }
            }
        }
        return;
    }
    if constexpr (HX_EPI2 != 0) {
        // Each wave transposes its tile through a private LDS patch (the activation / weight buffers are free now): the MFMA
        // layout gives a lane one channel of 16 pixels, i.e. 4-byte accesses 512 B apart; after the transpose a lane owns 4
        // consecutive channels of a pixel, so `init`, `aux`, `aux2` and the outputs move as 16-byte accesses (4x fewer
        // vector-memory instructions - the epilogue is issue-bound, not bandwidth-bound).
        constexpr int CW = WN * 32;                        // channels of a wave
        constexpr int PITCH = CW + 4;                      // floats per pixel row of the patch
        static_assert((long)NWAVES * 32 * PITCH * 4 <= (long)HX_A_BYTES + (long)NBUF * B_BYTES, "epilogue patch does not fit");
        float* Et = reinterpret_cast<float*>(hx_smem) + wave * (32 * PITCH);
        __syncthreads();                                   // main loop finished everywhere: LDS can be reused
#pragma unroll
        for (int m = 0; m < WM; ++m) {
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Et[((r & 3) + 8 * (r >> 2) + 4 * kg) * PITCH + n * 32 + li] = fmaf(accl[m][n][r], 1.0f / 2048.0f, accm[m][n][r]);
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the patch is wave-private
            const int gy = ty0 + wm * WM + m;
            constexpr int G = CW / 8;                      // 8-channel groups per pixel
#pragma unroll 1                                           // one item at a time: four interleaved items of 32 temporaries each spill
            for (int j = 0; j < 32 * G / 64; ++j) {
                const int idx = lane + 64 * j;
                const int px = idx / G, g = idx - px * G;
                const int gx = tx0 + px;
                const float4 va = *reinterpret_cast<const float4*>(Et + px * PITCH + 8 * g);
                const float4 vb = *reinterpret_cast<const float4*>(Et + px * PITCH + 8 * g + 4);
                if (gy >= a.h || gx >= a.w) continue;
                const long pix = (long)gy * a.w + gx;
                const int co = nb0 + wn * CW + 8 * g;
                float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
                auto ld8 = [](const float* q, float (&o)[8]) {
                    const float4 t0 = cer_ld4(q), t1 = cer_ld4(q + 4);
                    o[0] = t0.x; o[1] = t0.y; o[2] = t0.z; o[3] = t0.w; o[4] = t1.x; o[5] = t1.y; o[6] = t1.z; o[7] = t1.w;
                };
                auto st8 = [](float* q, const float (&o)[8]) {
                    *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(q + 4) = make_float4(o[4], o[5], o[6], o[7]);
                };
                if (a.init) {
                    float t[8];
                    ld8(a.init + pix * a.cout + co, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t[e];
                }
                if (EPI == CER_EPI_LINEAR || EPI == CER_EPI_RELU) {
                    if (EPI == CER_EPI_RELU)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    if (a.out_split) hx_store_split8(a.out, pix, a.cout, co, v);
                    else st8(a.out + pix * a.cout + co, v);
                } else if (EPI == CER_EPI_GATES) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = hx_sigmoid(v[e]);
                    if (co < half) {
                        st8(a.out + pix * half + co, v);                                 // z stays fp32 (blend operand of the q conv)
                    } else {
                        float hp[8];
                        if (a.aux_split) hx_load_split8(a.aux, pix, half, co - half, hp);
                        else ld8(a.aux + pix * half + (co - half), hp);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= hp[e];
                        if (a.out_split) hx_store_split8(a.out2, pix, half, co - half, v);
                        else st8(a.out2 + pix * half + (co - half), v);
                    }
                } else if (EPI == CER_EPI_GRU) {
                    float z[8], hp[8];
                    ld8(a.aux2 + pix * a.cout + co, z);
                    if (a.aux_split) hx_load_split8(a.aux, pix, a.cout, co, hp);
                    else ld8(a.aux + pix * a.cout + co, hp);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (1.0f - z[e]) * hp[e] + z[e] * tanhf(v[e]);
                    if (a.out_split) hx_store_split8(a.out, pix, a.cout, co, v);
                    else st8(a.out + pix * a.cout + co, v);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int gy = ty0 + wm * WM + m;
        if (gy >= a.h) continue;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int co = nb0 + (wn * WN + n) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gx = tx0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (gx >= a.w) continue;
                const long pix = (long)gy * a.w + gx;
                const float v = fmaf(accl[m][n][r], 1.0f / 2048.0f, accm[m][n][r]);
                if (EPI == CER_EPI_LINEAR) {
                    a.out[pix * a.cout + co] = v;
                } else if (EPI == CER_EPI_RELU) {
                    a.out[pix * a.cout + co] = fmaxf(v, 0.f);
                } else if (EPI == CER_EPI_GATES) {
                    const float g = hx_sigmoid(v);
                    if (co < half) a.out[pix * half + co] = g;
                    else a.out2[pix * half + (co - half)] = g * a.aux[pix * half + (co - half)];
                } else if (EPI == CER_EPI_GRU) {
                    const float q = tanhf(v);
                    const float z = a.aux2[pix * a.cout + co], hprev = a.aux[pix * a.cout + co];
                    a.out[pix * a.cout + co] = (1.0f - z) * hprev + z * q;
                }
            }
        }
    }
#if HX_TRACE
    if (EPI == CER_EPI_GATES && lane == 0) {
        trace_base[63 * 4 + 0] = trace_t0;
        trace_base[63 * 4 + 1] = trace_t1;
        trace_base[63 * 4 + 2] = trace_t2;
        trace_base[63 * 4 + 3] = __builtin_readcyclecounter();
        trace_base[62 * 4 + 0] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        trace_base[62 * 4 + 1] = (unsigned long long)nsteps;
        trace_base[62 * 4 + 2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
#endif
}

// ---------------------------------------------------------------------------------------- host side

static int hx_padded_channels(int ch, int kind) { return kind == 1 ? 64 : ((ch + HX_KC - 1) / HX_KC) * HX_KC; }   // kinds 0 and 3: tensors

extern "C" long cer_conv3x3_f16x3_packed_size(int Cout, int Kpad) {
    if (Cout <= 0 || Kpad <= 0 || Cout % 32 || Kpad % 32) return CER_ESHAPE;
    return (long)(Kpad / 32) * 9 * (Cout / 32) * 2048;     // in halves (2 bytes each)
}

// OIHW fp32 -> [chunk32][tap][ntile32][k16-step][hi|lo][lane][8] halves, lo scaled by 2^11
extern "C" int cer_conv3x3_f16x3_pack(const float* w, void* packed_v, int Cout, int Cin, const int* ch, const int* kind, int nsrc) {
    if (!w || !packed_v || !ch || !kind || nsrc <= 0 || nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    if (Cout % 32) return CER_ESHAPE;
    _Float16* packed = (_Float16*)packed_v;
    int real = 0, kpad = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (kind[s] == 1 && ch[s] != 49) return CER_ESHAPE;
        real += ch[s];
        kpad += hx_padded_channels(ch[s], kind[s]);
    }
    if (real != Cin) return CER_ESHAPE;
    int* map = new int[kpad];
    int k = 0, c = 0;
    for (int s = 0; s < nsrc; ++s) {
        const int pc = hx_padded_channels(ch[s], kind[s]);
        for (int i = 0; i < pc; ++i) map[k++] = (i < ch[s]) ? c + i : -1;
        c += ch[s];
    }
    const int NT = Cout / 32;
    for (int kc = 0; kc < kpad / 32; ++kc)
        for (int tap = 0; tap < 9; ++tap)
            for (int nt = 0; nt < NT; ++nt)
                for (int ks = 0; ks < 2; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = nt * 32 + (lane & 31);
                            const int ci = map[kc * 32 + ks * 16 + (lane >> 5) * 8 + e];
                            float v = ci < 0 ? 0.f : w[((long)co * Cin + ci) * 9 + tap];
                            v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
                            const long base = ((((long)kc * 9 + tap) * NT + nt) * 2 + ks) * 2;   // (hi|lo) plane index
                            packed[(base + 0) * 512 + lane * 8 + e] = hi;
                            packed[(base + 1) * 512 + lane * 8 + e] = lo;
                        }
    delete[] map;
    return CER_OK;
}

// Collapsed packing: same step order as above, except that a kind-1 (disparity) source contributes 3 single-tap steps
// holding the 81-tap filter  W9[co][s] = sum_{t + u = s} w[co][u][t]  -  [|s - 4| <= 1] * sum_u w[co][u][t = s - 3]
// (t over the 3x3 conv taps, u over the 7x7 unfold offsets, s over the 9x9 window; the second term is the centre
// subtraction of core/update.py:84).  Valid for pixels whose 3x3 neighbourhood lies inside the image.
extern "C" long cer_conv3x3_f16x3_collapsed_size(int Cout, const int* ch, const int* kind, int nsrc) {
    if (!ch || !kind || nsrc <= 0 || nsrc > CER_CONV_MAX_SRC || Cout <= 0 || Cout % 32) return CER_ESHAPE;
    long steps = 0;
    for (int s = 0; s < nsrc; ++s) steps += kind[s] == 1 ? 3 : (hx_padded_channels(ch[s], kind[s]) / 32) * 9;
    return steps * (Cout / 32) * 2048;
}

static void hx_pack_step(_Float16* packed, long step, int NT, int nt, const float* col /* [32 k][32 co] */) {
    for (int ks = 0; ks < 2; ++ks)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                float v = col[(ks * 16 + (lane >> 5) * 8 + e) * 32 + (lane & 31)];
                v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
                const long base = ((step * NT + nt) * 2 + ks) * 2;
                packed[(base + 0) * 512 + lane * 8 + e] = hi;
                packed[(base + 1) * 512 + lane * 8 + e] = lo;
            }
}

extern "C" int cer_conv3x3_f16x3_pack_collapsed(const float* w, void* packed_v, int Cout, int Cin, const int* ch, const int* kind, int nsrc) {
    if (!w || !packed_v || !ch || !kind || nsrc <= 0 || nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    if (Cout % 32) return CER_ESHAPE;
    int real = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (kind[s] == 1 && ch[s] != 49) return CER_ESHAPE;
        real += ch[s];
    }
    if (real != Cin) return CER_ESHAPE;
    _Float16* packed = (_Float16*)packed_v;
    const int NT = Cout / 32;
    float col[32 * 32];
    long step = 0;
    int c = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (kind[s] == 1) {
            for (int kc = 0; kc < 3; ++kc, ++step)
                for (int nt = 0; nt < NT; ++nt) {
                    for (int k = 0; k < 32; ++k)
                        for (int j = 0; j < 32; ++j) {
                            const int sidx = kc * 32 + k, co = nt * 32 + j;
                            double acc = 0.0;
                            if (sidx < 81) {
                                const int sy = sidx / 9, sx = sidx % 9;
                                for (int ty = 0; ty < 3; ++ty)
                                    for (int tx = 0; tx < 3; ++tx) {
                                        const int uy = sy - ty, ux = sx - tx;
                                        if (uy >= 0 && uy < 7 && ux >= 0 && ux < 7) acc += (double)w[((long)co * Cin + c + uy * 7 + ux) * 9 + ty * 3 + tx];
                                    }
                                if (sy >= 3 && sy <= 5 && sx >= 3 && sx <= 5) {
                                    const int t = (sy - 3) * 3 + (sx - 3);
                                    for (int u = 0; u < 49; ++u) acc -= (double)w[((long)co * Cin + c + u) * 9 + t];
                                }
                            }
                            col[k * 32 + j] = (float)acc;
                        }
                    hx_pack_step(packed, step, NT, nt, col);
                }
        } else {
            const int pc = hx_padded_channels(ch[s], kind[s]);
            for (int kc = 0; kc < pc / 32; ++kc)
                for (int tap = 0; tap < 9; ++tap, ++step)
                    for (int nt = 0; nt < NT; ++nt) {
                        for (int k = 0; k < 32; ++k)
                            for (int j = 0; j < 32; ++j) {
                                const int ci = kc * 32 + k;
                                col[k * 32 + j] = ci < ch[s] ? w[((long)(nt * 32 + j) * Cin + c + ci) * 9 + tap] : 0.f;
                            }
                        hx_pack_step(packed, step, NT, nt, col);
                    }
        }
        c += ch[s];
    }
    return CER_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int NBUF, int MINW, bool PS>
static int hx_launch_ps(const ConvArgsX& a, int epi, int nby, hipStream_t st) {
    constexpr int NB = WAVES_N * WN * 32;
    const size_t smem = HX_A_BYTES + NBUF * NB * 128 + HX_D_BYTES;
    const int tiles_y = (a.h + HX_TH - 1) / HX_TH;
    dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)nby), block(64 * WAVES_M * WAVES_N);
    switch (epi) {
        case CER_EPI_LINEAR: hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, CER_EPI_LINEAR, PS>), grid, block, smem, st, a); break;
        case CER_EPI_RELU: hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, CER_EPI_RELU, PS>), grid, block, smem, st, a); break;
        case CER_EPI_GATES: hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, CER_EPI_GATES, PS>), grid, block, smem, st, a); break;
        case CER_EPI_GRU: hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, CER_EPI_GRU, PS>), grid, block, smem, st, a); break;
        case HX_EPI_DELTA:
            if constexpr (WAVES_N * WN * 32 == 128 && WM == 1) {
                hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, HX_EPI_DELTA, PS>), grid, block, smem, st, a);
                break;
            } else {
                return CER_ESHAPE;
            }
        default: return CER_EINVAL;
    }
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int NBUF, int MINW>
static int hx_launch(const ConvArgsX& a, int epi, int nby, hipStream_t st) {
    bool any3 = false, any0 = false;
    for (int s = 0; s < a.nsrc; ++s) {
        any3 = any3 || a.kind[s] == 3;
        any0 = any0 || a.kind[s] == 0;
    }
    if (any3 && any0) return CER_EINVAL;                   // tensor sources of one call are all fp32 or all split32
    if (any3) return hx_launch_ps<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, true>(a, epi, nby, st);
    return hx_launch_ps<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, false>(a, epi, nby, st);
}

extern "C" int cer_conv3x3_f16x3(const cer_conv_inputs* in, const void* packed_w, const void* packed_collapsed, const float* bias,
                                 const float* init, float* out, float* out2, const float* aux, const float* aux2, int h, int w, int Cout, int epi, void* stream) {
    if (!in || !packed_w || !out || h <= 0 || w <= 0 || Cout <= 0) return CER_EINVAL;
    if (in->nsrc <= 0 || in->nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    const int out_split = (epi & CER_EPI_OUT_SPLIT) != 0, aux_split = (epi & CER_EPI_AUX_SPLIT) != 0;
    epi &= ~(CER_EPI_OUT_SPLIT | CER_EPI_AUX_SPLIT);
    if ((out_split || aux_split) && (epi == HX_EPI_DELTA || !HX_EPI2)) return CER_EINVAL;
    if (epi == CER_EPI_GATES && (!out2 || !aux)) return CER_EINVAL;
    if (epi == CER_EPI_GRU && (!aux || !aux2)) return CER_EINVAL;
    if (epi == HX_EPI_DELTA && (!aux || Cout % 128 != 0)) return CER_EINVAL;
    if (Cout % 64 != 0) return CER_ESHAPE;
    ConvArgsX a;
    memset(&a, 0, sizeof(a));
    a.nsrc = in->nsrc;
    for (int s = 0; s < in->nsrc; ++s) {
        if (!in->src[s]) return CER_EINVAL;
        if (in->kind[s] != 0 && in->kind[s] != 1 && in->kind[s] != 3) return CER_EINVAL;
        if (in->kind[s] != 1 && (in->ch[s] % HX_KC != 0)) return CER_ESHAPE;
        if (in->kind[s] == 1 && in->ch[s] != 49) return CER_ESHAPE;
        if (in->kind[s] != 1 && !cer_aligned16(in->src[s])) return CER_EALIGN;
        a.src[s] = in->src[s];
        a.ch[s] = in->ch[s];
        a.kind[s] = in->kind[s];
        a.chpad[s] = hx_padded_channels(in->ch[s], in->kind[s]);
    }
    if (!cer_aligned16(packed_w) || !cer_aligned16(packed_collapsed)) return CER_EALIGN;
    a.wpk = (const _Float16*)packed_w;
    a.wpk_c = (const _Float16*)packed_collapsed;
    a.bias = bias;
    a.init = init;
    a.out = out;
    a.out2 = out2;
    a.aux = aux;
    a.aux2 = aux2;
    a.h = h;
    a.w = w;
    a.cout = Cout;
    a.tiles_x = (w + HX_TW - 1) / HX_TW;
    a.out_split = out_split;
    a.aux_split = aux_split;
    hipStream_t st = (hipStream_t)stream;
    // 128 output channels per block: 8 waves (4 x 2) of 32 px x 64 ch, 3-slot weight ring, 2 blocks (16 waves) per CU;
    //  64 output channels per block: 8 waves (4 x 2) of 32 px x 32 ch, 3-slot ring, 2 blocks (16 waves) per CU.
#if HX_TH == 8                          // experiment: 8 x 32 pixel tile, 8 waves of 64 px x (64 | 32) ch, 256 registers, 1 block/CU
    if (epi == HX_EPI_DELTA) return CER_ESHAPE;
    if (Cout % 128 == 0) return hx_launch<4, 2, 2, 2, 3, 2>(a, epi, Cout / 128, st);
    return hx_launch<4, 2, 2, 1, 3, 2>(a, epi, Cout / 64, st);
#elif defined(HX_CFG) && HX_CFG == 1      // experiment: 4-wave blocks, 64 px x 64 ch per wave, 256 registers, 2 blocks/CU
    if (Cout % 128 == 0 && epi != HX_EPI_DELTA) return hx_launch<2, 2, 2, 2, 3, 2>(a, epi, Cout / 128, st);
#elif defined(HX_CFG) && HX_CFG == 2    // experiment: 4-wave blocks, 128 px x 32 ch per wave
    if (Cout % 128 == 0 && epi != HX_EPI_DELTA) return hx_launch<1, 4, 4, 1, 3, 2>(a, epi, Cout / 128, st);
#elif defined(HX_CFG) && HX_CFG == 3    // experiment: 4-wave blocks, 32 px x 128 ch per wave
    if (Cout % 128 == 0 && epi != HX_EPI_DELTA) return hx_launch<4, 1, 1, 4, 3, 2>(a, epi, Cout / 128, st);
#endif
#if HX_TH == 4
    if (Cout % 128 == 0) return hx_launch<4, 2, 1, 2, 3, 4>(a, epi, Cout / 128, st);
    return hx_launch<4, 2, 1, 1, 3, 4>(a, epi, Cout / 64, st);
#endif
}

// ---- delta head tail for the fused path ------------------------------------------------------------------
// w2 OIHW [1, C, 3, 3] -> B fragments of the [C x 9 (padded to 32)] projection, per 128-channel half:
// [half][k16-step 8][hi|lo][lane 64][8] halves; lane (tap = lane & 31, kg = lane >> 5) holds channels
// half*128 + ks*16 + kg*8 + e.
extern "C" long cer_delta_proj_packed_size(int C) { return C % 128 ? CER_ESHAPE : (long)(C / 128) * 8 * 2 * 512; }

extern "C" int cer_delta_proj_pack(const float* w2, void* packed_v, int C) {
    if (!w2 || !packed_v) return CER_EINVAL;
    if (C % 128) return CER_ESHAPE;
    _Float16* packed = (_Float16*)packed_v;
    for (int hf = 0; hf < C / 128; ++hf)
        for (int ks = 0; ks < 8; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int tap = lane & 31, c = hf * 128 + ks * 16 + (lane >> 5) * 8 + e;
                    float v = tap < 9 ? w2[(long)c * 9 + tap] : 0.f;
                    v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
                    packed[(((long)hf * 8 + ks) * 2 + 0) * 512 + lane * 8 + e] = hi;
                    packed[(((long)hf * 8 + ks) * 2 + 1) * 512 + lane * 8 + e] = lo;
                }
    return CER_OK;
}

// delta[p] = 0.01 * (bias + sum_half sum_tap T[half][tap][p + (ky-1, kx-1)]) (zero outside), disp_out = disp_in + delta
__global__ __launch_bounds__(256) void delta_sum_kernel(const float* __restrict__ T, int nhalf, float bias, const float* __restrict__ disp_in,
                                                        float* __restrict__ disp_out, float* __restrict__ delta, int h, int w) {
    const long P = (long)h * w;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int y = (int)(p / w), x = (int)(p % w);
    float s = 0.f;
    for (int hf = 0; hf < nhalf; ++hf)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) s += T[((long)hf * 9 + tap) * P + (long)yy * w + xx];
        }
    const float dl = 0.01f * (s + bias);
    if (delta) delta[p] = dl;
    disp_out[p] = disp_in[p] + dl;
}

extern "C" int cer_delta_sum_f32(const float* T, int nhalf, float bias, const float* disp_in, float* disp_out, float* delta, int h, int w,
                                 void* stream) {
    if (!T || !disp_in || !disp_out || nhalf <= 0 || h <= 0 || w <= 0) return CER_EINVAL;
    const long P = (long)h * w;
    hipLaunchKernelGGL(delta_sum_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, nhalf, bias, disp_in, disp_out,
                       delta, h, w);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// ---- split32 layout conversion (model load / API boundaries): fp32 [P, C] <-> per pixel and 32-channel chunk 32 hi | 32 lo halves
__global__ __launch_bounds__(256) void split32_kernel(const float* __restrict__ src, float* __restrict__ dst, long n8, int C, int inverse) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per 8 consecutive channels
    if (i >= n8) return;
    const long e = i * 8;
    const long pix = e / C;
    const int co = (int)(e - pix * C);
    float v[8];
    if (!inverse) {
        const float4 t0 = cer_ld4(src + e), t1 = cer_ld4(src + e + 4);
        v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
        hx_store_split8(dst, pix, C, co, v);
    } else {
        hx_load_split8(src, pix, C, co, v);
        *reinterpret_cast<float4*>(dst + e) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + e + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

extern "C" int cer_split32_f32(const float* src, float* dst, long P, int C, int inverse, void* stream) {
    if (!src || !dst || P <= 0 || C <= 0) return CER_EINVAL;
    if (C % 32 != 0) return CER_ESHAPE;
    if (!cer_aligned16(src) || !cer_aligned16(dst)) return CER_EALIGN;
    const long n8 = P * C / 8;
    hipLaunchKernelGGL(split32_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n8, C, inverse);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
