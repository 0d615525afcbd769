// Encoder stem on the matrix cores: 7x7 stride-2 pad-3 convolution 3 -> 32 on the raw image (reference: core/extractor.py:81,145;
// normalisation core/raft.py:40-41), round 2.  The direct fp32 kernel (enc_conv.hip: enc_stem_kernel) is VALU-bound at 4.7x its HBM
// floor; here the conv is an implicit GEMM on v_mfma_f32_32x32x16_f16 with the single-accumulator split-f16 arithmetic of
// conv_s16.hip (x*2^14 and w*2^kw as hi | lo halves, 3 MFMAs per product into one fp32 accumulator: fp32-class).
//
// K layout.  K = 147 = 7 rows x (7 columns x 3 channels) is ragged; pixels are padded to 4 channels and rows to 8 columns, so that
// K = 7 x 32 = 14 k16-steps and - the point - every B fragment is ONE aligned ds_read_b128: the input patch of a tile sits in LDS
// as [row][column][4 halves] (8 bytes per pixel; hi plane and lo plane), an output pixel (oy, ox) of k16-step s = (ky, half)
// needs columns 2ox + 4half + 2kg + {0, 1} of row 2oy + ky: 16 bytes at 16 * (row * 36 + ox + 2half + kg).  27 % of the MFMA
// work multiplies by the zero weights of the padding; the matrix pipe is not what bounds this kernel.
//
// Block = 4 waves, persistent over 8 x 32-pixel output tiles (wave w owns output rows 2w, 2w+1 = two m-tiles of 32 consecutive
// pixels); all 14 weight fragments (hi | lo) stay in registers for the whole kernel.  Per tile: stage the 21 x 69 input patch
// (three coalesced dword loads per pixel from the NCHW planes, normalise, split, two ds_write_b64), 14 x 6 MFMAs per wave, raw
// output as 16-byte stores (A = weights: a lane ends up with 4 consecutive channels of one pixel), per-tile (sum, sum of squares)
// per channel for the instance norm that follows (butterfly over the 32 pixel lanes, combined across the waves in LDS;
// deterministic: one record per tile, reduced in fp64 by cer_enc_stats_reduce_f32).
#include "common.hpp"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

typedef _Float16 sm_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 sm_half4 __attribute__((ext_vector_type(4)));
typedef float sm_floatx16 __attribute__((ext_vector_type(16)));

#define SM_TH 8                        // output rows per tile
#define SM_TW 32                       // output columns per tile
#define SM_PR (2 * SM_TH + 5)          // patch rows    (21)
#define SM_PC 72                       // patch columns (2 * 32 + 5 = 69, padded: the last k-group of ox = 31 reads column 69)
#define SM_PLANE (SM_PR * SM_PC * 8)   // bytes per plane
#define SM_STEPS 14
#define SM_XLOG2 14                    // activation scale 2^14 (|x| <= 1 after normalisation)

__global__ __launch_bounds__(256, 2) void enc_stem_s16_kernel(const float* __restrict__ img, const _Float16* __restrict__ wpk,
                                                              const float* __restrict__ bias, float* __restrict__ out, float* __restrict__ part,
                                                              int H, int W, int ho, int wo, int tiles_x, int tiles_per_img, int total,
                                                              int normalize, float invS) {
    __shared__ __attribute__((aligned(16))) char patch[2 * SM_PLANE];
    __shared__ float red[4][32][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kg = lane >> 5;
    sm_half8 wh[SM_STEPS], wl[SM_STEPS];
#pragma unroll
    for (int s = 0; s < SM_STEPS; ++s) {
        wh[s] = *reinterpret_cast<const sm_half8*>(wpk + ((s * 2 + 0) * 64 + lane) * 8);
        wl[s] = *reinterpret_cast<const sm_half8*>(wpk + ((s * 2 + 1) * 64 + lane) * 8);
    }
    float4 bj[4];                                          // this lane's channels 8j + 4kg + 0..3
#pragma unroll
    for (int j = 0; j < 4; ++j) bj[j] = cer_ld4(bias + 8 * j + 4 * kg);
    const long Po = (long)ho * wo;
    const long plane = (long)H * W;
    // the raw patch values of the NEXT tile are requested into registers right after the current tile's patch is in LDS, and arrive
    // under its MFMA phase and epilogue
    constexpr int NIT = (SM_PR * SM_PC + 255) / 256;
    float px[NIT][3];
    auto request = [&](int tile_, int tidv) {
        const int n = tile_ / tiles_per_img, t = tile_ - n * tiles_per_img;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int iy0 = 2 * ty * SM_TH - 3, ix0 = 2 * tx * SM_TW - 3;
        const float* im = img + (long)n * 3 * plane;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tidv + 256 * i;
            const int r = idx / SM_PC, c = idx - r * SM_PC;
            const int iy = iy0 + r, ix = ix0 + c;
            px[i][0] = px[i][1] = px[i][2] = 0.f;
            if (idx < SM_PR * SM_PC && c < 2 * SM_TW + 5 && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const float* p = im + (long)iy * W + ix;
                px[i][0] = p[0]; px[i][1] = p[plane]; px[i][2] = p[2 * plane];
                if (normalize) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) px[i][k] = px[i][k] * (2.0f / 255.0f) - 1.0f;
                }
            }
        }
    };
    if ((int)blockIdx.x < total) request(blockIdx.x, tid);
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / tiles_per_img, t = tile - n * tiles_per_img;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int oy0 = ty * SM_TH, ox0 = tx * SM_TW;
        __syncthreads();                                   // the previous tile's fragments and `red` have been read
        const int tidv = tid ^ ((tile >> 30) * 0x11111111);    // = tid; derived from the loop counter so that the staging addresses are
                                                           // re-formed per tile instead of being hoisted out of the tile loop and spilled
        // ---- patch -> LDS: zero outside the image (the conv pads the NORMALISED image) and in the padding columns
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tidv + 256 * i;
            if (idx < SM_PR * SM_PC) {
                const float sc = (float)(1 << SM_XLOG2);
                cer_f2 a = (cer_f2){px[i][0], px[i][1]} * sc, b = (cer_f2){px[i][2], 0.f} * sc;
                a = __builtin_elementwise_min(__builtin_elementwise_max(a, (cer_f2){-65504.0f, -65504.0f}), (cer_f2){65504.0f, 65504.0f});
                b = __builtin_elementwise_min(__builtin_elementwise_max(b, (cer_f2){-65504.0f, -65504.0f}), (cer_f2){65504.0f, 65504.0f});
                const cer_h2 ah = __builtin_convertvector(a, cer_h2), bh = __builtin_convertvector(b, cer_h2);
                const cer_h2 al = __builtin_convertvector(a - __builtin_convertvector(ah, cer_f2), cer_h2);
                const cer_h2 bl = __builtin_convertvector(b - __builtin_convertvector(bh, cer_f2), cer_h2);
                *reinterpret_cast<sm_half4*>(patch + idx * 8) = (sm_half4){ah.x, ah.y, bh.x, bh.y};
                *reinterpret_cast<sm_half4*>(patch + SM_PLANE + idx * 8) = (sm_half4){al.x, al.y, bl.x, bl.y};
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total) request(tile + gridDim.x, tidv);
        // ---- 14 k16-steps: step s = (ky = s >> 1, column half = s & 1); this lane's k-group covers columns 4 half + 2 kg + {0, 1}
        sm_floatx16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        // (fragments one step ahead, pinned with scheduler fences: left alone hipcc requests all 56 fragments up front and spills)
        const int base = (2 * (2 * wave) * SM_PC + 2 * li + 2 * kg) * 8;
        sm_half8 xh[2], xl[2];
        auto load_x = [&](int s) {
            const int off = base + ((s >> 1) * SM_PC + 4 * (s & 1)) * 8;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                xh[m] = *reinterpret_cast<const sm_half8*>(patch + off + m * 2 * SM_PC * 8);
                xl[m] = *reinterpret_cast<const sm_half8*>(patch + SM_PLANE + off + m * 2 * SM_PC * 8);
            }
        };
        load_x(0);
#pragma unroll
        for (int s = 0; s < SM_STEPS; ++s) {
            const sm_half8 h0 = xh[0], h1 = xh[1], l0 = xl[0], l1 = xl[1];
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < SM_STEPS) load_x(s + 1);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], h0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], h1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], h0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], h1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], l0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], l1, acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: acc[m][4j + e] = 2^(14 + kw) * conv for channel 8j + 4kg + e of pixel (oy0 + 2 wave + m, ox0 + li).
        // Channel group j at a time (few live registers: the 28 weight fragments stay resident): raw output as 16-byte stores, then
        // the group's 4 sums and 4 sums of squares over this wave's 2 x 32 pixels by xor-shuffles inside each 32-lane half
        const int ox = ox0 + li;
        const bool v0 = oy0 + 2 * wave < ho && ox < wo, v1 = oy0 + 2 * wave + 1 < ho && ox < wo;
        float* o = out + ((long)n * Po + (long)(oy0 + 2 * wave) * wo + ox) * 32 + 4 * kg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 r0 = make_float4(acc[0][4 * j] * invS + bj[j].x, acc[0][4 * j + 1] * invS + bj[j].y, acc[0][4 * j + 2] * invS + bj[j].z,
                                          acc[0][4 * j + 3] * invS + bj[j].w);
            const float4 r1 = make_float4(acc[1][4 * j] * invS + bj[j].x, acc[1][4 * j + 1] * invS + bj[j].y, acc[1][4 * j + 2] * invS + bj[j].z,
                                          acc[1][4 * j + 3] * invS + bj[j].w);
            if (v0) *reinterpret_cast<float4*>(o + 8 * j) = r0;
            if (v1) *reinterpret_cast<float4*>(o + (long)wo * 32 + 8 * j) = r1;
            if (part) {
                float q[8];
                const float a0 = v0 ? 1.f : 0.f, a1 = v1 ? 1.f : 0.f;
                q[0] = a0 * r0.x + a1 * r1.x; q[1] = a0 * r0.y + a1 * r1.y; q[2] = a0 * r0.z + a1 * r1.z; q[3] = a0 * r0.w + a1 * r1.w;
                q[4] = a0 * r0.x * r0.x + a1 * r1.x * r1.x; q[5] = a0 * r0.y * r0.y + a1 * r1.y * r1.y;
                q[6] = a0 * r0.z * r0.z + a1 * r1.z * r1.z; q[7] = a0 * r0.w * r0.w + a1 * r1.w * r1.w;
#pragma unroll
                for (int hb = 16; hb >= 1; hb >>= 1)
#pragma unroll
                    for (int e = 0; e < 8; ++e) q[e] += __shfl_xor(q[e], hb);
                if (li == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { red[wave][8 * j + 4 * kg + e][0] = q[e]; red[wave][8 * j + 4 * kg + e][1] = q[4 + e]; }
                }
            }
        }
        if (part) {
            __syncthreads();
            if (tid < 64) {
                const int c = tid >> 1, w2 = tid & 1;
                part[(((long)n * tiles_per_img + t) * 32 + c) * 2 + w2] = red[0][c][w2] + red[1][c][w2] + red[2][c][w2] + red[3][c][w2];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- round 4: producer / consumer form
// The kernel above runs stage -> barrier -> MFMA -> epilogue -> barrier in every wave: 27.6 k cycles per tile and block (two
// blocks per CU) against 2.7 k of matrix work and ~5 k cycles of memory time per tile and CU.  Here, as in enc_pc.hip, a 512-thread
// persistent block (one per CU) is split by role: waves 0-3 fetch the NEXT tile's input patch as 16-byte loads of an aligned
// superset of its columns (patch column c = image column 64 tx - 4 + c; the kernel row is packed with its zero padding in FRONT so
// that the fragments stay 16-byte aligned: cer_enc_stem_s16_pack's second layout), normalise, split with v_fma_mix and fill one of
// two LDS patches; waves 4-7 multiply (orientation pixels x channels: a lane ends up with one channel of 16 pixels, so the
// statistics are in-lane sums) and store through a wave-private LDS transpose as 16-byte stores.  One s_barrier per tile.
// Needs W % 4 == 0 and a 16-byte aligned image base; everything else takes the kernel above.
#define SP_PATCH (2 * SM_PLANE)                    // hi plane | lo plane of one patch
#define SP_RED (2 * 4 * 32 * 2 * 4)                // statistics patches: [parity][consumer wave][channel][2]
#define SP_TR (4 * 32 * 36 * 4)                    // epilogue transposes: 32 pixels x 32 channels (pitch 36) per consumer wave
#define SP_SMEM (2 * SP_PATCH + SP_RED + SP_TR)

__device__ __forceinline__ void sp_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ __launch_bounds__(512, 2) void enc_stem_pc_kernel(const float* __restrict__ img, const _Float16* __restrict__ wpk,
                                                             const float* __restrict__ bias, float* __restrict__ out, float* __restrict__ part,
                                                             int H, int W, int ho, int wo, int tiles_x, int tiles_per_img, int total,
                                                             int normalize, float invS) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    float* red = reinterpret_cast<float*>(sp_smem + 2 * SP_PATCH);
    float* trp = reinterpret_cast<float*>(sp_smem + 2 * SP_PATCH + SP_RED);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work order as in enc_pc.hip: the 32 blocks of an XCD (block b: XCD b % 8) take vertically adjacent tiles of a column-major
    // enumeration, so that the 5 patch rows two such tiles share come from that XCD's L2
    const int G = (int)gridDim.x;
    const int woff = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
    const int ntile = max(0, (total - woff + G - 1) / G);
    const int tiles_y = tiles_per_img / tiles_x;
    auto work_tile = [&](int k, int& n, int& t, int& ty, int& tx) {
        const int widx = woff + k * G;
        n = widx / tiles_per_img;
        const int c = widx - n * tiles_per_img;
        tx = c / tiles_y;
        ty = c - tx * tiles_y;
        t = ty * tiles_x + tx;
    };
    const long plane = (long)H * W;
    const long Po = (long)ho * wo;
    if (wave < 4) {
        // ---------------- producers: item = (patch row r, column quad q): three 16-byte loads (one per colour plane) -> 4 pixels x
        // [3 channels + zero] halves, hi and lo: two ds_write_b128 each.  378 items per tile: 256 + 122.
        constexpr int NQ = SM_PC / 4, ITEMS = SM_PR * NQ;                  // 18, 378
        const float nsc = normalize ? (2.0f / 255.0f) * (float)(1 << SM_XLOG2) : (float)(1 << SM_XLOG2);
        const float nof = normalize ? -(float)(1 << SM_XLOG2) : 0.f;
        float4 px[2][3];
        unsigned padbits = 0;                             // bit i: quad i of the requested patch is zero padding (a predicate of its own -
                                                          // an in-band NaN marker would turn a real NaN in the image into padding instead of propagating it)
        auto request = [&](int k) {
            int n, t, ty, tx;
            work_tile(k, n, t, ty, tx);
            const int iy0 = 2 * ty * SM_TH - 3, ix0 = 2 * tx * SM_TW - 4;
            const float* im = img + (long)n * 3 * plane;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx / NQ, q = idx - r * NQ;
                const int iy = iy0 + r, ix = ix0 + 4 * q;
                const bool ok = idx < ITEMS && iy >= 0 && iy < H && ix >= 0 && ix < W;     // (W % 4 == 0: a quad is inside or outside as a whole)
                const float* p = im + (long)(ok ? iy : 0) * W + (ok ? ix : 0);
#pragma unroll
                for (int c = 0; c < 3; ++c) px[i][c] = cer_ld4(p + c * plane);
                padbits = (padbits & ~(1u << i)) | ((ok ? 0u : 1u) << i);                   // the quad is zero padding (of the NORMALISED image)
            }
        };
        auto stage = [&](int k) {
            char* buf = sp_smem + (k & 1) * SP_PATCH;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i;
                if (idx < ITEMS) {
                    const bool pad = (padbits >> i) & 1u;
                    const float v[3][4] = {{px[i][0].x, px[i][0].y, px[i][0].z, px[i][0].w}, {px[i][1].x, px[i][1].y, px[i][1].z, px[i][1].w},
                                           {px[i][2].x, px[i][2].y, px[i][2].z, px[i][2].w}};
                    unsigned hw[8], lw[8];                                 // pixel j: words 2j (channels 0, 1), 2j + 1 (channel 2, zero)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float y[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float t_ = __builtin_amdgcn_fmed3f(fmaf(v[c][j], nsc, nof), -65504.0f, 65504.0f);
                            y[c] = pad ? 0.f : t_;
                        }
                        const float one = 1.0f, mone = -1.0f;
                        unsigned h0, h1 = 0, l0, l1 = 0;
                        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(y[0]), "s"(one));
                        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(y[1]), "s"(one));
                        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(y[2]), "s"(one));
                        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h0), "s"(mone), "v"(y[0]));
                        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(h0), "s"(mone), "v"(y[1]));
                        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(h1), "s"(mone), "v"(y[2]));
                        hw[2 * j] = h0; hw[2 * j + 1] = h1; lw[2 * j] = l0; lw[2 * j + 1] = l1;
                    }
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4* dh = reinterpret_cast<u32x4*>(buf + idx * 32);
                    u32x4* dl = reinterpret_cast<u32x4*>(buf + SM_PLANE + idx * 32);
                    dh[0] = (u32x4){hw[0], hw[1], hw[2], hw[3]};
                    dh[1] = (u32x4){hw[4], hw[5], hw[6], hw[7]};
                    dl[0] = (u32x4){lw[0], lw[1], lw[2], lw[3]};
                    dl[1] = (u32x4){lw[4], lw[5], lw[6], lw[7]};
                }
            }
        };
        auto finalize = [&](int k) {                                       // statistics record of tile k: the four consumer waves in order
            if (!part || tid >= 64) return;
            int n, t, ty, tx;
            work_tile(k, n, t, ty, tx);
            const int tile_ = n * tiles_per_img + t;
            const int c = tid >> 1, w2 = tid & 1;
            const float* r = red + (k & 1) * (4 * 32 * 2);
            part[((long)tile_ * 32 + c) * 2 + w2] = r[(0 * 32 + c) * 2 + w2] + r[(1 * 32 + c) * 2 + w2] + r[(2 * 32 + c) * 2 + w2] + r[(3 * 32 + c) * 2 + w2];
        };
        if (ntile > 0) request(0);
        for (int k = 0; k < ntile; ++k) {
            stage(k);
            if (k + 1 < ntile) request(k + 1);
            sp_barrier();                                                  // #k: patch k & 1 is complete
            if (k >= 1) finalize(k - 1);
        }
        sp_barrier();                                                      // #ntile
        if (ntile >= 1) finalize(ntile - 1);
    } else {
        // ---------------- consumers: wave cw owns output rows 2 cw, 2 cw + 1 of the tile
        const int cw = wave - 4;
        const int li = lane & 31, kg = lane >> 5;
        sm_half8 wh[SM_STEPS], wl[SM_STEPS];
        const _Float16* wp2 = wpk + (long)SM_STEPS * 2 * 64 * 8;          // second layout: kernel rows with the padding column in front
#pragma unroll
        for (int s = 0; s < SM_STEPS; ++s) {
            wh[s] = *reinterpret_cast<const sm_half8*>(wp2 + ((s * 2 + 0) * 64 + lane) * 8);
            wl[s] = *reinterpret_cast<const sm_half8*>(wp2 + ((s * 2 + 1) * 64 + lane) * 8);
        }
        const float bch = bias[li];
        float* Et_w = trp + cw * (32 * 36) + 4 * kg * 36 + li;
        const float* Et_r = trp + cw * (32 * 36) + (lane >> 3) * 36 + 4 * (lane & 7);
        const int abase = (2 * (2 * cw) * SM_PC + 2 * li + 2 * kg) * 8;
        for (int k = 0; k < ntile; ++k) {
            int n, t, ty, tx;
            work_tile(k, n, t, ty, tx);
            const int oy0 = ty * SM_TH + 2 * cw, ox0 = tx * SM_TW;
            sp_barrier();                                                  // #k
            const char* patch = sp_smem + (k & 1) * SP_PATCH + abase;
            sm_floatx16 acc[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            sm_half8 xh[2], xl[2];
            auto load_x = [&](int s) {
                const int off = ((s >> 1) * SM_PC + 4 * (s & 1)) * 8;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    xh[m] = *reinterpret_cast<const sm_half8*>(patch + off + m * 2 * SM_PC * 8);
                    xl[m] = *reinterpret_cast<const sm_half8*>(patch + SM_PLANE + off + m * 2 * SM_PC * 8);
                }
            };
            load_x(0);
#pragma unroll
            for (int s = 0; s < SM_STEPS; ++s) {
                const sm_half8 h0 = xh[0], h1 = xh[1], l0 = xl[0], l1 = xl[1];
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < SM_STEPS) load_x(s + 1);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, wh[s], acc[0], 0, 0, 0);       // rows = pixels, columns = channels
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, wh[s], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, wl[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, wl[s], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(l0, wh[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(l1, wh[s], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue: acc[m][r] = 2^(14 + kw) conv of channel li at pixel (oy0 + m, ox0 + (r & 3) + 8 (r >> 2) + 4 kg)
            float ssum = 0.f, ssq = 0.f;
            auto epilogue = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int gy = oy0 + m;
                    float* o = out + ((long)n * Po + (long)gy * wo + ox0) * 32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pxc = (r & 3) + 8 * (r >> 2);
                        const float v = fmaf(acc[m][r], invS, bch);
                        if (FULL || (gy < ho && ox0 + pxc + 4 * kg < wo)) {
                            ssum += v;
                            ssq = fmaf(v, v, ssq);
                        }
                        Et_w[pxc * 36] = v;
                    }
                    const int lane_off = (lane >> 3) * 32 + 4 * (lane & 7);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const float4 v4 = *reinterpret_cast<const float4*>(Et_r + 8 * jj * 36);
                        if (FULL || (gy < ho && ox0 + (lane >> 3) + 8 * jj < wo)) *reinterpret_cast<float4*>(o + lane_off + 8 * jj * 32) = v4;
                    }
                }
            };
            if (oy0 + 2 <= ho && ox0 + 32 <= wo) epilogue(std::true_type{});
            else epilogue(std::false_type{});
            if (part) {
                const float s2 = ssum + __shfl_xor(ssum, 32), q2 = ssq + __shfl_xor(ssq, 32);
                if (kg == 0) {
                    float* r = red + (k & 1) * (4 * 32 * 2) + (cw * 32 + li) * 2;
                    r[0] = s2;
                    r[1] = q2;
                }
            }
        }
        sp_barrier();                                                      // #ntile
    }
}

// ---------------------------------------------------------------------------------------- host side
extern "C" long cer_enc_stem_s16_packed_size(void) { return 2L * SM_STEPS * 2 * 64 * 8; }     // halves: two layouts (below)

extern "C" int cer_enc_stem_s16_tiles(int ho, int wo) { return ((ho + SM_TH - 1) / SM_TH) * ((wo + SM_TW - 1) / SM_TW); }

// w_oihw [32][3][7][7] (host) -> A fragments [step][hi | lo][lane][8]: lane (channel = lane & 31, kg = lane >> 5), element e:
// ky = step >> 1, column = 4 (step & 1) + 2 kg + (e >> 2), input channel = e & 3 (column 7 and channel 3 are padding: zero).
// A second copy follows with the padding column in FRONT (kernel column = that index - 1): the layout of enc_stem_pc_kernel, whose
// patch starts one image column earlier (16-byte aligned loads).  *log2s_w = the power-of-two weight scale that was applied.
extern "C" int cer_enc_stem_s16_pack(const float* w_oihw, void* packed_v, int* log2s_w) {
    if (!w_oihw || !packed_v || !log2s_w) return CER_EINVAL;
    double wmax = 0.0;
    for (int i = 0; i < 32 * 147; ++i) wmax = fmax(wmax, fabs((double)w_oihw[i]));
    int k = wmax > 0.0 ? (int)floor(log2(16384.0 / wmax)) : 0;
    if (k > 24) k = 24;
    if (k < -24) k = -24;
    *log2s_w = k;
    const float sc = ldexpf(1.0f, k);
    _Float16* packed = static_cast<_Float16*>(packed_v);
    for (int layout = 0; layout < 2; ++layout)
        for (int s = 0; s < SM_STEPS; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int ch = lane & 31, kg = lane >> 5;
                    const int ky = s >> 1, col = 4 * (s & 1) + 2 * kg + (e >> 2) - layout, ci = e & 3;
                    float v = 0.f;
                    if (col >= 0 && col < 7 && ci < 3) v = w_oihw[((ch * 3 + ci) * 7 + ky) * 7 + col] * sc;
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    _Float16* dst = packed + (long)layout * SM_STEPS * 2 * 64 * 8;
                    dst[((long)(s * 2 + 0) * 64 + lane) * 8 + e] = hi;
                    dst[((long)(s * 2 + 1) * 64 + lane) * 8 + e] = lo;
                }
    return CER_OK;
}

extern "C" int cer_enc_stem_s16(const float* images, const void* packed_w, const float* bias, float* out, float* stats_partial, int N, int H,
                                int W, int normalize, int log2s_w, void* stream) {
    if (!images || !packed_w || !bias || !out || N <= 0 || H <= 0 || W <= 0) return CER_EINVAL;
    if (!cer_aligned16(packed_w) || !cer_aligned16(bias) || !cer_aligned16(out)) return CER_EALIGN;
    const int ho = (H + 6 - 7) / 2 + 1, wo = (W + 6 - 7) / 2 + 1;
    const int tiles_x = (wo + SM_TW - 1) / SM_TW;
    const long per_img = (long)cer_enc_stem_s16_tiles(ho, wo);
    const long total = per_img * N;
    if (total >= (1L << 31)) return CER_ESHAPE;
    const int ncu = cer_num_cus();
    if (W % 4 == 0 && cer_aligned16(images) && !getenv("CER_STEM_TILED")) {       // producer / consumer form: 16-byte image loads
        int dev = 0;
        (void)hipGetDevice(&dev);
        static bool raised[64];
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)enc_stem_pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SP_SMEM);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
        hipLaunchKernelGGL(enc_stem_pc_kernel, dim3((unsigned)(total < ncu ? total : ncu)), dim3(512), SP_SMEM, (hipStream_t)stream, images,
                           (const _Float16*)packed_w, bias, out, stats_partial, H, W, ho, wo, tiles_x, (int)per_img, (int)total, normalize,
                           ldexpf(1.0f, -(SM_XLOG2 + log2s_w)));
        CER_RETURN_IF_LAUNCH_FAILED();
        return CER_OK;
    }
    const unsigned grid = (unsigned)(total < 2L * ncu ? total : 2L * ncu);
    hipLaunchKernelGGL(enc_stem_s16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, (const _Float16*)packed_w, bias, out,
                       stats_partial, H, W, ho, wo, tiles_x, (int)per_img, (int)total, normalize, ldexpf(1.0f, -(SM_XLOG2 + log2s_w)));
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
