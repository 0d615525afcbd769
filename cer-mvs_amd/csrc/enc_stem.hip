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

// ---------------------------------------------------------------------------------------- host side
extern "C" long cer_enc_stem_s16_packed_size(void) { return (long)SM_STEPS * 2 * 64 * 8; }     // halves

extern "C" int cer_enc_stem_s16_tiles(int ho, int wo) { return ((ho + SM_TH - 1) / SM_TH) * ((wo + SM_TW - 1) / SM_TW); }

// w_oihw [32][3][7][7] (host) -> A fragments [step][hi | lo][lane][8]: lane (channel = lane & 31, kg = lane >> 5), element e:
// ky = step >> 1, column = 4 (step & 1) + 2 kg + (e >> 2), input channel = e & 3 (column 7 and channel 3 are padding: zero).
// *log2s_w = the power-of-two weight scale that was applied.
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
    for (int s = 0; s < SM_STEPS; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                const int ch = lane & 31, kg = lane >> 5;
                const int ky = s >> 1, col = 4 * (s & 1) + 2 * kg + (e >> 2), ci = e & 3;
                float v = 0.f;
                if (col < 7 && ci < 3) v = w_oihw[((ch * 3 + ci) * 7 + ky) * 7 + col] * sc;
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                packed[((long)(s * 2 + 0) * 64 + lane) * 8 + e] = hi;
                packed[((long)(s * 2 + 1) * 64 + lane) * 8 + e] = lo;
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
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const unsigned grid = (unsigned)(total < 2L * ncu ? total : 2L * ncu);
    hipLaunchKernelGGL(enc_stem_s16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, (const _Float16*)packed_w, bias, out,
                       stats_partial, H, W, ho, wo, tiles_x, (int)per_img, (int)total, normalize, ldexpf(1.0f, -(SM_XLOG2 + log2s_w)));
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
