// Encoder convolutions (reference: core/extractor.py:60-155 BasicEncoder "HR") on the same
// split-f16 MFMA engine as the update block (gru_f16x3.hip), generalised for the encoder's needs:
//   * batched images (blockIdx.z), channels-last fp32 activations [N, h*w, C];
//   * 3x3 / 1x1 kernels, stride 1 / 2, 32 or 64 output channels per block;
//   * instance norm + ReLU of the PRODUCER applied on the fly while the halo tile is staged
//     (x -> relu((x - mean[n,c]) * rstd[n,c])), so a normalised activation is never written to HBM;
//   * the epilogue emits per-block partial sums (sum, sum of squares) per output channel for the NEXT
//     instance norm (deterministic: partials + a tiny fp64 reduce kernel, no atomics);
//   * final 1x1 convs write straight into the consumers' layouts: feature maps x 1/8 with the 2-texel zero
//     border of the cost-build kernel; context map split into tanh (net) / relu (inp).
// Also here: the 7x7 stride-2 stem as a direct fp32 kernel (K = 147 is too ragged for MFMA tiles and it is 6 %
// of the encoder flops), the residual-merge kernel, and the statistics reduce.
#include "common.hpp"
#include <stdlib.h>
#include <string.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifndef ES_STREAM
#define ES_STREAM 1                    // 32 -> 32 3x3 stride-1 convolutions through the persistent streaming kernel (0: tiled kernel)
#endif
#define EC_AS 144                      // LDS bytes per halo pixel: 32 hi | 32 lo | 16 pad
#define EC_EPI_RAW 0                   // out = conv + bias (raw, pre-norm), optional stats partials
#define EC_EPI_FMAP 1                  // out = (conv + bias) * scale into a (possibly bordered) channels-last map
#define EC_EPI_CTX 2                   // co < Cout/2: tanh -> out ; else relu -> out2   (core/raft.py:57-60)

struct EncArgs {
    const float* src;                  // [N, h*w, Cin] channels-last
    const float* tf;                   // producer statistics [N*Cin][2] (mean, rstd) or NULL
    int tf_relu;                       // ReLU after the transform (also without statistics)
    const _Float16* wpk;
    const float* bias;
    float* out;
    float* out2;
    float* part;                       // stats partials [N][nblk][Cout][2] or NULL
    int h, w, cin;                     // input geometry
    int ho, wo, cout;                  // output geometry
    int tiles_x, nblk;
    int out_border;                    // EC_EPI_FMAP: zero border of the destination map
    float out_scale;
};

__device__ __forceinline__ void ec_split(float v, _Float16& hi, _Float16& lo) {
    const float x = fminf(fmaxf(v, -65504.0f), 65504.0f);
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * 2048.0f);
}

template <int NB, int NWAVES>
__device__ __forceinline__ void ec_issue_B(char* __restrict__ ldsB, const _Float16* __restrict__ slice) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
    constexpr int PIECES = NB / 8;
#pragma unroll
    for (int i = 0; i < (PIECES + NWAVES - 1) / NWAVES; ++i) {
        const int piece = wave + NWAVES * i;
        if (piece < PIECES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slice + piece * 512 + lane * 8),
                                             (__attribute__((address_space(3))) void*)(ldsB + piece * 1024), 16, 0, 0);
    }
}

// TH x 32 output pixels per block; WAVES_M x WAVES_N waves each owning WM x WN MFMA tiles of 32 px x 32 ch
template <int WAVES_M, int WAVES_N, int WM, int WN, int NBUF, int MINW, int TH, int STRIDE, int TAPS, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINW) void enc_conv_kernel(const EncArgs a) {
    static_assert(WAVES_M * WM == TH, "tile config");
    constexpr int NWAVES = WAVES_M * WAVES_N, NTHR = 64 * NWAVES;
    constexpr int NB = WAVES_N * WN * 32;
    constexpr int KS = (TAPS == 9) ? 3 : 1, PAD = (TAPS == 9) ? 1 : 0;
    constexpr int HH = (TH - 1) * STRIDE + KS, HW = 31 * STRIDE + KS, ROWS = HH * HW;
    constexpr int A_BYTES = ROWS * EC_AS;
    constexpr int B_BYTES = NB * 128;
    constexpr int PIECES = NB / 8;
    constexpr int DMA_PER_WAVE = (PIECES + NWAVES - 1) / NWAVES;
    extern __shared__ __attribute__((aligned(16))) char ec_smem[];
    char* ldsA = ec_smem;
    char* ldsB = ec_smem + A_BYTES;

    const int img = blockIdx.z;
    const int tile = blockIdx.x;
    const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * 32;      // output coordinates
    const int nb0 = blockIdx.y * NB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, kg = lane >> 5;
    const int NT = a.cout / 32;
    const int nchunks = a.cin / 32, nsteps = nchunks * TAPS;
    const float* src = a.src + (long)img * a.h * a.w * a.cin;
    const float* tf = a.tf ? a.tf + (long)img * a.cin * 2 : nullptr;

    const _Float16* wbase = a.wpk + (long)(nb0 / 32) * 2048;
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < nsteps) ec_issue_B<NB, NWAVES>(ldsB + i * B_BYTES, wbase + (long)i * NT * 2048);

    floatx16 accm[WM][WN], accl[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const float b = a.bias ? a.bias[nb0 + (wn * WN + n) * 32 + li] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accm[m][n][r] = b;
                accl[m][n][r] = 0.f;
            }
        }

    int step = 0;
    for (int c0 = 0; c0 < a.cin; c0 += 32) {
        __syncthreads();
        // ---- stage the halo of this 32-channel chunk: load all, then transform + split + write.
        // item = (halo pixel, 8-channel group); NTHR % 4 == 0, so a thread keeps the same channel group g for all its items
        // and its 8 (mean, rstd) pairs are loaded once per chunk.
        {
            constexpr int ITEMS = (ROWS * 4 + NTHR - 1) / NTHR;
            const int g = threadIdx.x & 3, prow0 = threadIdx.x >> 2;
            float4 raw[ITEMS][2];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int row = min(prow0 + (NTHR / 4) * i, ROWS - 1);
                const int hy = row / HW, hx = row - hy * HW;
                const int gy = min(max(ty0 * STRIDE + hy - PAD, 0), a.h - 1), gx = min(max(tx0 * STRIDE + hx - PAD, 0), a.w - 1);
                const float* p = src + ((long)gy * a.w + gx) * a.cin + c0 + 8 * g;
                raw[i][0] = cer_ld4(p);
                raw[i][1] = cer_ld4(p + 4);
            }
            float mu[8], rs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                mu[e] = tf ? tf[2 * (c0 + 8 * g + e)] : 0.f;
                rs[e] = tf ? tf[2 * (c0 + 8 * g + e) + 1] : 1.f;
            }
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int row = prow0 + (NTHR / 4) * i;
                if (row < ROWS) {
                    const int hy = row / HW, hx = row - hy * HW;
                    const int gy = ty0 * STRIDE + hy - PAD, gx = tx0 * STRIDE + hx - PAD;
                    const bool inside = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
                    float v[8] = {raw[i][0].x, raw[i][0].y, raw[i][0].z, raw[i][0].w, raw[i][1].x, raw[i][1].y, raw[i][1].z, raw[i][1].w};
                    half8 hi, lo;
                    float xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = tf ? (v[e] - mu[e]) * rs[e] : v[e];
                        if (a.tf_relu) x = fmaxf(x, 0.f);
                        xv[e] = inside ? x : 0.f;         // zero padding of the (normalised) activation
                    }
                    cer_split8(xv, hi, lo);
                    *reinterpret_cast<half8*>(ldsA + row * EC_AS + g * 16) = hi;
                    *reinterpret_cast<half8*>(ldsA + row * EC_AS + 64 + g * 16) = lo;
                }
            }
        }
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap, ++step) {
            const int younger = min(NBUF - 2, nsteps - 1 - step);
            if (NBUF >= 3 && younger >= 1) __builtin_amdgcn_s_waitcnt(0x0F70 | DMA_PER_WAVE);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");    // (the bare builtin does not order LDS accesses for the compiler)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            if (step + NBUF - 1 < nsteps)
                ec_issue_B<NB, NWAVES>(ldsB + ((step + NBUF - 1) % NBUF) * B_BYTES, wbase + (long)(step + NBUF - 1) * NT * 2048);
            const char* B = ldsB + (step % NBUF) * B_BYTES;
            const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    const int row = ((wm * WM + m) * STRIDE + dy) * HW + li * STRIDE + dx;
                    const char* p = ldsA + row * EC_AS + ks * 32 + kg * 16;
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
                        accm[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], accm[m][n], 0, 0, 0);
                        accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], accl[m][n], 0, 0, 0);
                        accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], accl[m][n], 0, 0, 0);
                    }
            }
        }
    }

    // ---- epilogue: lane holds channel co, pixels x = tx0 + (r&3) + 8*(r>>2) + 4*kg of output row gy
    float ssum[WN], ssq[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) { ssum[n] = 0.f; ssq[n] = 0.f; }
    const int wob = a.wo + 2 * a.out_border;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int gy = ty0 + wm * WM + m;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int co = nb0 + (wn * WN + n) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gx = tx0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (gy >= a.ho || gx >= a.wo) continue;
                const float v = fmaf(accl[m][n][r], 1.0f / 2048.0f, accm[m][n][r]);
                if (EPI == EC_EPI_RAW) {
                    // (the store itself goes through the LDS transpose below: 16-byte accesses)
                    ssum[n] += v;
                    ssq[n] = fmaf(v, v, ssq[n]);
                } else if (EPI == EC_EPI_FMAP) {
                    a.out[(((long)img * (a.ho + 2 * a.out_border) + gy + a.out_border) * wob + gx + a.out_border) * a.cout + co] = v * a.out_scale;
                } else {
                    const int half = a.cout / 2;
                    const long pix = ((long)img * a.ho + gy) * a.wo + gx;
                    if (co < half) a.out[pix * half + co] = tanhf(v);
                    else a.out2[pix * half + co - half] = fmaxf(v, 0.f);
                }
            }
        }
    }
    if (EPI == EC_EPI_RAW) {
        // raw output through a wave-private LDS transpose: the MFMA layout gives a lane one channel of 16 pixels (4-byte stores
        // 4*Cout bytes apart); afterwards a lane owns 4 consecutive channels of a pixel and stores 16 bytes
        constexpr int CW = WN * 32, PITCH = CW + 4, G = CW / 4;
        float* Et = reinterpret_cast<float*>(ec_smem) + (threadIdx.x >> 6) * (32 * PITCH);
        __syncthreads();                                   // A/B LDS no longer needed
#pragma unroll
        for (int m = 0; m < WM; ++m) {
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Et[((r & 3) + 8 * (r >> 2) + 4 * kg) * PITCH + n * 32 + li] = fmaf(accl[m][n][r], 1.0f / 2048.0f, accm[m][n][r]);
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the patch is wave-private
            const int gy = ty0 + wm * WM + m;
#pragma unroll
            for (int j = 0; j < 32 * G / 64; ++j) {
                const int idx = (threadIdx.x & 63) + 64 * j;
                const int px = idx / G, g = idx - px * G;
                const int gx = tx0 + px;
                const float4 v = *reinterpret_cast<const float4*>(Et + px * PITCH + 4 * g);
                if (gy < a.ho && gx < a.wo)
                    *reinterpret_cast<float4*>(a.out + (((long)img * a.ho + gy) * a.wo + gx) * a.cout + nb0 + wn * CW + 4 * g) = v;
            }
        }
    }
    if (EPI == EC_EPI_RAW && a.part) {
        // per-block partial statistics for the next instance norm: combine lane halves, then the WAVES_M waves through LDS
        __syncthreads();                                   // the transpose patches are no longer needed
        float* red = reinterpret_cast<float*>(ec_smem);    // [WAVES_M][NB][2]
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const float s = ssum[n] + __shfl_xor(ssum[n], 32), q = ssq[n] + __shfl_xor(ssq[n], 32);
            if (kg == 0) {
                red[(wm * NB + (wn * WN + n) * 32 + li) * 2 + 0] = s;
                red[(wm * NB + (wn * WN + n) * 32 + li) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (threadIdx.x < NB) {
            float s = 0.f, q = 0.f;
            for (int i = 0; i < WAVES_M; ++i) {
                s += red[(i * NB + threadIdx.x) * 2 + 0];
                q += red[(i * NB + threadIdx.x) * 2 + 1];
            }
            float* dst = a.part + (((long)img * a.nblk + tile) * a.cout + nb0 + threadIdx.x) * 2;
            dst[0] = s;
            dst[1] = q;
        }
    }
}

// ---- streaming variant for the half-resolution 32 -> 32 3x3 stride-1 convolutions (layer 1: core/extractor.py:87,109-111) ----
// These four convolutions per encoder move 1.3 GB each for 96 GFLOP: they are memory-streaming kernels, and in the tiled kernel
// above every one of the 20 350 blocks re-fetches the same 36 KiB of weights and pays its own prologue and per-tap barriers.
// Here persistent 4-wave blocks (two per CU, so that one block's VALU/LDS "commit" phase overlaps the other's MFMAs) keep all
// nine taps' weights in LDS and walk over the 8 x 32-pixel tiles, each as two 4 x 32 halves:
//   wait for the prefetched halo of the half (registers) -> normalise + split + write the LDS tile -> barrier -> issue the global
//   loads of the NEXT half into registers -> 9 taps x 2 k16-steps of MFMAs, fragments requested two steps ahead, with NO barrier
//   and NO vector memory (weights resident) -> barrier -> raw output through a wave-private transpose that overlays the tile
//   (16-byte stores) -> barrier.
// Same arithmetic, same tiling for the partial statistics (one record per 8 x 32 tile) as enc_conv_kernel<8,1,1,1,...>.
#ifndef ES_ABL
#define ES_ABL 0                        // profiling ablations: 1 no MFMA loop, 2 no commit (LDS tile not written), 4 no output stores, 8 no halo loads
#endif
#ifndef ES_TRACE
#define ES_TRACE 0                      // debug build: per-wave phase cycle sums (s_memtime) written to a.out2 [blocks][4 waves][16] (u64)
#endif
#if ES_TRACE
#define ES_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tsum[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define ES_T(k) do { } while (0)
#endif
#define ES_TH 8                         // tile rows (statistics granularity); processed as two halves of ES_SH rows
#define ES_SH 4
#define ES_HH (ES_SH + 2)
#define ES_HW 34
#define ES_ROWS (ES_HH * ES_HW)                 // 204 halo pixels
#define ES_A_BYTES (ES_ROWS * EC_AS)            // 29 376
#define ES_B_BYTES (9 * 4096)                   // nine taps x (32 ch x 32 k x hi|lo) halves
#define ES_NTHR 256
__global__ __launch_bounds__(ES_NTHR, 2) void enc_conv32_stream_kernel(const EncArgs a, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) char ec_smem[];
    char* ldsA = ec_smem;
    char* ldsB = ec_smem + ES_A_BYTES;
    float* red = reinterpret_cast<float*>(ec_smem + ES_A_BYTES + ES_B_BYTES);      // [4 waves][32][2]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
    const int li = lane & 31, kg = lane >> 5;
    const int g = threadIdx.x & 3, prow0 = threadIdx.x >> 2;
    constexpr int ITEMS = (ES_ROWS * 4 + ES_NTHR - 1) / ES_NTHR;       // 4
    constexpr int RSTEP = ES_NTHR / 4;
    static_assert(4 * 32 * 36 * 4 <= ES_A_BYTES, "transpose patches overlay the activation tile");
    // all nine taps' weights: 36 pieces of 1 KiB, once per block
    for (int piece = wave; piece < 36; piece += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wpk + piece * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(ldsB + piece * 1024), 16, 0, 0);
    const float bias = a.bias ? a.bias[li] : 0.f;
    float4 raw[ITEMS][2];
    auto issue = [&](int hf) {                             // global loads of half-tile hf's halo (addresses clamped; masked at commit)
        const int t = hf >> 1;
        const int img = t / a.nblk, tile = t - img * a.nblk;
        const int ty0 = (tile / a.tiles_x) * ES_TH + (hf & 1) * ES_SH, tx0 = (tile % a.tiles_x) * 32;
        const float* src = a.src + (long)img * a.h * a.w * 32;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int row = min(prow0 + RSTEP * i, ES_ROWS - 1);
            const int hy = row / ES_HW, hx = row - hy * ES_HW;
            const int gy = min(max(ty0 + hy - 1, 0), a.h - 1), gx = min(max(tx0 + hx - 1, 0), a.w - 1);
            const float* p = src + ((long)gy * a.w + gx) * 32 + 8 * g;
            raw[i][0] = cer_ld4(p);
            raw[i][1] = cer_ld4(p + 4);
        }
    };
    int t = blockIdx.x;
    if (t < total_tiles) issue(2 * t);
    int cur_img = -1;
    float mu[8], rs[8];
#if ES_TRACE
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif
    for (; t < total_tiles; t += gridDim.x) {
        const int img = t / a.nblk, tile = t - img * a.nblk;
        const int tx0 = (tile % a.tiles_x) * 32;
        if (img != cur_img) {                              // producer statistics of this image (block-uniform branch)
            cur_img = img;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                mu[e] = a.tf ? a.tf[2 * ((long)img * 32 + 8 * g + e)] : 0.f;
                rs[e] = a.tf ? a.tf[2 * ((long)img * 32 + 8 * g + e) + 1] : 1.f;
            }
        }
        float ssum = 0.f, ssq = 0.f;
#pragma unroll 1
        for (int sub = 0; sub < 2; ++sub) {
            const int ty0 = (tile / a.tiles_x) * ES_TH + sub * ES_SH;
#if ES_TRACE
            __builtin_amdgcn_s_waitcnt(0x0F70);            // (trace build) make the halo wait its own phase
            ES_T(6);
#endif
            // ---- commit the prefetched halo: transform + split + LDS write
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int row = prow0 + RSTEP * i;
                if (row < ES_ROWS && !(ES_ABL & 2)) {
                    const int hy = row / ES_HW, hx = row - hy * ES_HW;
                    const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
                    const bool inside = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
                    const float v[8] = {raw[i][0].x, raw[i][0].y, raw[i][0].z, raw[i][0].w, raw[i][1].x, raw[i][1].y, raw[i][1].z, raw[i][1].w};
                    half8 hi, lo;
                    float xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = a.tf ? (v[e] - mu[e]) * rs[e] : v[e];
                        if (a.tf_relu) x = fmaxf(x, 0.f);
                        xv[e] = inside ? x : 0.f;
                    }
                    cer_split8(xv, hi, lo);
                    *reinterpret_cast<half8*>(ldsA + row * EC_AS + g * 16) = hi;
                    *reinterpret_cast<half8*>(ldsA + row * EC_AS + 64 + g * 16) = lo;
                }
            }
            ES_T(0);                                       // commit (incl. waiting for the prefetched halo)
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): (first half) the weight DMAs have landed
            __syncthreads();                               // tile (and weights) visible to every wave
            ES_T(1);                                       // vmcnt(0) + barrier
            {                                              // next half's halo travels during the MFMAs
                const int nh = sub == 0 ? 2 * t + 1 : 2 * (t + (int)gridDim.x);
                if (nh < 2 * total_tiles && !(ES_ABL & 8)) issue(nh);
            }
            // ---- 9 taps x 2 k16-steps, weights resident: no barrier, no vector memory.  Software pipeline in source: the
            // fragments of step i+2 are requested before the MFMAs of step i and the scheduler is fenced per step - left alone,
            // hipcc places every ds_read directly in front of the MFMA that needs it and pays the LDS latency 36 times per tile
            floatx16 accm, accl;
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[r] = bias; accl[r] = 0.f; }
            half8 fa[3][2], fb[3][2];                      // [buffer][hi|lo]
            auto load = [&](int buf, int i) {
                const int tap = i >> 1, ks = i & 1, dy = tap / 3, dx = tap - dy * 3;
                const char* p = ldsA + ((wave + dy) * ES_HW + li + dx) * EC_AS + ks * 32 + kg * 16;
                fa[buf][0] = *reinterpret_cast<const half8*>(p);
                fa[buf][1] = *reinterpret_cast<const half8*>(p + 64);
                const char* q = ldsB + tap * 4096 + (ks * 2) * 1024 + lane * 16;
                fb[buf][0] = *reinterpret_cast<const half8*>(q);
                fb[buf][1] = *reinterpret_cast<const half8*>(q + 1024);
            };
            load(0, 0);
            load(1, 1);
#pragma unroll
            for (int i = 0; i < ((ES_ABL & 1) ? 1 : 18); ++i) {
                if (i + 2 < 18) load((i + 2) % 3, i + 2);
                __builtin_amdgcn_sched_barrier(0);         // the requests stay in front of this step's MFMAs
                const int b = i % 3;
                accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[b][0], fb[b][0], accm, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[b][0], fb[b][1], accl, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[b][1], fb[b][0], accl, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            ES_T(2);                                       // halo issue + 18 MFMA steps
            __syncthreads();                               // every wave is done reading the tile: the patches may overlay it
            ES_T(3);                                       // barrier
            // ---- epilogue: statistics in the MFMA layout, raw output through the transpose
            const int gy = ty0 + wave;
            float* Et = reinterpret_cast<float*>(ldsA) + wave * (32 * 36);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = (r & 3) + 8 * (r >> 2) + 4 * kg;
                const float v = fmaf(accl[r], 1.0f / 2048.0f, accm[r]);
                Et[px * 36 + li] = v;
                if (gy < a.ho && tx0 + px < a.wo) {
                    ssum += v;
                    ssq = fmaf(v, v, ssq);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);            // the patch is wave-private
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = lane + 64 * j;
                const int px = idx >> 3, g4 = idx & 7;
                const float4 v = *reinterpret_cast<const float4*>(Et + px * 36 + 4 * g4);
                if (gy < a.ho && tx0 + px < a.wo && !(ES_ABL & 4))
                    *reinterpret_cast<float4*>(a.out + (((long)img * a.ho + gy) * a.wo + tx0 + px) * 32 + 4 * g4) = v;
            }
            if (a.part && sub == 1) {
                const float s2 = ssum + __shfl_xor(ssum, 32), q2 = ssq + __shfl_xor(ssq, 32);
                if (kg == 0) {
                    red[(wave * 32 + li) * 2 + 0] = s2;
                    red[(wave * 32 + li) * 2 + 1] = q2;
                }
            }
            ES_T(4);                                       // epilogue (transpose, stores, statistics)
            __syncthreads();                               // patches read back: the tile may be rewritten; partials visible
            ES_T(5);                                       // barrier
        }
        if (a.part && threadIdx.x < 32) {
            float s2 = 0.f, q2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s2 += red[(i * 32 + threadIdx.x) * 2 + 0];
                q2 += red[(i * 32 + threadIdx.x) * 2 + 1];
            }
            float* dst = a.part + (((long)img * a.nblk + tile) * 32 + threadIdx.x) * 2;
            dst[0] = s2;
            dst[1] = q2;
        }
    }
#if ES_TRACE
    if (lane == 0 && a.out2) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.out2) + ((long)blockIdx.x * 4 + wave) * 16;
        for (int k = 0; k < 7; ++k) o[k] = tsum[k];
        o[8] = __builtin_readcyclecounter() - tstart;
        o[9] = (unsigned long long)((total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// generic weight packing: OIHW [Cout, Cin, k, k] (k = 1 or 3) -> [chunk32][tap][ntile32][k16-step][hi|lo][lane][8]
extern "C" long cer_enc_conv_packed_size(int Cout, int Cin, int taps) {
    if (Cout <= 0 || Cin <= 0 || Cout % 32 || Cin % 32 || (taps != 1 && taps != 9)) return CER_ESHAPE;
    return (long)(Cin / 32) * taps * (Cout / 32) * 2048;
}

extern "C" int cer_enc_conv_pack(const float* w, void* packed_v, int Cout, int Cin, int taps) {
    if (!w || !packed_v) return CER_EINVAL;
    if (Cout % 32 || Cin % 32 || (taps != 1 && taps != 9)) return CER_ESHAPE;
    _Float16* packed = (_Float16*)packed_v;
    const int NT = Cout / 32;
    for (int kc = 0; kc < Cin / 32; ++kc)
        for (int tap = 0; tap < taps; ++tap)
            for (int nt = 0; nt < NT; ++nt)
                for (int ks = 0; ks < 2; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = nt * 32 + (lane & 31);
                            const int ci = kc * 32 + ks * 16 + (lane >> 5) * 8 + e;
                            float v = w[((long)co * Cin + ci) * taps + tap];
                            v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
                            const long base = ((((long)kc * taps + tap) * NT + nt) * 2 + ks) * 2;
                            packed[(base + 0) * 512 + lane * 8 + e] = hi;
                            packed[(base + 1) * 512 + lane * 8 + e] = lo;
                        }
    return CER_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int NBUF, int MINW, int TH, int STRIDE, int TAPS>
static int ec_launch(EncArgs a, int nimg, int epi, hipStream_t st) {
    constexpr int NB = WAVES_N * WN * 32;
    constexpr int KS = (TAPS == 9) ? 3 : 1;
    constexpr int HH = (TH - 1) * STRIDE + KS, HW = 31 * STRIDE + KS;
    size_t smem = (size_t)HH * HW * EC_AS + NBUF * NB * 128;
    const size_t red = (size_t)WAVES_M * NB * 2 * sizeof(float);
    if (smem < red) smem = red;
    const size_t patch = (size_t)WAVES_M * WAVES_N * 32 * (WN * 32 + 4) * sizeof(float);   // epilogue transpose patches
    if (smem < patch) smem = patch;
    a.tiles_x = (a.wo + 31) / 32;
    const int tiles_y = (a.ho + TH - 1) / TH;
    a.nblk = a.tiles_x * tiles_y;
    dim3 grid((unsigned)a.nblk, (unsigned)(a.cout / NB), (unsigned)nimg), block(64 * WAVES_M * WAVES_N);
    switch (epi) {
        case EC_EPI_RAW: hipLaunchKernelGGL((enc_conv_kernel<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, TH, STRIDE, TAPS, EC_EPI_RAW>), grid, block, smem, st, a); break;
        case EC_EPI_FMAP: hipLaunchKernelGGL((enc_conv_kernel<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, TH, STRIDE, TAPS, EC_EPI_FMAP>), grid, block, smem, st, a); break;
        case EC_EPI_CTX: hipLaunchKernelGGL((enc_conv_kernel<WAVES_M, WAVES_N, WM, WN, NBUF, MINW, TH, STRIDE, TAPS, EC_EPI_CTX>), grid, block, smem, st, a); break;
        default: return CER_EINVAL;
    }
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// Number of blocks (tiles) per image the kernel will use for an output of ho x wo - the size of the partials buffer.
extern "C" int cer_enc_conv_tiles(int ho, int wo, int stride, int taps, int Cout) {
    const int th = (stride == 2) ? 2 : 8;
    (void)taps; (void)Cout;
    return ((wo + 31) / 32) * ((ho + th - 1) / th);
}

extern "C" int cer_enc_conv_f16x3(const float* src, const float* tf_stats, int tf_relu, const void* packed_w, const float* bias, float* out,
                                  float* out2, float* stats_partial, int N, int h, int w, int Cin, int Cout, int taps, int stride, int epi,
                                  int out_border, float out_scale, void* stream) {
    if (!src || !packed_w || !out || N <= 0 || h <= 0 || w <= 0) return CER_EINVAL;
    if (Cin % 32 || Cout % 32 || (taps != 1 && taps != 9) || (stride != 1 && stride != 2)) return CER_ESHAPE;
    if (epi == EC_EPI_CTX && !out2) return CER_EINVAL;
    if (!cer_aligned16(src) || !cer_aligned16(packed_w)) return CER_EALIGN;
    EncArgs a;
    memset(&a, 0, sizeof(a));
    a.src = src;
    a.tf = tf_stats;
    a.tf_relu = tf_relu;
    a.wpk = (const _Float16*)packed_w;
    a.bias = bias;
    a.out = out;
    a.out2 = out2;
    a.part = stats_partial;
    a.h = h;
    a.w = w;
    a.cin = Cin;
    a.cout = Cout;
    const int pad = taps == 9 ? 1 : 0, ks = taps == 9 ? 3 : 1;
    a.ho = (h + 2 * pad - ks) / stride + 1;
    a.wo = (w + 2 * pad - ks) / stride + 1;
    a.out_border = out_border;
    a.out_scale = out_scale;
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1 && taps == 9 && Cin == 32 && Cout == 32 && epi == EC_EPI_RAW && ES_STREAM) {
        // layer-1 shape: persistent streaming kernel, two 4-wave blocks per CU
        a.tiles_x = (a.wo + 31) / 32;
        a.nblk = a.tiles_x * ((a.ho + ES_TH - 1) / ES_TH);
        const long total = (long)a.nblk * N;
        if (total >= (1L << 31)) return CER_ESHAPE;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        const size_t smem = ES_A_BYTES + ES_B_BYTES + 4 * 32 * 2 * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)enc_conv32_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                return CER_EINVAL;
            attr_set = true;
        }
        hipLaunchKernelGGL(enc_conv32_stream_kernel, dim3((unsigned)(total < 2 * cus ? total : 2 * cus)), dim3(ES_NTHR), smem, st, a, (int)total);
        CER_RETURN_IF_LAUNCH_FAILED();
        return CER_OK;
    }
    if (stride == 1 && taps == 9) {
        if (Cout % 64 == 0) return ec_launch<8, 1, 1, 2, 3, 4, 8, 1, 9>(a, N, epi, st);       // 8 x 32 px x 64 ch per block
        return ec_launch<8, 1, 1, 1, 3, 4, 8, 1, 9>(a, N, epi, st);                            // 8 x 32 px x 32 ch
    }
    if (stride == 1 && taps == 1) {
        if (Cout % 64 == 0) return ec_launch<8, 1, 1, 2, 2, 4, 8, 1, 1>(a, N, epi, st);
        return ec_launch<8, 1, 1, 1, 2, 4, 8, 1, 1>(a, N, epi, st);
    }
    if (Cout % 64) return CER_ESHAPE;
    if (taps == 9) return ec_launch<2, 2, 1, 1, 3, 2, 2, 2, 9>(a, N, epi, st);                  // stride 2: 2 x 32 px x 64 ch
    return ec_launch<2, 2, 1, 1, 2, 2, 2, 2, 1>(a, N, epi, st);
}

// ---- statistics reduce: partials [N][nblk][C][2] -> stats [N*C][2] = (mean, rstd), fp64 accumulation ----------
__global__ __launch_bounds__(256) void enc_stats_reduce_kernel(const float* __restrict__ part, float* __restrict__ stats, int nblk, int C,
                                                               double count, float eps) {
    const int n = blockIdx.x / C, c = blockIdx.x % C;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float* p = part + (((long)n * nblk + b) * C + c) * 2;
        s += p[0];
        q += p[1];
    }
    __shared__ double sh[2][4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = s;
        sh[1][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double S = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3], Q = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const double mean = S / count;
        double var = Q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * blockIdx.x] = (float)mean;
        stats[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

extern "C" int cer_enc_stats_reduce_f32(const float* partial, float* stats, int N, int nblk, int C, long pixels, float eps, void* stream) {
    if (!partial || !stats || N <= 0 || nblk <= 0 || C <= 0 || pixels <= 0) return CER_EINVAL;
    hipLaunchKernelGGL(enc_stats_reduce_kernel, dim3((unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream, partial, stats, nblk, C,
                       (double)pixels, eps);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// ---- residual merge, channels-last: out = relu( fa(a) + fb(b) ), f = optional instance norm (+ ReLU) ------------
// flags: 1 relu on a, 2 relu on b, 4 relu on the sum
__global__ __launch_bounds__(256) void enc_merge_kernel(const float* __restrict__ xa, const float* __restrict__ sa, const float* __restrict__ xb,
                                                        const float* __restrict__ sb, float* __restrict__ out, long per_image, int C, int flags,
                                                        long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        const int n = (int)(e / per_image), c = (int)(e % C);
        const float4 va = cer_ld4(xa + e);
        float a4[4] = {va.x, va.y, va.z, va.w}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (xb) {
            const float4 vb = cer_ld4(xb + e);
            b4[0] = vb.x; b4[1] = vb.y; b4[2] = vb.z; b4[3] = vb.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = a4[k];
            if (sa) t = (t - sa[2 * (n * C + c + k)]) * sa[2 * (n * C + c + k) + 1];
            if (flags & 1) t = fmaxf(t, 0.f);
            if (xb) {
                float u = b4[k];
                if (sb) u = (u - sb[2 * (n * C + c + k)]) * sb[2 * (n * C + c + k) + 1];
                if (flags & 2) u = fmaxf(u, 0.f);
                t += u;
            }
            if (flags & 4) t = fmaxf(t, 0.f);
            a4[k] = t;
        }
        *reinterpret_cast<float4*>(out + e) = make_float4(a4[0], a4[1], a4[2], a4[3]);
    }
}

extern "C" int cer_enc_merge_f32(const float* a, const float* a_stats, const float* b, const float* b_stats, float* out, int N, long pixels,
                                 int C, int flags, void* stream) {
    if (!a || !out || N <= 0 || pixels <= 0 || C <= 0) return CER_EINVAL;
    if (C % 4) return CER_ESHAPE;
    if (!cer_aligned16(a) || !cer_aligned16(out) || (b && !cer_aligned16(b))) return CER_EALIGN;
    const long total4 = (long)N * pixels * C / 4;
    long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(enc_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, a_stats, b, b_stats, out, pixels * C, C,
                       flags, total4);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// ---- stem: 7x7 stride-2 pad-3 convolution 3 -> 32 on the raw image (core/extractor.py:81,145; core/raft.py:40-41) ----
// NCHW image in 0..255 -> x*(2/255) - 1 on the fly -> channels-last raw output [N, ho*wo, 32] + stats partials.
// Direct fp32: one thread = one output pixel x 32 channels; weights [147][32] through the scalar cache.
#define ST_PIX 256                      // threads per block; each thread owns TWO horizontally adjacent output pixels
typedef float st_float2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(ST_PIX) void enc_stem_kernel(const float* __restrict__ img, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                          float* __restrict__ out, float* __restrict__ part, int H, int W, int ho, int wo,
                                                          int nblk, int normalize) {
    __shared__ float red[4][32][2];
    const int n = blockIdx.y;
    const int wo2 = (wo + 1) / 2;                          // pixel pairs per output row
    const long q = (long)blockIdx.x * ST_PIX + threadIdx.x;
    const long Po = (long)ho * wo;
    const bool valid0 = q < (long)ho * wo2;
    const int oy = valid0 ? (int)(q / wo2) : 0, ox = valid0 ? 2 * (int)(q % wo2) : 0;
    const bool valid1 = valid0 && ox + 1 < wo;
    float acc0[32], acc1[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc0[c] = acc1[c] = bias[c];
    const float* im = img + (long)n * 3 * H * W;
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = oy * 2 + ky - 3;
            if (iy < 0 || iy >= H) continue;
            // the two pixels' 7-tap windows overlap in 5 columns: 9 loads feed 14 taps, and every weight vector is used twice
            float xr[9];
            const float* row = im + ((long)ci * H + iy) * W;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int ix = ox * 2 + j - 3;
                float x = 0.f;
                if (ix >= 0 && ix < W) {
                    x = row[ix];
                    if (normalize) x = x * (2.0f / 255.0f) - 1.0f;
                }
                xr[j] = x;
            }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                // wave-uniform address: the 32 weights of this tap arrive through the scalar cache (s_load), no LDS traffic
                const float* wr = wgt + ((ci * 7 + ky) * 7 + kx) * 32;
                const st_float2 x0 = {xr[kx], xr[kx]}, x1 = {xr[kx + 2], xr[kx + 2]};
#pragma unroll
                for (int c = 0; c < 16; ++c) {               // v_pk_fma_f32: each component is the same fused multiply-add as fmaf
                    const st_float2 ww = {wr[2 * c], wr[2 * c + 1]};
                    st_float2 a = {acc0[2 * c], acc0[2 * c + 1]}, b = {acc1[2 * c], acc1[2 * c + 1]};
                    a = __builtin_elementwise_fma(x0, ww, a);
                    b = __builtin_elementwise_fma(x1, ww, b);
                    acc0[2 * c] = a.x; acc0[2 * c + 1] = a.y;
                    acc1[2 * c] = b.x; acc1[2 * c + 1] = b.y;
                }
            }
        }
    if (valid0) {
        float* o = out + ((long)n * Po + (long)oy * wo + ox) * 32;
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(o + 4 * k) = make_float4(acc0[4 * k], acc0[4 * k + 1], acc0[4 * k + 2], acc0[4 * k + 3]);
        if (valid1) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<float4*>(o + 32 + 4 * k) = make_float4(acc1[4 * k], acc1[4 * k + 1], acc1[4 * k + 2], acc1[4 * k + 3]);
        }
    }
    if (part) {
        // 64 values per lane (32 sums, 32 sums of squares) -> butterfly: 63 exchanges leave the wave total of value l on lane l
        // (a plain 6-step shuffle reduction per value was 768 ds_bpermute per wave and dominated the kernel)
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
        float v[64];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float a = valid0 ? acc0[c] : 0.f, b = valid1 ? acc1[c] : 0.f;
            v[c] = a + b;
            v[32 + c] = a * a + b * b;
        }
#pragma unroll
        for (int hb = 32; hb >= 1; hb >>= 1) {
            const bool up = (lane & hb) != 0;
#pragma unroll
            for (int i = 0; i < hb; ++i) {
                const float send = up ? v[i] : v[i + hb];
                const float keep = up ? v[i + hb] : v[i];
                v[i] = keep + __shfl_xor(send, hb);
            }
        }
        red[wave][lane & 31][lane >> 5] = v[0];           // lane l: l < 32 -> sum of channel l, else sum of squares of channel l-32
        __syncthreads();
        if (threadIdx.x < 32) {
            const int c = threadIdx.x;
            float* dst = part + (((long)n * nblk + blockIdx.x) * 32 + c) * 2;
            dst[0] = red[0][c][0] + red[1][c][0] + red[2][c][0] + red[3][c][0];
            dst[1] = red[0][c][1] + red[1][c][1] + red[2][c][1] + red[3][c][1];
        }
    }
}

extern "C" int cer_enc_stem_tiles(int ho, int wo) { return (int)(((long)ho * ((wo + 1) / 2) + ST_PIX - 1) / ST_PIX); }

// wgt: [3*7*7][32] (ci, ky, kx major; output channel minor), device pointer
extern "C" int cer_enc_stem_f32(const float* images, const float* wgt_k_co, const float* bias, float* out, float* stats_partial, int N, int H,
                                int W, int normalize, void* stream) {
    if (!images || !wgt_k_co || !bias || !out || N <= 0 || H <= 0 || W <= 0) return CER_EINVAL;
    if (N > 65535) return CER_ESHAPE;
    const int ho = (H + 6 - 7) / 2 + 1, wo = (W + 6 - 7) / 2 + 1;
    const int nblk = cer_enc_stem_tiles(ho, wo);
    hipLaunchKernelGGL(enc_stem_kernel, dim3((unsigned)nblk, (unsigned)N), dim3(ST_PIX), 0, (hipStream_t)stream, images, wgt_k_co, bias, out,
                       stats_partial, H, W, ho, wo, nblk, normalize);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
