// K3, wave-specialised variant (experimental, round 1): the same split-f16 implicit-GEMM 3x3 convolution as
// gru_f16x3.hip (reference: core/update.py:13-25,61-71), restructured so that the matrix waves never touch vector
// memory inside the K loop.
//
// One persistent 12-wave block per CU:
//   waves 0-7   matrix waves: 4 x 2 grid of 32 px x NB/2 ch tiles; per step: s_barrier, ds_read_b128 fragments, MFMAs
//   waves 8-9   weight loaders: global -> LDS DMA ring (NSLOT steps deep), counted vmcnt
//   waves 10-11 activation stagers: next chunk's 6 x 34 halo global -> registers (issued at tap 0), split to hi|lo f16
//               and written to the OTHER activation buffer at tap 3; double buffered, runs across work items
// Every wave executes exactly one s_barrier per (chunk, tap) step; the barrier of step g publishes the weights of step
// g and (at tap 0) the activation chunk, and frees the ring slot / activation buffer read in step g-1.
#include "../common.hpp"
#include <string.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define WS_TH 4
#define WS_TW 32
#define WS_HH 6
#define WS_HW 34
#define WS_ROWS (WS_HH * WS_HW)
#define WS_KC 32
#define WS_AS 144
#define WS_A_BYTES (WS_ROWS * WS_AS)
#define WS_STAGERS 128                  // threads staging activations (2 waves)
#define WS_ITEMS ((WS_ROWS * 4 + WS_STAGERS - 1) / WS_STAGERS)
#ifndef WS_ABL
#define WS_ABL 0                        // profiling ablations (compile time): 1 no MFMA, 2 no epilogue stores, 4 no LDS reads in the loop
#endif

struct WsArgs {
    const float* src[CER_CONV_MAX_SRC];
    int ch[CER_CONV_MAX_SRC];
    int nsrc;
    const _Float16* wpk;
    const float* bias;
    float* out;
    int h, w, cout;
    int tiles_x, ntiles, nitems, nchunks;
};

__device__ __forceinline__ void ws_split(float v, _Float16& hi, _Float16& lo) {
    const float x = fminf(fmaxf(v, -65504.0f), 65504.0f);
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * 2048.0f);
}

// chunk index within an item -> (source, first channel)
__device__ __forceinline__ void ws_chunk_src(const WsArgs& a, int c, int& s, int& c0) {
    s = 0;
    c0 = c * WS_KC;
    while (c0 >= a.ch[s]) { c0 -= a.ch[s]; ++s; }
}

template <int NB, int NSLOT, int EPI>
__global__ __launch_bounds__(768) void conv3x3_ws_kernel(const WsArgs a) {
    constexpr int B_BYTES = NB * 128;
    constexpr int WN = NB / 64;                            // 32-channel tiles per matrix wave
    extern __shared__ __attribute__((aligned(16))) char ws_smem[];
    char* ldsA = ws_smem;                                  // 2 activation buffers
    char* ldsB = ws_smem + 2 * WS_A_BYTES;                 // NSLOT weight slots
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int C = a.nchunks, S = 9 * C;
    const int my_items = (a.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // items of this block
    const int NT = a.cout / 32;
    const long total_steps = (long)my_items * S;
    if (my_items <= 0) return;

    if (wave < 8) {
        // ------------------------------------------------------------------ matrix waves
        const int wm = wave >> 1, wn = wave & 1;
        const int li = lane & 31, kg = lane >> 5;
        long g = 0;
        int gc = 0;
        for (int it = 0; it < my_items; ++it) {
            const int item = blockIdx.x + it * gridDim.x;
            const int tile = item % a.ntiles, by = item / a.ntiles;
            const int ty0 = (tile / a.tiles_x) * WS_TH, tx0 = (tile % a.tiles_x) * WS_TW;
            const int nb0 = by * NB;
            floatx16 accm[WN], accl[WN];
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const float b = a.bias ? a.bias[nb0 + (wn * WN + n) * 32 + li] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { accm[n][r] = b; accl[n][r] = 0.f; }
            }
            for (int c = 0; c < C; ++c, ++gc) {
                const char* A = ldsA + (gc & 1) * WS_A_BYTES;
#pragma unroll 1
                for (int tap = 0; tap < 9; ++tap, ++g) {
                    __builtin_amdgcn_s_barrier();
                    const char* B = ldsB + (int)(g % NSLOT) * B_BYTES;
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const int row = (wm + dy) * WS_HW + li + dx;
                    half8 ah[2], al[2], bh[2][WN], bl[2][WN];
                    if (WS_ABL & 4) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { ah[ks][e] = (_Float16)(float)(lane + tap); al[ks][e] = (_Float16)1; }
#pragma unroll
                            for (int n = 0; n < WN; ++n)
#pragma unroll
                                for (int e = 0; e < 8; ++e) { bh[ks][n][e] = (_Float16)(float)lane; bl[ks][n][e] = (_Float16)2; }
                        }
                    } else
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const char* p = A + row * WS_AS + ks * 32 + kg * 16;
                        ah[ks] = *reinterpret_cast<const half8*>(p);
                        al[ks] = *reinterpret_cast<const half8*>(p + 64);
#pragma unroll
                        for (int n = 0; n < WN; ++n) {
                            const char* q = B + (((wn * WN + n) * 2 + ks) * 2) * 1024 + lane * 16;
                            bh[ks][n] = *reinterpret_cast<const half8*>(q);
                            bl[ks][n] = *reinterpret_cast<const half8*>(q + 1024);
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int n = 0; n < WN; ++n) {
                            if (WS_ABL & 1) {
#if defined(__HIP_DEVICE_COMPILE__)
                                asm volatile("" ::"v"(ah[ks]), "v"(al[ks]), "v"(bh[ks][n]), "v"(bl[ks][n]));
#endif
                                continue;
                            }
                            accm[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks][n], accm[n], 0, 0, 0);
                            accl[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks][n], accl[n], 0, 0, 0);
                            accl[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks][n], accl[n], 0, 0, 0);
                        }
                }
            }
            // ---- epilogue: lane holds channel co, pixels x = tx0 + (r&3) + 8*(r>>2) + 4*kg of row gy
            const int gy = ty0 + wm;
            if (WS_ABL & 2) {
#if defined(__HIP_DEVICE_COMPILE__)
                for (int n = 0; n < WN; ++n) asm volatile("" ::"v"(accm[n]), "v"(accl[n]));
#endif
            } else if (gy < a.h) {
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    const int co = nb0 + (wn * WN + n) * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gx = tx0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        if (gx >= a.w) continue;
                        float v = fmaf(accl[n][r], 1.0f / 2048.0f, accm[n][r]);
                        if (EPI == CER_EPI_RELU) v = fmaxf(v, 0.f);
                        a.out[((long)gy * a.w + gx) * a.cout + co] = v;
                    }
                }
            }
        }
    } else if (wave < 10) {
        // ------------------------------------------------------------------ weight loaders (DMA ring)
        constexpr int DPW = NB / 8 / 2;                    // 1-KiB DMA pieces per loader wave per step
        const int lw = wave - 8;
        auto issue = [&](long G) {
            const int it = (int)(G / S), s = (int)(G % S);
            const int item = blockIdx.x + it * gridDim.x;
            const int by = item / a.ntiles;
            const _Float16* slice = a.wpk + ((long)s * NT + by * (NB / 32)) * 2048;
            char* dst = ldsB + (int)(G % NSLOT) * B_BYTES;
#pragma unroll
            for (int i = 0; i < DPW; ++i) {
                const int piece = lw * DPW + i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slice + piece * 512 + lane * 8),
                                                 (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
            }
        };
        for (long G = 0; G < NSLOT - 1 && G < total_steps; ++G) issue(G);
        for (long g = 0; g < total_steps; ++g) {
            // steps issued so far: <= g + NSLOT - 2; the DMAs of the younger steps may stay in flight
            const long younger = min((long)(NSLOT - 2), total_steps - 1 - g);
            if (younger >= NSLOT - 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (((NSLOT - 2) * DPW) & 15) | ((((NSLOT - 2) * DPW) >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            if (g + NSLOT - 1 < total_steps) issue(g + NSLOT - 1);
        }
    } else {
        // ------------------------------------------------------------------ activation stagers
        const int t = threadIdx.x - 640;                   // 0..127
        float4 raw[WS_ITEMS][2];
        auto load_chunk = [&](int gcn) {                   // global chunk index -> issue the global loads
            const int it = gcn / C, c = gcn - it * C;
            const int item = blockIdx.x + it * gridDim.x;
            const int tile = item % a.ntiles;
            const int ty0 = (tile / a.tiles_x) * WS_TH, tx0 = (tile % a.tiles_x) * WS_TW;
            int s, c0;
            ws_chunk_src(a, c, s, c0);
#pragma unroll
            for (int i = 0; i < WS_ITEMS; ++i) {
                const int idx = min(t + WS_STAGERS * i, WS_ROWS * 4 - 1);
                const int row = idx >> 2, g8 = idx & 3;
                const int hy = row / WS_HW, hx = row - hy * WS_HW;
                const int gy = min(max(ty0 + hy - 1, 0), a.h - 1), gx = min(max(tx0 + hx - 1, 0), a.w - 1);
                const float* p = a.src[s] + ((long)gy * a.w + gx) * a.ch[s] + c0 + 8 * g8;
                raw[i][0] = cer_ld4(p);
                raw[i][1] = cer_ld4(p + 4);
            }
        };
        auto store_chunk = [&](int gcn) {                  // split + write into buffer gcn & 1 (zero padding outside the image)
            const int it = gcn / C;
            const int item = blockIdx.x + it * gridDim.x;
            const int tile = item % a.ntiles;
            const int ty0 = (tile / a.tiles_x) * WS_TH, tx0 = (tile % a.tiles_x) * WS_TW;
            char* A = ldsA + (gcn & 1) * WS_A_BYTES;
#pragma unroll
            for (int i = 0; i < WS_ITEMS; ++i) {
                const int idx = t + WS_STAGERS * i;
                if (idx < WS_ROWS * 4) {
                    const int row = idx >> 2, g8 = idx & 3;
                    const int hy = row / WS_HW, hx = row - hy * WS_HW;
                    const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
                    const float m = (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) ? 1.0f : 0.0f;
                    const float v[8] = {raw[i][0].x * m, raw[i][0].y * m, raw[i][0].z * m, raw[i][0].w * m,
                                        raw[i][1].x * m, raw[i][1].y * m, raw[i][1].z * m, raw[i][1].w * m};
                    half8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        _Float16 hh, ll;
                        ws_split(v[e], hh, ll);
                        hi[e] = hh;
                        lo[e] = ll;
                    }
                    *reinterpret_cast<half8*>(A + row * WS_AS + g8 * 16) = hi;
                    *reinterpret_cast<half8*>(A + row * WS_AS + 64 + g8 * 16) = lo;
                }
            }
        };
        const int total_chunks = my_items * C;
        load_chunk(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
        store_chunk(0);
        for (int gcn = 0; gcn < total_chunks; ++gcn) {
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): chunk gcn is completely written
            __builtin_amdgcn_s_barrier();                  // tap 0: publishes chunk gcn; buffer (gcn+1)&1 is free from here
            const bool more = gcn + 1 < total_chunks;
            if (more) load_chunk(gcn + 1);
            __builtin_amdgcn_s_barrier();                  // tap 1
            __builtin_amdgcn_s_barrier();                  // tap 2
            __builtin_amdgcn_s_barrier();                  // tap 3
            if (more) {
                __builtin_amdgcn_s_waitcnt(0x0F70);
                store_chunk(gcn + 1);
            }
            __builtin_amdgcn_s_barrier();                  // tap 4
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();                  // tap 8
        }
    }
}

template <int NB, int NSLOT>
static int ws_launch(const WsArgs& a, int epi, hipStream_t st) {
    const size_t smem = 2 * WS_A_BYTES + (size_t)NSLOT * NB * 128;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    dim3 grid((unsigned)(a.nitems < cus ? a.nitems : cus)), block(768);
    auto kern = epi == CER_EPI_RELU ? conv3x3_ws_kernel<NB, NSLOT, CER_EPI_RELU> : conv3x3_ws_kernel<NB, NSLOT, CER_EPI_LINEAR>;
    static bool attr_set[2] = {false, false};
    const int ai = epi == CER_EPI_RELU ? 1 : 0;
    if (!attr_set[ai]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return CER_EINVAL;
        attr_set[ai] = true;
    }
    hipLaunchKernelGGL(kern, grid, block, smem, st, a);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// Experimental entry point (not yet part of include/cer_mvs.h): tensor sources only (kind 0), LINEAR / RELU epilogues,
// weights in cer_conv3x3_f16x3_pack order.
extern "C" int cer_conv3x3_ws_f16x3(const cer_conv_inputs* in, const void* packed_w, const float* bias, float* out, int h, int w, int Cout,
                                    int epi, void* stream) {
    if (!in || !packed_w || !out || h <= 0 || w <= 0 || Cout <= 0) return CER_EINVAL;
    if (in->nsrc <= 0 || in->nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    if (epi != CER_EPI_LINEAR && epi != CER_EPI_RELU) return CER_EINVAL;
    if (Cout % 64 != 0) return CER_ESHAPE;
    WsArgs a;
    memset(&a, 0, sizeof(a));
    a.nsrc = in->nsrc;
    int chunks = 0;
    for (int s = 0; s < in->nsrc; ++s) {
        if (!in->src[s]) return CER_EINVAL;
        if (in->kind[s] != 0 || in->ch[s] % WS_KC != 0) return CER_ESHAPE;
        if (!cer_aligned16(in->src[s])) return CER_EALIGN;
        a.src[s] = in->src[s];
        a.ch[s] = in->ch[s];
        chunks += in->ch[s] / WS_KC;
    }
    if (!cer_aligned16(packed_w)) return CER_EALIGN;
    a.wpk = (const _Float16*)packed_w;
    a.bias = bias;
    a.out = out;
    a.h = h;
    a.w = w;
    a.cout = Cout;
    a.tiles_x = (w + WS_TW - 1) / WS_TW;
    a.ntiles = a.tiles_x * ((h + WS_TH - 1) / WS_TH);
    a.nchunks = chunks;
    hipStream_t st = (hipStream_t)stream;
    if (Cout % 128 == 0) {
        a.nitems = a.ntiles * (Cout / 128);
        return ws_launch<128, 5>(a, epi, st);
    }
    a.nitems = a.ntiles * (Cout / 64);
    return ws_launch<64, 8>(a, epi, st);
}
