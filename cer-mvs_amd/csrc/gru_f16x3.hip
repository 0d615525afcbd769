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
// Structure (per block: 4 waves, 4 x 32 pixel tile x NB output channels):
//   * K loop over 32-channel chunks of the concatenated sources; per chunk the 6 x 34 halo is read once
//     (fp32, coalesced), split to hi|lo f16 and stored to LDS with a 144-B pixel stride (conflict-free
//     ds_read_b128 A fragments for every tap: 32 consecutive pixels x 16 B hit 16 distinct 16-B slots);
//   * weights are pre-split and pre-packed on the host in B-fragment order; per (chunk, tap) the block's
//     NB/8 KiB slice is DMA'd global->LDS with global_load_lds_dwordx4 (lane-linear image == fragment
//     order), double-buffered: tap t+1 streams in while tap t is multiplied;
//   * per tap and k16-step a wave reads its A hi/lo and B hi/lo fragments (ds_read_b128) and issues
//     3 MFMAs per 32x32 output tile;
//   * gate math in the epilogue, identical to the fp32 kernel.
#include "common.hpp"
#include <string.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define HX_TH 4
#define HX_TW 32
#define HX_HH (HX_TH + 2)
#define HX_HW (HX_TW + 2)
#define HX_ROWS (HX_HH * HX_HW)        // 204 halo pixels
#define HX_KC 32                       // channels per chunk
#define HX_AS 144                      // LDS bytes per halo pixel: 32 hi | 32 lo | 16 pad
#define HX_A_BYTES (HX_ROWS * HX_AS)   // 29376

struct ConvArgsX {
    const float* src[CER_CONV_MAX_SRC];
    int ch[CER_CONV_MAX_SRC];
    int chpad[CER_CONV_MAX_SRC];
    int kind[CER_CONV_MAX_SRC];
    int nsrc;
    const _Float16* wpk;
    const float* bias;
    const float* init;
    float* out;
    float* out2;
    const float* aux;
    const float* aux2;
    int h, w, cout;
    int tiles_x;
};

__device__ __forceinline__ float hx_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void hx_split(float v, _Float16& hi, _Float16& lo) {
    const float x = fminf(fmaxf(v, -65504.0f), 65504.0f);
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * 2048.0f);
}

// stage one 32-channel chunk of source s (channel offset c0) for the tile at (ty0, tx0): fp32 -> hi|lo f16
__device__ __forceinline__ void hx_stage(char* __restrict__ ldsA, const ConvArgsX& a, int s, int c0, int ty0, int tx0) {
    const int kind = a.kind[s];
    for (int idx = threadIdx.x; idx < HX_ROWS * 4; idx += 256) {
        const int row = idx >> 2, g = idx & 3;
        const int hy = row / HX_HW, hx = row - hy * HX_HW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool inside = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (inside) {
            if (kind == 0) {
                const float* p = a.src[s] + ((long)gy * a.w + gx) * a.ch[s] + c0 + 8 * g;
                const float4 lo4 = cer_ld4(p), hi4 = cer_ld4(p + 4);
                v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w;
                v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
            } else {
                const float* d = a.src[s];
                const float ctr = d[(long)gy * a.w + gx];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = c0 + 8 * g + i;
                    if (c < 49) {
                        const int uy = c / 7, ux = c - uy * 7;
                        const int yy = gy + uy - 3, xx = gx + ux - 3;
                        const float nb = (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) ? d[(long)yy * a.w + xx] : 0.f;
                        v[i] = 100.0f * (nb - ctr);
                    }
                }
            }
        }
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            _Float16 h, l;
            hx_split(v[i], h, l);
            hi[i] = h;
            lo[i] = l;
        }
        *reinterpret_cast<half8*>(ldsA + row * HX_AS + g * 16) = hi;
        *reinterpret_cast<half8*>(ldsA + row * HX_AS + 64 + g * 16) = lo;
    }
}

// DMA the block's weight slice of (chunk, tap) into an LDS buffer: NB/8 KiB = NB/8 pieces of 1 KiB
template <int NB>
__device__ __forceinline__ void hx_issue_B(char* __restrict__ ldsB, const _Float16* __restrict__ slice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NB / 32; ++i) {
        const int piece = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slice + piece * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(ldsB + piece * 1024), 16, 0, 0);
    }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_f16x3_kernel(const ConvArgsX a) {
    static_assert(WAVES_M * WAVES_N == 4 && WAVES_M * WM == HX_TH, "tile config");
    constexpr int NB = WAVES_N * WN * 32;                  // output channels per block
    constexpr int B_BYTES = NB * 128;                      // per (chunk, tap): NB/32 n-tiles x 4 KiB
    extern __shared__ __attribute__((aligned(16))) char hx_smem[];
    char* ldsA = hx_smem;
    char* ldsB = hx_smem + HX_A_BYTES;                     // two buffers of B_BYTES

    const int tile = blockIdx.x;
    const int ty0 = (tile / a.tiles_x) * HX_TH, tx0 = (tile % a.tiles_x) * HX_TW;
    const int nb0 = blockIdx.y * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, kg = lane >> 5;
    const int NT = a.cout / 32;

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
            if (a.init) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gx = tx0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (gy < a.h && gx < a.w) v[r] = a.init[((long)gy * a.w + gx) * a.cout + co];
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

    int chunk = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        for (int c0 = 0; c0 < a.chpad[s]; c0 += HX_KC, ++chunk) {
            const _Float16* wchunk = a.wpk + ((long)chunk * 9 * NT + nb0 / 32) * 2048;    // halves: 4 KiB per n-tile
            __syncthreads();                               // previous chunk fully consumed (A and both B buffers)
            hx_stage(ldsA, a, s, c0, ty0, tx0);
            hx_issue_B<NB>(ldsB, wchunk);
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of B[tap] has landed in LDS
                __syncthreads();                           // everyone's share landed, A visible, tap-1 consumed
                if (tap < 8) hx_issue_B<NB>(ldsB + ((tap + 1) & 1) * B_BYTES, wchunk + (long)(tap + 1) * NT * 2048);
                const char* B = ldsB + (tap & 1) * B_BYTES;
                const int dy = tap / 3, dx = tap - dy * 3;
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
                            accm[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], accm[m][n], 0, 0, 0);
                            accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], accl[m][n], 0, 0, 0);
                            accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], accl[m][n], 0, 0, 0);
                        }
                }
            }
        }
    }

    // ---- epilogue: lane holds channel co, pixels x = tx0 + (r&3) + 8*(r>>2) + 4*kg of row gy
    const int half = a.cout / 2;
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
                } else {   // CER_EPI_GRU
                    const float q = tanhf(v);
                    const float z = a.aux2[pix * a.cout + co], hprev = a.aux[pix * a.cout + co];
                    a.out[pix * a.cout + co] = (1.0f - z) * hprev + z * q;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- host side

static int hx_padded_channels(int ch, int kind) { return kind == 1 ? 64 : ((ch + HX_KC - 1) / HX_KC) * HX_KC; }

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

template <int WAVES_M, int WAVES_N, int WM, int WN>
static int hx_launch(const ConvArgsX& a, int epi, int nby, hipStream_t st) {
    constexpr int NB = WAVES_N * WN * 32;
    const size_t smem = HX_A_BYTES + 2 * NB * 128;
    const int tiles_y = (a.h + HX_TH - 1) / HX_TH;
    dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)nby);
    switch (epi) {
        case CER_EPI_LINEAR: hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, CER_EPI_LINEAR>), grid, dim3(256), smem, st, a); break;
        case CER_EPI_RELU: hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, CER_EPI_RELU>), grid, dim3(256), smem, st, a); break;
        case CER_EPI_GATES: hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, CER_EPI_GATES>), grid, dim3(256), smem, st, a); break;
        case CER_EPI_GRU: hipLaunchKernelGGL((conv3x3_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, CER_EPI_GRU>), grid, dim3(256), smem, st, a); break;
        default: return CER_EINVAL;
    }
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_conv3x3_f16x3(const cer_conv_inputs* in, const void* packed_w, const float* bias, const float* init, float* out,
                                 float* out2, const float* aux, const float* aux2, int h, int w, int Cout, int epi, void* stream) {
    if (!in || !packed_w || !out || h <= 0 || w <= 0 || Cout <= 0) return CER_EINVAL;
    if (in->nsrc <= 0 || in->nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    if (epi == CER_EPI_GATES && (!out2 || !aux)) return CER_EINVAL;
    if (epi == CER_EPI_GRU && (!aux || !aux2)) return CER_EINVAL;
    if (Cout % 64 != 0) return CER_ESHAPE;
    ConvArgsX a;
    memset(&a, 0, sizeof(a));
    a.nsrc = in->nsrc;
    for (int s = 0; s < in->nsrc; ++s) {
        if (!in->src[s]) return CER_EINVAL;
        if (in->kind[s] == 0 && (in->ch[s] % HX_KC != 0)) return CER_ESHAPE;
        if (in->kind[s] == 1 && in->ch[s] != 49) return CER_ESHAPE;
        if (in->kind[s] == 0 && !cer_aligned16(in->src[s])) return CER_EALIGN;
        a.src[s] = in->src[s];
        a.ch[s] = in->ch[s];
        a.kind[s] = in->kind[s];
        a.chpad[s] = hx_padded_channels(in->ch[s], in->kind[s]);
    }
    if (!cer_aligned16(packed_w)) return CER_EALIGN;
    a.wpk = (const _Float16*)packed_w;
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
    hipStream_t st = (hipStream_t)stream;
    if (Cout % 128 == 0) return hx_launch<2, 2, 2, 2>(a, epi, Cout / 128, st);
    return hx_launch<4, 1, 1, 2>(a, epi, Cout / 64, st);
}
