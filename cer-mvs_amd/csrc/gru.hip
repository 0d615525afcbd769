// K3: the update block's 3x3 convolutions as implicit GEMMs on exact-fp32 MFMA
// (reference: core/update.py:13-25 ConvGRU, :61-71 corr_encoder/delta heads, :80-85 disp_encoder,
//  :87-120 UpdateBlock.forward).  Channels-last activations [h*w, C].
//
// GEMM view: M = pixels, N = output channels, K = 9 taps x (concatenated, padded) input channels.
//   * block = 256 threads (4 waves) owns an 8 x 16 pixel tile (8 M-tiles of 16 pixels: one image
//     row segment each) and NB = 64 or 128 output channels;
//   * per 32-channel K chunk the (8+2) x (16+2) halo of the input is staged once into LDS
//     (row stride 40 floats: the ds_read_b128 A-fragment reads of 16 consecutive pixels are then
//     bank-conflict free for every tap offset) and reused by all 9 taps;
//   * v_mfma_f32_16x16x4_f32: A[pixel = lane&15][k = lane>>4], B[k = lane>>4][cout = lane&15]; one
//     16-B LDS read gives a lane its A operand for 4 consecutive MFMAs (channels 4*(lane>>4)+s),
//     one 16-B global read of the pre-packed weights gives the matching B operands (weights are
//     packed per (16-channel chunk, tap, n-tile) as 64 lanes x float4 = 1 KiB, fully coalesced and
//     L2 resident: <= 1.2 MB per conv);
//   * the 49-channel disparity encoder (unfold 7x7 minus centre, x100) is never materialised: it
//     is generated while staging (source kind 1);
//   * gate math (sigmoid / tanh / GRU blend / ReLU) runs in the MFMA epilogue on the accumulators.
#include "common.hpp"
#include <string.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CV_TH 8
#define CV_TW 16
#define CV_HH (CV_TH + 2)
#define CV_HW (CV_TW + 2)
#define CV_ROWS (CV_HH * CV_HW)      // 180 halo pixels
#define CV_KC 32                     // channels per staged chunk
#define CV_LS 40                     // LDS floats per halo pixel (32 + 8 pad)

struct ConvArgs {
    const float* src[CER_CONV_MAX_SRC];
    int ch[CER_CONV_MAX_SRC];        // real channel stride of the source tensor
    int chpad[CER_CONV_MAX_SRC];     // channels rounded up to 32 (64 for kind 1)
    int kind[CER_CONV_MAX_SRC];
    int nsrc;
    const float* wpk;
    const float* bias;
    const float* init;
    float* out;
    float* out2;
    const float* aux;
    const float* aux2;
    int h, w, cout;                  // cout = total output channels of the conv
    int tiles_x;
};

__device__ __forceinline__ float cv_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// stage one 32-channel chunk of source s (channel offset c0) for the tile at (ty0, tx0)
__device__ __forceinline__ void cv_stage(float* __restrict__ lds, const ConvArgs& a, int s, int c0, int ty0, int tx0) {
    const int kind = a.kind[s];
    for (int idx = threadIdx.x; idx < CV_ROWS * (CV_KC / 4); idx += 256) {
        const int row = idx >> 3, q = idx & 7;
        const int hy = row / CV_HW, hx = row - hy * CV_HW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool inside = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inside) {
            if (kind == 0) {
                v = cer_ld4(a.src[s] + ((long)gy * a.w + gx) * a.ch[s] + c0 + 4 * q);
            } else {
                const float* d = a.src[s];
                const float ctr = d[(long)gy * a.w + gx];
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + 4 * q + i;
                    float val = 0.f;
                    if (c < 49) {
                        const int uy = c / 7, ux = c - uy * 7;
                        const int yy = gy + uy - 3, xx = gx + ux - 3;
                        const float nb = (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) ? d[(long)yy * a.w + xx] : 0.f;
                        val = 100.0f * (nb - ctr);
                    }
                    t[i] = val;
                }
                v = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
        *reinterpret_cast<float4*>(&lds[row * CV_LS + 4 * q]) = v;
    }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void conv3x3_kernel(const ConvArgs a) {
    static_assert(WAVES_M * WAVES_N == 4 && WAVES_M * WM == CV_TH, "tile config");
    __shared__ __attribute__((aligned(16))) float lds[CV_ROWS * CV_LS];
    constexpr int NB = WAVES_N * WN * 16;                 // output channels per block
    const int tile = blockIdx.x;
    const int ty0 = (tile / a.tiles_x) * CV_TH, tx0 = (tile % a.tiles_x) * CV_TW;
    const int nb0 = blockIdx.y * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 15, kq = lane >> 4;
    const int NT = a.cout / 16;                           // n-tiles in the packed weights
    const int nt0 = nb0 / 16 + wn * WN;                   // this wave's first n-tile

    floatx4 acc[WM][WN];
    // ---- accumulator init: per-pixel `init` tensor, bias vector, or zero
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int gy = ty0 + wm * WM + m;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int co = (nt0 + n) * 16 + li;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (a.init) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gx = tx0 + kq * 4 + r;
                    if (gy < a.h && gx < a.w) v[r] = a.init[((long)gy * a.w + gx) * a.cout + co];
                }
            } else if (a.bias) {
                const float b = a.bias[co];
                v = (floatx4){b, b, b, b};
            }
            acc[m][n] = v;
        }
    }

    int kc16 = 0;                                         // running 16-channel chunk index into the packed weights
    for (int s = 0; s < a.nsrc; ++s) {
        for (int c0 = 0; c0 < a.chpad[s]; c0 += CV_KC) {
            __syncthreads();                              // previous chunk's readers are done
            cv_stage(lds, a, s, c0, ty0, tx0);
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < CV_KC / 16; ++sub) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap % 3;
                    float4 bf[WN], af[WM];
                    const float* wp = a.wpk + (((long)(kc16 + sub) * 9 + tap) * NT + nt0) * 256 + lane * 4;
#pragma unroll
                    for (int n = 0; n < WN; ++n) bf[n] = cer_ld4(wp + n * 256);
#pragma unroll
                    for (int m = 0; m < WM; ++m) {
                        const int row = (wm * WM + m + dy) * CV_HW + li + dx;
                        af[m] = *reinterpret_cast<const float4*>(&lds[row * CV_LS + sub * 16 + kq * 4]);
                    }
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, bf[n].x, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, bf[n].y, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, bf[n].z, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, bf[n].w, acc[m][n], 0, 0, 0);
                        }
                }
            }
            kc16 += CV_KC / 16;
        }
    }

    // ---- epilogue: lane holds pixels x = tx0 + kq*4 + r (r = 0..3) of row gy, channel co
    const int half = a.cout / 2;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int gy = ty0 + wm * WM + m;
        if (gy >= a.h) continue;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int co = (nt0 + n) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gx = tx0 + kq * 4 + r;
                if (gx >= a.w) continue;
                const long pix = (long)gy * a.w + gx;
                const float v = acc[m][n][r];
                if (EPI == CER_EPI_LINEAR) {
                    a.out[pix * a.cout + co] = v;
                } else if (EPI == CER_EPI_RELU) {
                    a.out[pix * a.cout + co] = fmaxf(v, 0.f);
                } else if (EPI == CER_EPI_GATES) {
                    const float g = cv_sigmoid(v);
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

static int padded_channels(int ch, int kind) { return kind == 1 ? 64 : ((ch + CV_KC - 1) / CV_KC) * CV_KC; }

extern "C" long cer_conv3x3_packed_size(int Cout, int Kpad) {
    if (Cout <= 0 || Kpad <= 0 || Cout % 16 || Kpad % 16) return CER_ESHAPE;
    return (long)(Kpad / 16) * 9 * (Cout / 16) * 256;
}

extern "C" int cer_conv3x3_pack_f32(const float* w, float* packed, int Cout, int Cin, const int* ch, const int* kind, int nsrc) {
    if (!w || !packed || !ch || !kind || nsrc <= 0 || nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    if (Cout % 16) return CER_ESHAPE;
    int real = 0, kpad = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (kind[s] == 1 && ch[s] != 49) return CER_ESHAPE;
        real += ch[s];
        kpad += padded_channels(ch[s], kind[s]);
    }
    if (real != Cin) return CER_ESHAPE;
    // padded K index -> real input channel (or -1)
    int* map = new int[kpad];
    int k = 0, c = 0;
    for (int s = 0; s < nsrc; ++s) {
        const int pc = padded_channels(ch[s], kind[s]);
        for (int i = 0; i < pc; ++i) map[k++] = (i < ch[s]) ? c + i : -1;
        c += ch[s];
    }
    const int NT = Cout / 16;
    for (int kc = 0; kc < kpad / 16; ++kc)
        for (int tap = 0; tap < 9; ++tap)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int co = nt * 16 + (lane & 15);
                        const int ci = map[kc * 16 + (lane >> 4) * 4 + s4];
                        const float v = ci < 0 ? 0.f : w[((long)co * Cin + ci) * 9 + tap];
                        packed[((((long)kc * 9 + tap) * NT + nt) * 64 + lane) * 4 + s4] = v;
                    }
    delete[] map;
    return CER_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN>
static int launch_conv(const ConvArgs& a, int epi, int nby, hipStream_t st) {
    const int tiles_y = (a.h + CV_TH - 1) / CV_TH;
    dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)nby);
    switch (epi) {
        case CER_EPI_LINEAR: hipLaunchKernelGGL((conv3x3_kernel<WAVES_M, WAVES_N, WM, WN, CER_EPI_LINEAR>), grid, dim3(256), 0, st, a); break;
        case CER_EPI_RELU: hipLaunchKernelGGL((conv3x3_kernel<WAVES_M, WAVES_N, WM, WN, CER_EPI_RELU>), grid, dim3(256), 0, st, a); break;
        case CER_EPI_GATES: hipLaunchKernelGGL((conv3x3_kernel<WAVES_M, WAVES_N, WM, WN, CER_EPI_GATES>), grid, dim3(256), 0, st, a); break;
        case CER_EPI_GRU: hipLaunchKernelGGL((conv3x3_kernel<WAVES_M, WAVES_N, WM, WN, CER_EPI_GRU>), grid, dim3(256), 0, st, a); break;
        default: return CER_EINVAL;
    }
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_conv3x3_f32(const cer_conv_inputs* in, const float* packed_w, const float* bias, const float* init, float* out,
                               float* out2, const float* aux, const float* aux2, int h, int w, int Cout, int epi, void* stream) {
    if (!in || !packed_w || !out || h <= 0 || w <= 0 || Cout <= 0) return CER_EINVAL;
    if (in->nsrc <= 0 || in->nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    if (epi == CER_EPI_GATES && (!out2 || !aux)) return CER_EINVAL;
    if (epi == CER_EPI_GRU && (!aux || !aux2)) return CER_EINVAL;
    if (Cout % 64 != 0) return CER_ESHAPE;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nsrc = in->nsrc;
    for (int s = 0; s < in->nsrc; ++s) {
        if (!in->src[s]) return CER_EINVAL;
        if (in->kind[s] == 0 && (in->ch[s] % CV_KC != 0)) return CER_ESHAPE;
        if (in->kind[s] == 1 && in->ch[s] != 49) return CER_ESHAPE;
        if (in->kind[s] == 0 && !cer_aligned16(in->src[s])) return CER_EALIGN;
        a.src[s] = in->src[s];
        a.ch[s] = in->ch[s];
        a.kind[s] = in->kind[s];
        a.chpad[s] = padded_channels(in->ch[s], in->kind[s]);
    }
    if (!cer_aligned16(packed_w)) return CER_EALIGN;
    a.wpk = packed_w;
    a.bias = bias;
    a.init = init;
    a.out = out;
    a.out2 = out2;
    a.aux = aux;
    a.aux2 = aux2;
    a.h = h;
    a.w = w;
    a.cout = Cout;
    a.tiles_x = (w + CV_TW - 1) / CV_TW;
    hipStream_t st = (hipStream_t)stream;
    if (Cout % 128 == 0) return launch_conv<2, 2, 4, 4>(a, epi, Cout / 128, st);
    return launch_conv<4, 1, 2, 4>(a, epi, Cout / 64, st);
}

// ---- delta tail: 3x3 conv C -> 1 on the ReLU'd hidden map + disparity update ----------------------
// One wave walks a 16-pixel row strip; lane owns channels 4*lane.. (+256*q); three rolling
// accumulators turn the 9 taps into 3 loads per pixel.
__global__ __launch_bounds__(256) void delta_tail_kernel(const float* __restrict__ hid, const float* __restrict__ wgt, float bias,
                                                         const float* __restrict__ disp_in, float* __restrict__ disp_out,
                                                         float* __restrict__ delta, int h, int w, int C, int strips_x) {
    const int lane = threadIdx.x & 63;
    const long strip = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int y = (int)(strip / strips_x);
    const int x0 = (int)(strip % strips_x) * 16;
    if (y >= h) return;
    const int NQ = C / 256;
    float accA = 0.f, accB = 0.f;                          // partial sums for x = xx+1 and x = xx
    for (int xx = x0 - 1; xx <= x0 + 16 && xx <= w; ++xx) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;                // contributions of column xx through kx = 0, 1, 2
        if (xx >= 0 && xx < w) {
            for (int q = 0; q < NQ; ++q) {
                const int ch = q * 256 + lane * 4;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    if (yy < 0 || yy >= h) continue;
                    const float4 hv = cer_ld4(hid + ((long)yy * w + xx) * C + ch);
                    c0 = cer_dot4(hv, cer_ld4(wgt + (ky * 3 + 0) * C + ch), c0);
                    c1 = cer_dot4(hv, cer_ld4(wgt + (ky * 3 + 1) * C + ch), c1);
                    c2 = cer_dot4(hv, cer_ld4(wgt + (ky * 3 + 2) * C + ch), c2);
                }
            }
        }
        // column xx feeds x = xx+1 (kx=0), x = xx (kx=1), x = xx-1 (kx=2)
        float done = accB + c2;                             // x = xx-1 is complete
        accB = accA + c1;
        accA = c0;
        const int xo = xx - 1;
        if (xo >= x0 && xo < x0 + 16 && xo < w) {
            done = cer_row16_sum(done);
            done += __shfl_xor(done, 16);
            done += __shfl_xor(done, 32);
            if (lane == 0) {
                const long p = (long)y * w + xo;
                const float dl = 0.01f * (done + bias);
                if (delta) delta[p] = dl;
                disp_out[p] = disp_in[p] + dl;
            }
        }
    }
}

extern "C" int cer_delta_tail_f32(const float* hid, const float* w, float bias, const float* disp_in, float* disp_out, float* delta, int h,
                                  int w_, int C, void* stream) {
    if (!hid || !w || !disp_in || !disp_out || h <= 0 || w_ <= 0 || C <= 0) return CER_EINVAL;
    if (C % 256 != 0) return CER_ESHAPE;
    if (!cer_aligned16(hid) || !cer_aligned16(w)) return CER_EALIGN;
    const int strips_x = (w_ + 15) / 16;
    const long strips = (long)strips_x * h;
    hipLaunchKernelGGL(delta_tail_kernel, dim3((unsigned)((strips + 3) / 4)), dim3(256), 0, (hipStream_t)stream, hid, w, bias, disp_in, disp_out,
                       delta, h, w_, C, strips_x);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
