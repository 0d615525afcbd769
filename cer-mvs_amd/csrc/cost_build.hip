// K1: epipolar cost-volume build, one cascade stage, fused
// (reference: core/corr.py:56-91 + utils/projective_ops.py:5-28 + core/corr.py:28-43 +
//  alt_cuda_corr/correlation_kernel.cu:18-119), and the correlation pyramid (core/corr.py:94-97).
//
// v1 mapping: one 16-lane DPP row per reference pixel; the row walks hypotheses k (and, in the
// view-sum modes, views v inside k).  Lane `sub` owns channel quads {4*sub + 64*q}, so every
// wave load instruction fetches four whole 256-B texels; coordinates are computed in registers
// from Pij (no coordinate tensors, unlike projective_ops.py:13-28 which materialises 3 x 121 MB
// per view at 1600x1184).  The lane-partial dots of all local views are summed BEFORE the single
// cross-lane DPP reduction; results for 16 consecutive hypotheses are parked one per lane and
// stored as one 64-B segment.
#include "common.hpp"

template <int NQ, bool SUM>
__global__ __launch_bounds__(256) void cost_build_kernel(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                                                         const float* __restrict__ Pij, const float* __restrict__ disp_in,
                                                         float* __restrict__ vol, float* __restrict__ origin_out, int V, int h1, int w1,
                                                         int h2, int w2, int C, int D, int rs, float incre, float lim, int shift,
                                                         int accumulate) {
    const int sub = threadIdx.x & 15;
    const long P = (long)h1 * w1;
    const long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= P) return;
    const float px = (float)(p % w1), py = (float)(p / w1);
    float origin = disp_in[p];
    if (shift && origin < lim) origin = lim;
    if (origin_out && sub == 0 && blockIdx.y == 0) origin_out[p] = origin;
    float4 f1q[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) f1q[q] = cer_ld4(fmap1 + p * C + 4 * sub + 64 * q);
    const int half = D / 2;
    const long P2C = (long)h2 * w2 * C;
    // SUM: blockIdx.y == 0 handles all views; else blockIdx.y is the view
    const int v_lo = SUM ? 0 : (int)blockIdx.y, v_hi = SUM ? V : (int)blockIdx.y + 1;
    float* orow = SUM ? vol + p * rs : vol + ((long)blockIdx.y * P + p) * rs;
    for (int k0 = 0; k0 < D; k0 += 16) {
        float mine = 0.f;
        const int kend = min(16, D - k0);
        for (int j = 0; j < kend; ++j) {
            // two roundings, as torch computes (arange(D) - D//2) * incre + origin (core/corr.py:56,65): no fma contraction
            const float hyp = __fadd_rn(__fmul_rn((float)(k0 + j - half), incre), origin);
            float tot = 0.f;
            for (int v = v_lo; v < v_hi; ++v) {
                const float* m = Pij + v * 16;
                const float X = fmaf(m[3], hyp, fmaf(m[1], py, m[0] * px) + m[2]);
                const float Y = fmaf(m[7], hyp, fmaf(m[5], py, m[4] * px) + m[6]);
                const float Z = fmaf(m[11], hyp, fmaf(m[9], py, m[8] * px) + m[10]);
                float u = X / Z, w = Y / Z;
                const bool ok = (u == u) && (w == w);        // NaN (0/0) samples nothing
                u = fminf(fmaxf(u, -1e4f), 1e4f);             // core/corr.py:88
                w = fminf(fmaxf(w, -1e4f), 1e4f);
                if (ok) {
                    const float fu = floorf(u), fw = floorf(w);
                    tot += cer_bilerp_dot<NQ>(fmap2 + (long)v * P2C, h2, w2, C, sub, fu, fw, u - fu, w - fw, f1q);
                }
            }
            tot = cer_row16_sum(tot);
            if (sub == j) mine = tot;
        }
        if (sub < kend) {
            float* o = orow + k0 + sub;
            *o = accumulate ? (*o + mine) : mine;
        }
    }
}

template <int NQ>
static int launch_build(const float* f1, const float* f2, const float* Pij, const float* disp_in, float* vol, float* origin_out, int V,
                        int h1, int w1, int h2, int w2, int C, int D, int rs, double incre_d, int shift, int mode, hipStream_t st) {
    const long P = (long)h1 * w1;
    const float lim = (float)((D / 2) * incre_d);
    const float incre = (float)incre_d;
    const unsigned gx = (unsigned)((P + 15) / 16);
    if (mode == 0)
        hipLaunchKernelGGL((cost_build_kernel<NQ, false>), dim3(gx, (unsigned)V), dim3(256), 0, st, f1, f2, Pij, disp_in, vol, origin_out, V,
                           h1, w1, h2, w2, C, D, rs, incre, lim, shift, 0);
    else
        hipLaunchKernelGGL((cost_build_kernel<NQ, true>), dim3(gx, 1), dim3(256), 0, st, f1, f2, Pij, disp_in, vol, origin_out, V, h1, w1, h2,
                           w2, C, D, rs, incre, lim, shift, mode == 2 ? 1 : 0);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_cost_build_f32(const float* fmap1, const float* fmap2, const float* Pij, const float* disp_in, float* vol,
                                  float* origin_out, int V, int h1, int w1, int h2, int w2, int C, int D, int row_stride, double incre,
                                  int shift, int mode, void* stream) {
    if (!fmap1 || !fmap2 || !Pij || !disp_in || !vol) return CER_EINVAL;
    if (V <= 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || C <= 0 || D <= 0 || row_stride < D || mode < 0 || mode > 2) return CER_EINVAL;
    if (C % 64 != 0 || C > 256 || V > 65535) return CER_ESHAPE;
    if (!cer_aligned16(fmap1) || !cer_aligned16(fmap2)) return CER_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    switch (C / 64) {
        case 1: return launch_build<1>(fmap1, fmap2, Pij, disp_in, vol, origin_out, V, h1, w1, h2, w2, C, D, row_stride, incre, shift, mode, st);
        case 2: return launch_build<2>(fmap1, fmap2, Pij, disp_in, vol, origin_out, V, h1, w1, h2, w2, C, D, row_stride, incre, shift, mode, st);
        case 3: return launch_build<3>(fmap1, fmap2, Pij, disp_in, vol, origin_out, V, h1, w1, h2, w2, C, D, row_stride, incre, shift, mode, st);
        default: return launch_build<4>(fmap1, fmap2, Pij, disp_in, vol, origin_out, V, h1, w1, h2, w2, C, D, row_stride, incre, shift, mode, st);
    }
}

// ---- pyramid: one thread per row; rows are <= a few hundred bytes and stay in L1/L2 -------------
__global__ __launch_bounds__(256) void pyramid_kernel(float* __restrict__ vol, long rows, int D, int rs, int L, float scale) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float* row = vol + r * rs;
    if (scale != 1.0f)
        for (int k = 0; k < D; ++k) row[k] *= scale;
    int off = 0, n = D;
    for (int l = 1; l < L; ++l) {
        const int m = n / 2;
        float* src = row + off;
        float* dst = row + off + n;
        // F.avg_pool2d([1,2]): (a + b) / 2, computed as sum * 0.5 (exact either way)
        for (int k = 0; k < m; ++k) dst[k] = (src[2 * k] + src[2 * k + 1]) * 0.5f;
        off += n;
        n = m;
    }
}

extern "C" int cer_pyramid_f32(float* vol, long rows, int D, int row_stride, int num_levels, float scale, void* stream) {
    if (!vol || rows <= 0 || D <= 0 || num_levels <= 0) return CER_EINVAL;
    int need = 0, n = D;
    for (int l = 0; l < num_levels; ++l) { need += n; n /= 2; }
    if (row_stride < need) return CER_ESHAPE;
    hipLaunchKernelGGL(pyramid_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vol, rows, D, row_stride,
                       num_levels, scale);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// ---- layout helpers -----------------------------------------------------------------------------
// NCHW [C,P] -> NHWC [P,C] * scale through a 64x64 LDS tile (both sides coalesced).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long P, float scale) {
    __shared__ float tile[64][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i;
        const long p = p0 + tx;
        tile[i][tx] = (c < C && p < P) ? src[(long)c * P + p] * scale : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const long p = p0 + i;
        const int c = c0 + tx;
        if (p < P && c < C) dst[p * C + c] = tile[tx][i];
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long P, float scale) {
    __shared__ float tile[64][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const long p = p0 + i;
        const int c = c0 + tx;
        tile[i][tx] = (c < C && p < P) ? src[p * C + c] * scale : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i;
        const long p = p0 + tx;
        if (p < P && c < C) dst[(long)c * P + p] = tile[tx][i];
    }
}
extern "C" int cer_nchw_to_nhwc_f32(const float* src, float* dst, int C, long P, float scale, void* stream) {
    if (!src || !dst || C <= 0 || P <= 0) return CER_EINVAL;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((P + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, C, P, scale);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
extern "C" int cer_nhwc_to_nchw_f32(const float* src, float* dst, int C, long P, float scale, void* stream) {
    if (!src || !dst || C <= 0 || P <= 0) return CER_EINVAL;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((P + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, C, P, scale);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
