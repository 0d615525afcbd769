// alt_cuda_corr forward/backward with the reference's literal semantics
// (reference: alt_cuda_corr/correlation_kernel.cu:18-119,122-256; correlation.cpp:23-48).
// Design: 16 lanes per (b, pixel) - lane `sub` owns channel quads - so a wavefront moves four
// 256-B texels per load instruction and reduces with DPP row adds; no LDS, no barriers, no
// reliance on lock-step execution (the reference's 32-thread blocks lean on warp lock-step).
#include "common.hpp"

template <int NQ>
__global__ __launch_bounds__(256) void alt_corr_fwd_kernel(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                                                           const float* __restrict__ coords, float* __restrict__ corr,
                                                           int N, int H1, int W1, int H2, int W2, int C, int r) {
    const int b = blockIdx.y;
    const int sub = threadIdx.x & 15;
    const long P1 = (long)H1 * W1;
    const long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= P1) return;                                  // whole 16-lane rows exit together
    const float* f1 = fmap1 + ((long)b * P1 + p) * C + 4 * sub;
    float4 f1q[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) f1q[q] = cer_ld4(f1 + 64 * q);
    const float* f2 = fmap2 + (long)b * H2 * W2 * C;
    const int rd = 2 * r + 1;
    for (int n = 0; n < N; ++n) {
        const float* cp = coords + (((long)b * N + n) * P1 + p) * 2;
        const float x = cp[0], y = cp[1];
        const bool finite = (fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f);
        const float fx = floorf(x), fy = floorf(y);
        const float dx = x - fx, dy = y - fy;
        float* out = corr + (((long)b * N + n) * rd * rd) * P1 + p;
        for (int kx = 0; kx < rd; ++kx)
            for (int ky = 0; ky < rd; ++ky) {
                float s = finite ? cer_bilerp_dot<NQ>(f2, H2, W2, C, sub, fx + (float)(kx - r), fy + (float)(ky - r), dx, dy, f1q) : 0.f;
                s = cer_row16_sum(s);
                if (sub == 0) out[(long)(ky + rd * kx) * P1] = s;
            }
    }
}

// backward, any radius: one 16-lane row per (b, pixel); fmap2_grad through float atomics (G2 = true: the literal one-call form of
// correlation_kernel.cu:122-256) or not at all (G2 = false: the deterministic path adds it by a sorted segmented reduction, below).
template <int NQ, bool G2 = true>
__global__ __launch_bounds__(256) void alt_corr_bwd_kernel(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                                                           const float* __restrict__ coords, const float* __restrict__ corr_grad,
                                                           float* __restrict__ fmap1_grad, float* __restrict__ fmap2_grad,
                                                           int N, int H1, int W1, int H2, int W2, int C, int r) {
    const int b = blockIdx.y;
    const int sub = threadIdx.x & 15;
    const long P1 = (long)H1 * W1;
    const long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= P1) return;
    const float* f1 = fmap1 + ((long)b * P1 + p) * C + 4 * sub;
    float4 f1q[NQ], g1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        f1q[q] = cer_ld4(f1 + 64 * q);
        g1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* f2 = fmap2 + (long)b * H2 * W2 * C;
    float* g2 = fmap2_grad + (long)b * H2 * W2 * C;
    const int rd = 2 * r + 1;
    for (int n = 0; n < N; ++n) {
        const float* cp = coords + (((long)b * N + n) * P1 + p) * 2;
        const float x = cp[0], y = cp[1];
        if (!((fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f))) continue;
        const float fx = floorf(x), fy = floorf(y);
        const float dx = x - fx, dy = y - fy;
        if (fx < -2.0e9f || fx > 2.0e9f || fy < -2.0e9f || fy > 2.0e9f) continue;
        const int ix0 = (int)fx - r, iy0 = (int)fy - r;
        const float* gp = corr_grad + (((long)b * N + n) * rd * rd) * P1 + p;
        // texel (iy, ix), iy, ix in [0, rd]: gathers the four outputs whose footprint contains it
        for (int iy = 0; iy <= rd; ++iy)
            for (int ix = 0; ix <= rd; ++ix) {
                const int h2 = iy0 + iy, w2 = ix0 + ix;
                if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;
                float g = 0.f;
                if (iy > 0 && ix > 0) g += gp[(long)((iy - 1) + rd * (ix - 1)) * P1] * dy * dx;
                if (iy > 0 && ix < rd) g += gp[(long)((iy - 1) + rd * ix) * P1] * dy * (1.f - dx);
                if (iy < rd && ix > 0) g += gp[(long)(iy + rd * (ix - 1)) * P1] * (1.f - dy) * dx;
                if (iy < rd && ix < rd) g += gp[(long)(iy + rd * ix) * P1] * (1.f - dy) * (1.f - dx);
                const long toff = ((long)h2 * W2 + w2) * C + 4 * sub;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float4 t = cer_ld4(f2 + toff + 64 * q);
                    g1[q].x = fmaf(g, t.x, g1[q].x);
                    g1[q].y = fmaf(g, t.y, g1[q].y);
                    g1[q].z = fmaf(g, t.z, g1[q].z);
                    g1[q].w = fmaf(g, t.w, g1[q].w);
                    if (G2) {
                        float* gq = g2 + toff + 64 * q;
                        atomicAdd(gq + 0, g * f1q[q].x);
                        atomicAdd(gq + 1, g * f1q[q].y);
                        atomicAdd(gq + 2, g * f1q[q].z);
                        atomicAdd(gq + 3, g * f1q[q].w);
                    }
                }
            }
    }
    float* o1 = fmap1_grad + ((long)b * P1 + p) * C + 4 * sub;
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<float4*>(o1 + 64 * q) = g1[q];
}

template <int NQ>
static int launch_fwd(const float* f1, const float* f2, const float* co, float* corr, int B, int N, int H1, int W1, int H2, int W2, int C,
                      int r, hipStream_t st) {
    dim3 grid((unsigned)(((long)H1 * W1 + 15) / 16), (unsigned)B);
    hipLaunchKernelGGL(alt_corr_fwd_kernel<NQ>, grid, dim3(256), 0, st, f1, f2, co, corr, N, H1, W1, H2, W2, C, r);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_alt_corr_forward_f32(const float* fmap1, const float* fmap2, const float* coords, float* corr, int B, int N, int H1,
                                        int W1, int H2, int W2, int C, int radius, void* stream) {
    if (!fmap1 || !fmap2 || !coords || !corr) return CER_EINVAL;
    if (B <= 0 || N <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0 || radius < 0) return CER_EINVAL;
    if (C % 64 != 0 || C > 256) return CER_ESHAPE;
    if (!cer_aligned16(fmap1) || !cer_aligned16(fmap2)) return CER_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    switch (C / 64) {
        case 1: return launch_fwd<1>(fmap1, fmap2, coords, corr, B, N, H1, W1, H2, W2, C, radius, st);
        case 2: return launch_fwd<2>(fmap1, fmap2, coords, corr, B, N, H1, W1, H2, W2, C, radius, st);
        case 3: return launch_fwd<3>(fmap1, fmap2, coords, corr, B, N, H1, W1, H2, W2, C, radius, st);
        default: return launch_fwd<4>(fmap1, fmap2, coords, corr, B, N, H1, W1, H2, W2, C, radius, st);
    }
}

template <int NQ>
static int launch_bwd(const float* f1, const float* f2, const float* co, const float* cg, float* g1, float* g2, int B, int N, int H1, int W1,
                      int H2, int W2, int C, int r, hipStream_t st) {
    dim3 grid((unsigned)(((long)H1 * W1 + 15) / 16), (unsigned)B);
    if (g2) hipLaunchKernelGGL((alt_corr_bwd_kernel<NQ, true>), grid, dim3(256), 0, st, f1, f2, co, cg, g1, g2, N, H1, W1, H2, W2, C, r);
    else hipLaunchKernelGGL((alt_corr_bwd_kernel<NQ, false>), grid, dim3(256), 0, st, f1, f2, co, cg, g1, g2, N, H1, W1, H2, W2, C, r);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_alt_corr_backward_f32(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                                         float* fmap1_grad, float* fmap2_grad, float* coords_grad, int B, int N, int H1, int W1, int H2,
                                         int W2, int C, int radius, void* stream) {
    if (!fmap1 || !fmap2 || !coords || !corr_grad || !fmap1_grad) return CER_EINVAL;      // fmap2_grad NULL: fmap1 gradient only
    if (B <= 0 || N <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0 || radius < 0) return CER_EINVAL;
    if (C % 64 != 0 || C > 256) return CER_ESHAPE;
    if (!cer_aligned16(fmap1) || !cer_aligned16(fmap2) || !cer_aligned16(fmap1_grad)) return CER_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (fmap2_grad) {
        e = hipMemsetAsync(fmap2_grad, 0, sizeof(float) * (size_t)B * H2 * W2 * C, st);
        if (e != hipSuccess) return (int)e;
    }
    if (coords_grad) {
        e = hipMemsetAsync(coords_grad, 0, sizeof(float) * (size_t)B * N * H1 * W1 * 2, st);
        if (e != hipSuccess) return (int)e;
    }
    switch (C / 64) {
        case 1: return launch_bwd<1>(fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, B, N, H1, W1, H2, W2, C, radius, st);
        case 2: return launch_bwd<2>(fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, B, N, H1, W1, H2, W2, C, radius, st);
        case 3: return launch_bwd<3>(fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, B, N, H1, W1, H2, W2, C, radius, st);
        default: return launch_bwd<4>(fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, B, N, H1, W1, H2, W2, C, radius, st);
    }
}

// ---- deterministic fmap2 gradient ("atomics on fmap2_grad need a CDNA-friendly reduction") -------------
// fmap2_grad[b, texel, :] = sum over the samples (n, p) whose footprint holds the texel of  g(sample, texel) * fmap1[b, p, :].
// 1. cer_alt_corr_bwd_tuples_f32: one (key, coefficient, source pixel) per (sample, footprint texel): key = b*H2*W2 + texel, or
//    the sentinel B*H2*W2 when the texel is outside the map / the coordinate is not finite;
// 2. the caller sorts the keys (stable: ties keep the sample order) and finds the segment bounds of every texel;
// 3. cer_alt_corr_bwd_reduce_f32: one 16-lane row per texel adds its segment in sorted order - a fixed summation order, no atomics.
__global__ __launch_bounds__(256) void alt_corr_bwd_tuples_kernel(const float* __restrict__ coords, const float* __restrict__ corr_grad,
                                                                  long* __restrict__ keys, float* __restrict__ coef, int* __restrict__ src, int B,
                                                                  int N, int H1, int W1, int H2, int W2, int r) {
    const long P1 = (long)H1 * W1;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (b, n, p)
    if (i >= (long)B * N * P1) return;
    const long p = i % P1;
    const int n = (int)((i / P1) % N), b = (int)(i / (P1 * N));
    const int rd = 2 * r + 1, fp = (rd + 1) * (rd + 1);
    const long sentinel = (long)B * H2 * W2;
    const float* cp = coords + i * 2;
    const float x = cp[0], y = cp[1];
    const float fx = floorf(x), fy = floorf(y);
    const bool ok = (fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f) && !(fx < -2.0e9f || fx > 2.0e9f || fy < -2.0e9f || fy > 2.0e9f);
    const float dx = x - fx, dy = y - fy;
    const int ix0 = (int)fx - r, iy0 = (int)fy - r;
    const float* gp = corr_grad + (((long)b * N + n) * rd * rd) * P1 + p;
    for (int iy = 0; iy <= rd; ++iy)
        for (int ix = 0; ix <= rd; ++ix) {
            const int h2 = iy0 + iy, w2 = ix0 + ix;
            const long o = i * fp + iy * (rd + 1) + ix;
            float g = 0.f;
            const bool in = ok && h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2;
            if (in) {
                if (iy > 0 && ix > 0) g += gp[(long)((iy - 1) + rd * (ix - 1)) * P1] * dy * dx;
                if (iy > 0 && ix < rd) g += gp[(long)((iy - 1) + rd * ix) * P1] * dy * (1.f - dx);
                if (iy < rd && ix > 0) g += gp[(long)(iy + rd * (ix - 1)) * P1] * (1.f - dy) * dx;
                if (iy < rd && ix < rd) g += gp[(long)(iy + rd * ix) * P1] * (1.f - dy) * (1.f - dx);
            }
            keys[o] = in ? ((long)b * H2 + h2) * W2 + w2 : sentinel;
            coef[o] = g;
            src[o] = (int)((long)b * P1 + p);
        }
}

extern "C" int cer_alt_corr_bwd_tuples_f32(const float* coords, const float* corr_grad, long* keys, float* coef, int* src, int B, int N, int H1,
                                           int W1, int H2, int W2, int radius, void* stream) {
    if (!coords || !corr_grad || !keys || !coef || !src) return CER_EINVAL;
    if (B <= 0 || N <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || radius < 0) return CER_EINVAL;
    if ((long)B * H1 * W1 >= (1L << 31)) return CER_ESHAPE;
    const long n = (long)B * N * H1 * W1;
    hipLaunchKernelGGL(alt_corr_bwd_tuples_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords, corr_grad, keys, coef,
                       src, B, N, H1, W1, H2, W2, radius);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

template <int NQ>
__global__ __launch_bounds__(256) void alt_corr_bwd_reduce_kernel(const float* __restrict__ fmap1, const long* __restrict__ order,
                                                                  const float* __restrict__ coef, const int* __restrict__ src,
                                                                  const long* __restrict__ seg, float* __restrict__ fmap2_grad, long T, int C) {
    const int sub = threadIdx.x & 15;
    const long t = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (t >= T) return;
    float4 g[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) g[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = seg[t]; i < seg[t + 1]; ++i) {          // sorted order: the sum is the same bits on every run
        const long k = order[i];
        const float c = coef[k];
        const float* f1 = fmap1 + (long)src[k] * C + 4 * sub;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 v = cer_ld4(f1 + 64 * q);
            g[q].x = fmaf(c, v.x, g[q].x); g[q].y = fmaf(c, v.y, g[q].y); g[q].z = fmaf(c, v.z, g[q].z); g[q].w = fmaf(c, v.w, g[q].w);
        }
    }
    float* o = fmap2_grad + t * C + 4 * sub;
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<float4*>(o + 64 * q) = g[q];
}

extern "C" int cer_alt_corr_bwd_reduce_f32(const float* fmap1, const long* order, const float* coef, const int* src, const long* seg,
                                           float* fmap2_grad, long T, int C, void* stream) {
    if (!fmap1 || !order || !coef || !src || !seg || !fmap2_grad || T <= 0) return CER_EINVAL;
    if (C % 64 != 0 || C > 256) return CER_ESHAPE;
    if (!cer_aligned16(fmap1) || !cer_aligned16(fmap2_grad)) return CER_EALIGN;
    dim3 grid((unsigned)((T + 15) / 16));
    hipStream_t st = (hipStream_t)stream;
    switch (C / 64) {
        case 1: hipLaunchKernelGGL(alt_corr_bwd_reduce_kernel<1>, grid, dim3(256), 0, st, fmap1, order, coef, src, seg, fmap2_grad, T, C); break;
        case 2: hipLaunchKernelGGL(alt_corr_bwd_reduce_kernel<2>, grid, dim3(256), 0, st, fmap1, order, coef, src, seg, fmap2_grad, T, C); break;
        case 3: hipLaunchKernelGGL(alt_corr_bwd_reduce_kernel<3>, grid, dim3(256), 0, st, fmap1, order, coef, src, seg, fmap2_grad, T, C); break;
        default: hipLaunchKernelGGL(alt_corr_bwd_reduce_kernel<4>, grid, dim3(256), 0, st, fmap1, order, coef, src, seg, fmap2_grad, T, C); break;
    }
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
