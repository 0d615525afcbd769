// K5: geometric-consistency vote of one reference depth map against its source views, fused into one pass
// (reference: fusion.py:39-83 reproject_with_depth, :86-106 check_geometric_consistency, :226-236 vote + averaged depth,
//  utils/bilinear_sampler.py:32-41 -> F.grid_sample(align_corners=True, zeros)).
//
// The reference materialises, per source view, ~40 [S,H,W] temporaries (meshgrids, homogeneous stacks, 3 batched matmuls,
// a grid_sample, 9 masks).  Here one thread owns one reference pixel and walks the source views: every per-pixel quantity
// lives in registers, the only gather is the 2x2 footprint of the source depth map, and what leaves the kernel is the
// vote mask (1 B), the averaged depth (4 B) and one mask-area counter - ~(4 + 5) B of compulsory traffic per pixel plus the
// (cached) depth gathers, against ~170 S B per pixel of temporaries in the op-by-op form.  The per-view tensors of the
// reference's API are written only on request (literal mode, for parity tests and for callers of the per-view API).
#include "common.hpp"

#define GEO_CAM_FLOATS 60     // K_ref^-1 [9] | E_src E_ref^-1 rows 0-2 [12] | K_src [9] | K_src^-1 [9] | E_ref E_src^-1 rows 0-2 [12] | K_ref [9]

__device__ __forceinline__ void geo_mat3(const float* __restrict__ m, float a, float b, float c, float& x, float& y, float& z) {
    x = fmaf(m[2], c, fmaf(m[1], b, m[0] * a));
    y = fmaf(m[5], c, fmaf(m[4], b, m[3] * a));
    z = fmaf(m[8], c, fmaf(m[7], b, m[6] * a));
}
__device__ __forceinline__ void geo_mat34(const float* __restrict__ m, float a, float b, float c, float& x, float& y, float& z) {
    x = fmaf(m[2], c, fmaf(m[1], b, m[0] * a)) + m[3];
    y = fmaf(m[6], c, fmaf(m[5], b, m[4] * a)) + m[7];
    z = fmaf(m[10], c, fmaf(m[9], b, m[8] * a)) + m[11];
}

// F.grid_sample(bilinear, zeros, align_corners=True) at pixel coordinates (px, py), through the reference's
// normalise / unnormalise round trip (utils/bilinear_sampler.py:35-36, ATen grid_sampler_unnormalize)
__device__ __forceinline__ float geo_sample(const float* __restrict__ img, int h, int w, float px, float py) {
    const float gx = 2.0f * px / (float)(w - 1) - 1.0f, gy = 2.0f * py / (float)(h - 1) - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    if (!(fx >= -1.0f && fx <= (float)w && fy >= -1.0f && fy <= (float)h)) return 0.0f;     // also catches NaN / inf
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = (float)x1 - ix, wy1 = iy - fy, wy0 = (float)y1 - iy;
    const bool ax0 = x0 >= 0 && x0 < w, ax1 = x1 >= 0 && x1 < w, ay0 = y0 >= 0 && y0 < h, ay1 = y1 >= 0 && y1 < h;
    float out = 0.0f;
    if (ay0 && ax0) out += img[(long)y0 * w + x0] * (wx0 * wy0);
    if (ay0 && ax1) out += img[(long)y0 * w + x1] * (wx1 * wy0);
    if (ay1 && ax0) out += img[(long)y1 * w + x0] * (wx0 * wy1);
    if (ay1 && ax1) out += img[(long)y1 * w + x1] * (wx1 * wy1);
    return out;
}

struct GeoThresholds {
    float dist[9], rel[9];                                 // (float)(i / thre1), (float)(i / thre2) for i = 2..10, rounded once on the host
};

__global__ __launch_bounds__(256) void geo_consistency_kernel(const float* __restrict__ depth_ref, const float* __restrict__ depth_src,
                                                              const float* __restrict__ cams, int S, int h, int w, const GeoThresholds th,
                                                              unsigned char* __restrict__ geo_mask, float* __restrict__ depth_est,
                                                              unsigned int* __restrict__ mask_count, unsigned char* __restrict__ masks9,
                                                              float* __restrict__ drep_out, float* __restrict__ xs_out,
                                                              float* __restrict__ ys_out, float* __restrict__ rel_out) {
    const long P = (long)h * w;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    bool geo = false;
    if (p < P) {
        const int yi = (int)(p / w), xi = (int)(p - (long)yi * w);
        const float x = (float)xi, y = (float)yi, d = depth_ref[p];
        int cnt[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) cnt[i] = 0;
        float dsum = 0.0f;
        for (int s = 0; s < S; ++s) {
            const float* c = cams + s * GEO_CAM_FLOATS;                 // wave-uniform: scalar loads
            float X, Y, Z, U, V, Wc;
            geo_mat3(c, x * d, y * d, d, X, Y, Z);                      // K_ref^-1 [x y 1]^T d                     (fusion.py:50-52)
            geo_mat34(c + 9, X, Y, Z, U, V, Wc);                        // into the source camera                  (:55-56)
            geo_mat3(c + 21, U, V, Wc, X, Y, Z);                        // K_src                                   (:58)
            const float xs = X / Z, ys = Y / Z;                         //                                         (:59)
            const float ds = geo_sample(depth_src + (long)s * P, h, w, xs, ys);                                 // (:67)
            geo_mat3(c + 30, xs * ds, ys * ds, ds, X, Y, Z);            // K_src^-1 [xs ys 1]^T ds                  (:71-72)
            geo_mat34(c + 39, X, Y, Z, U, V, Wc);                       // back into the reference camera          (:74-75)
            const float drep = Wc;                                      //                                         (:77)
            geo_mat3(c + 51, U, V, Wc, X, Y, Z);                        // K_ref                                   (:78)
            const float xr = X / Z, yr = Y / Z;                         //                                         (:79)
            const float ex = xr - x, ey = yr - y;
            const float dist = sqrtf(ex * ex + ey * ey);                //                                         (:94)
            const float rel = fabsf(drep - d) / d;                      //                                         (:97-98)
            bool m10 = false;
#pragma unroll
            for (int i = 2; i <= 10; ++i) {
                const bool m = dist < th.dist[i - 2] && rel < th.rel[i - 2];                                    // (:100-103)                                // (:100-103)
                cnt[i - 2] += m ? 1 : 0;
                if (masks9) masks9[((long)(i - 2) * S + s) * P + p] = m ? 1 : 0;
                if (i == 10) m10 = m;
            }
            dsum += m10 ? drep : 0.0f;                                  //                                         (:104, :236)
            if (drep_out) drep_out[(long)s * P + p] = m10 ? drep : 0.0f;
            if (xs_out) xs_out[(long)s * P + p] = xs;
            if (ys_out) ys_out[(long)s * P + p] = ys;
            if (rel_out) rel_out[(long)s * P + p] = rel;
        }
        const int n = 1 + S;
        geo = cnt[8] >= n;                                              //                                         (:232)
#pragma unroll
        for (int i = 2; i <= 10; ++i)
            if (i < n) geo = geo || cnt[i - 2] >= i;                    //                                         (:234-235)
        if (geo_mask) geo_mask[p] = geo ? 1 : 0;
        if (depth_est) depth_est[p] = (dsum + d) / (float)(cnt[8] + 1); //                                         (:236)
    }
    if (mask_count) {                 // mask area of this view: wave ballots -> one atomic per block, spread over 64 counters
        __shared__ unsigned int wave_cnt[4];
        const unsigned long long b = __ballot(geo);
        if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (unsigned int)__popcll(b);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int c = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            if (c) atomicAdd(mask_count + (blockIdx.x & (CER_GEO_COUNTERS - 1)), c);
        }
    }
}

extern "C" int cer_geo_consistency_f32(const float* depth_ref, const float* depth_src, const float* cams, int S, int h, int w, double thre1,
                                       double thre2, unsigned char* geo_mask, float* depth_est, unsigned int* mask_count,
                                       unsigned char* masks9, float* depth_reprojected, float* x_src, float* y_src, float* rel_diff,
                                       void* stream) {
    if (!depth_ref || !depth_src || !cams || S <= 0 || h <= 1 || w <= 1) return CER_EINVAL;
    if (S > 10) return CER_ESHAPE;                                      // the reference indexes masks[i-2] for i < 1+S <= 11 (fusion.py:226-228)
    if (!(thre1 > 0.0) || !(thre2 > 0.0)) return CER_EINVAL;
    const long P = (long)h * w;
    GeoThresholds th;                                      // python: dist < i / thre1 compares an fp32 tensor with the double i / thre1 cast to fp32
    for (int i = 2; i <= 10; ++i) {
        th.dist[i - 2] = (float)((double)i / thre1);
        th.rel[i - 2] = (float)((double)i / thre2);
    }
    hipLaunchKernelGGL(geo_consistency_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depth_ref, depth_src,
                       cams, S, h, w, th, geo_mask, depth_est, mask_count, masks9, depth_reprojected, x_src, y_src, rel_diff);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// ---- multi-resolution merge (reference: multires.py:16-40): the scale-1 depth map resized to the scale-2 map's size, then
//   im = where(|im1 - im2| < th * im1, im2, im1)
// and an optional down-sampling of the result.  Both resizes are cv2.resize(..., INTER_LINEAR) on float32 images, restated here:
// source coordinate fx = float((dx + 0.5) * (src / dst) - 0.5) (the product in double), sx = floor(fx), fx -= sx; sx < 0 -> (0, 0);
// sx >= src - 1 -> (src - 1, 0) with both taps on the last pixel; horizontal pass s[sx] * (1 - fx) + s[sx + 1] * fx on the two
// source rows, then the vertical pass with the row weights - float products and sums in that order (no fma: -ffp-contract=off).
__device__ __forceinline__ void mr_coord(int d, double scale, int ssize, int& s0, int& s1, float& f) {
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= ssize - 1) { sx = ssize - 1; fx = 0.f; s0 = sx; s1 = sx; f = fx; return; }
    s0 = sx; s1 = sx + 1; f = fx;
}
__device__ __forceinline__ float mr_sample(const float* __restrict__ src, int h, int w, int y, int x, double sy, double sx) {
    int x0, x1, y0, y1;
    float fx, fy;
    mr_coord(x, sx, w, x0, x1, fx);
    mr_coord(y, sy, h, y0, y1, fy);
    const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
    const float r0 = src[(long)y0 * w + x0] * a0 + src[(long)y0 * w + x1] * a1;
    const float r1 = src[(long)y1 * w + x0] * a0 + src[(long)y1 * w + x1] * a1;
    return r0 * b0 + r1 * b1;
}
__global__ __launch_bounds__(256) void multires_merge_kernel(const float* __restrict__ im1, int h1, int w1, const float* __restrict__ im2, int h2,
                                                             int w2, float th, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)h2 * w2) return;
    const int y = (int)(i / w2), x = (int)(i - (long)y * w2);
    const float a = (h1 == h2 && w1 == w2) ? im1[i] : mr_sample(im1, h1, w1, y, x, (double)h1 / h2, (double)w1 / w2);
    const float b = im2[i];
    out[i] = (fabsf(a - b) < th * a) ? b : a;              // (NaN compares false: keeps the scale-1 value like np.where)
}
__global__ __launch_bounds__(256) void resize_linear_kernel(const float* __restrict__ src, int h, int w, float* __restrict__ dst, int ho, int wo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)ho * wo) return;
    const int y = (int)(i / wo), x = (int)(i - (long)y * wo);
    dst[i] = mr_sample(src, h, w, y, x, (double)h / ho, (double)w / wo);
}

extern "C" int cer_multires_merge_f32(const float* im1, int h1, int w1, const float* im2, int h2, int w2, double th, float* out, void* stream) {
    if (!im1 || !im2 || !out || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0) return CER_EINVAL;
    const long P = (long)h2 * w2;
    hipLaunchKernelGGL(multires_merge_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, im1, h1, w1, im2, h2, w2,
                       (float)th, out);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_resize_linear_f32(const float* src, int h, int w, float* dst, int ho, int wo, void* stream) {
    if (!src || !dst || h <= 0 || w <= 0 || ho <= 0 || wo <= 0) return CER_EINVAL;
    const long P = (long)ho * wo;
    hipLaunchKernelGGL(resize_linear_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, h, w, dst, ho, wo);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
