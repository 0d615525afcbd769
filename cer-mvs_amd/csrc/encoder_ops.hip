// Fused normalisation / activation / residual passes of the feature and context encoders
// (reference: core/extractor.py:49-57 ResidualBlock.forward, :143-150 BasicEncoder.forward; InstanceNorm2d
// defaults :28-31,:75-76 = no affine, no running stats, eps 1e-5, biased variance).
//
// The encoder convolutions stay on MIOpen (BASELINE.json north_star); PyTorch then spends 15 tensor passes
// per residual block on statistics, normalise, ReLU, add, ReLU.  These two kernels do the same in 7:
//   cer_plane_stats_f32 : per (image, channel) plane mean and 1/sqrt(var + eps), accumulated in fp64
//   cer_norm_act_f32    : out = relu_out( relu_a(norm(x)) + relu_b(norm_r(res)) ), every piece optional
// NCHW planes (MIOpen's layout), HBM-bound: 16-B loads/stores, grid-stride.
#include "common.hpp"

__global__ __launch_bounds__(1024) void plane_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, long plane, float eps) {
    const float* p = x + (long)blockIdx.x * plane;
    double s = 0.0, s2 = 0.0;
    const long n4 = (plane % 4 == 0) ? plane / 4 : 0;      // planes that are not 16-B multiples take the scalar loop
    for (long i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = cer_ld4(p + 4 * i);
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (long i = n4 * 4 + threadIdx.x; i < plane; i += 1024) {
        const float v = p[i];
        s += v;
        s2 += (double)v * v;
    }
    __shared__ double sh[2][16];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o);
        s2 += __shfl_xor(s2, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sh[0][wave] = s;
        sh[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < 16; ++i) {
            a += sh[0][i];
            b += sh[1][i];
        }
        const double mean = a / (double)plane;
        double var = b / (double)plane - mean * mean;     // biased variance, as InstanceNorm2d
        if (var < 0.0) var = 0.0;
        stats[2 * blockIdx.x] = (float)mean;
        stats[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

extern "C" int cer_plane_stats_f32(const float* x, float* stats, long planes, long plane_size, float eps, void* stream) {
    if (!x || !stats || planes <= 0 || plane_size <= 0) return CER_EINVAL;
    if (!cer_aligned16(x)) return CER_EALIGN;
    hipLaunchKernelGGL(plane_stats_kernel, dim3((unsigned)planes), dim3(1024), 0, (hipStream_t)stream, x, stats, plane_size, eps);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// flags: 1 relu_a, 2 relu_b, 4 relu_out
__global__ __launch_bounds__(256) void norm_act_kernel(const float* __restrict__ x, const float* __restrict__ xs, const float* __restrict__ res,
                                                       const float* __restrict__ rs, float* __restrict__ out, long plane, int flags,
                                                       int chunks_per_plane) {
    const long pl = blockIdx.x / chunks_per_plane;
    const int ch = blockIdx.x % chunks_per_plane;
    const float mean = xs ? xs[2 * pl] : 0.f, rstd = xs ? xs[2 * pl + 1] : 1.f;
    const float rmean = rs ? rs[2 * pl] : 0.f, rrstd = rs ? rs[2 * pl + 1] : 1.f;
    const float* px = x + pl * plane;
    const float* pr = res ? res + pl * plane : nullptr;
    float* po = out + pl * plane;
    if (plane % 4 != 0) {                                  // scalar path for planes that are not 16-B multiples
        const long per1 = (plane + chunks_per_plane - 1) / chunks_per_plane;
        const long lo1 = ch * per1, hi1 = min(plane, lo1 + per1);
        for (long i = lo1 + threadIdx.x; i < hi1; i += 256) {
            float t = xs ? (px[i] - mean) * rstd : px[i];
            if (flags & 1) t = fmaxf(t, 0.f);
            if (pr) {
                float u = rs ? (pr[i] - rmean) * rrstd : pr[i];
                if (flags & 2) u = fmaxf(u, 0.f);
                t = t + u;
            }
            if (flags & 4) t = fmaxf(t, 0.f);
            po[i] = t;
        }
        return;
    }
    const long n4 = plane / 4;
    const long per = (n4 + chunks_per_plane - 1) / chunks_per_plane;
    const long lo = ch * per, hi = min(n4, lo + per);
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        float4 a = cer_ld4(px + 4 * i);
        float v[4] = {a.x, a.y, a.z, a.w};
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (pr) {
            const float4 b = cer_ld4(pr + 4 * i);
            r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = xs ? (v[k] - mean) * rstd : v[k];
            if (flags & 1) t = fmaxf(t, 0.f);
            if (pr) {
                float u = rs ? (r[k] - rmean) * rrstd : r[k];
                if (flags & 2) u = fmaxf(u, 0.f);
                t = t + u;
            }
            if (flags & 4) t = fmaxf(t, 0.f);
            v[k] = t;
        }
        *reinterpret_cast<float4*>(po + 4 * i) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

extern "C" int cer_norm_act_f32(const float* x, const float* x_stats, const float* res, const float* res_stats, float* out, long planes,
                                long plane_size, int flags, void* stream) {
    if (!x || !out || planes <= 0 || plane_size <= 0) return CER_EINVAL;
    if (!cer_aligned16(x) || !cer_aligned16(out) || (res && !cer_aligned16(res))) return CER_EALIGN;
    // ~64 KiB of fp32 per block
    int chunks = (int)((plane_size + 16383) / 16384);
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(norm_act_kernel, dim3((unsigned)(planes * chunks)), dim3(256), 0, (hipStream_t)stream, x, x_stats, res, res_stats, out,
                       plane_size, flags, chunks);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
