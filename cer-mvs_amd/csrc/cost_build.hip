// K1: epipolar cost-volume build, one cascade stage, fused
// (reference: core/corr.py:56-91 + utils/projective_ops.py:5-28 + core/corr.py:28-43 +
//  alt_cuda_corr/correlation_kernel.cu:18-119), and the correlation pyramid (core/corr.py:94-97).
//
// v2 mapping - one WAVE per reference pixel:
//   * projection: lane l computes the source coordinate of hypothesis kb + l, so all (up to) 64
//     hypotheses of a view are projected by ONE pass of straight-line VALU code (two IEEE divisions
//     per lane instead of per sample-row); no coordinate tensor exists (projective_ops.py:13-28
//     materialises 3 x 121 MB per view at 1600x1184);
//   * sampling: the wave then walks the 64 samples; a sample's integer texel, fractions and
//     validity are pulled out of lane j with v_readlane into SGPRs, so addressing and the
//     "same texel as the previous hypothesis?" test run on the scalar unit.  The four 16-lane rows
//     of the wave own the four bilinear corners, lane (row, sub) owns channel quad `sub` of its
//     corner: one 16-B load per lane fetches the whole 2x2 footprint (4 x 256 B) of the sample;
//   * source maps carry a 2-texel zero border so that clamped cells never need a bounds test;
//   * reuse: when consecutive hypotheses fall into the same texel cell (stage 1: 0.17-0.56 texel
//     steps) the lane-partial dot is reused and only the bilinear weights change (bilinearity:
//     <f1, bilerp(f2)> = bilerp(<f1, f2_texel>)) - no load, no dot;
//   * accumulation: acc[j] (one VGPR per hypothesis) sums weighted lane-partials over all local
//     views; a single 63-shuffle butterfly then leaves the total of hypothesis k on lane k, and the
//     wave writes its pixel's row with one coalesced 256-B store.
#include "common.hpp"

// NACC: accumulator registers per lane (hypotheses per 64-block that can be live): 48 for D <= 48 (stage 1: D = 44) keeps the
// kernel under 128 VGPRs = 4 waves per SIMD instead of 3; the walk is latency-bound, so the extra wave is throughput.
template <int NQ, bool SUM, int NACC>
__global__ __launch_bounds__(256) void cost_build_kernel(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                                                         const float* __restrict__ Pij, const float* __restrict__ disp_in,
                                                         float* __restrict__ vol, float* __restrict__ origin_out, int V, int h1, int w1,
                                                         int h2, int w2, int C, int D, int rs, float incre, float lim, int shift,
                                                         int accumulate, int y0, int levels, float scale) {
    const int lane = threadIdx.x & 63;
    const long P = (long)h1 * w1;
    // XCD-aware block order: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only), so give each XCD a
    // contiguous band of image rows - its private 4 MiB L2 then sees 1/8 of every source map instead of all of it.
    const unsigned nblk = gridDim.x, q8 = nblk / 8, r8 = nblk % 8, xcd = blockIdx.x % 8, within = blockIdx.x / 8;
    const unsigned bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const long p = (long)bid * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform pixel index in SGPRs
    if (p >= P) return;                                    // wave-uniform
    const int sub = lane & 15, cx = (lane >> 4) & 1, cy = lane >> 5;
    const float px = (float)(p % w1), py = (float)(p / w1 + y0);     // y0: first image row of a row slab (multi-GPU)
    float origin = disp_in[p];
    if (shift && origin < lim) origin = lim;
    if (origin_out && lane == 0 && blockIdx.y == 0) origin_out[p] = origin;
    float4 f1q[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) f1q[q] = cer_ld4(fmap1 + p * C + 4 * sub + 64 * q);
    const int half = D / 2;
    // source maps carry a 2-texel zero border: [V, h2+4, w2+4, C].  Clamping the integer texel to
    // [-2, w2] x [-2, h2] keeps all four bilinear corners inside the padded map and reproduces the
    // reference's "texel outside -> 0" (correlation_kernel.cu:80-83) without any bounds test here.
    const int wp = w2 + 4;
    const long P2C = (long)(h2 + 4) * wp * C;
    const int loff = (cy * wp + cx) * C + 4 * sub;         // this lane's corner + channel quad
    __shared__ __attribute__((aligned(16))) float wl_all[4][4 * 64];     // per wave: [corner][hypothesis] bilinear weights
    float* wl = wl_all[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    const int corner = cy * 2 + cx;
    const int v_lo = SUM ? 0 : (int)blockIdx.y, v_hi = SUM ? V : (int)blockIdx.y + 1;
    float* orow = SUM ? vol + p * rs : vol + ((long)blockIdx.y * P + p) * rs;

    for (int kb = 0; kb < D; kb += 64) {
        float acc[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = 0.f;
        const int nk = min(64, D - kb);
        for (int v = v_lo; v < v_hi; ++v) {
            const float* m = Pij + v * 16;
            // ---- projection of hypothesis kb + lane (utils/projective_ops.py:26-28, core/corr.py:56,65,88)
            const float hyp = __fadd_rn(__fmul_rn((float)(kb + lane - half), incre), origin);
            const float X = fmaf(m[3], hyp, fmaf(m[1], py, m[0] * px) + m[2]);
            const float Y = fmaf(m[7], hyp, fmaf(m[5], py, m[4] * px) + m[6]);
            const float Z = fmaf(m[11], hyp, fmaf(m[9], py, m[8] * px) + m[10]);
            float u = X / Z, w = Y / Z;
            const bool ok = (u == u) && (w == w);          // 0/0 samples nothing
            u = fminf(fmaxf(u, -1e4f), 1e4f);
            w = fminf(fmaxf(w, -1e4f), 1e4f);
            const float fu = floorf(u), fw = floorf(w);
            const float du = ok ? u - fu : 0.f, dw = ok ? w - fw : 0.f;
            const int iu = ok ? min(max((int)fu, -2), w2) : -2, iw = ok ? min(max((int)fw, -2), h2) : -2;
            const int off = ((iw + 2) * wp + (iu + 2)) * C;    // element offset of the cell's top-left texel
            // bilinear corner weights of hypothesis `lane`, transposed through LDS: afterwards every lane reads the weights of
            // ITS corner for 8 consecutive samples with two 16-B loads (broadcast within the 16-lane row), so that a sample
            // costs one v_readlane (cell offset) + one fma instead of rebuilding its weight on every lane.
            {
                const float wy1 = dw, wy0 = 1.0f - dw, wx1 = du, wx0 = 1.0f - du;
                __builtin_amdgcn_s_waitcnt(0xC07F);        // the previous view's weight reads of this wave are done
                wl[0 * 64 + lane] = wy0 * wx0;
                wl[1 * 64 + lane] = wy0 * wx1;
                wl[2 * 64 + lane] = wy1 * wx0;
                wl[3 * 64 + lane] = wy1 * wx1;
                __builtin_amdgcn_s_waitcnt(0xC07F);        // wave-local exchange: LDS ops of a wave complete in order
            }
            // "new texel cell?" for all 64 hypotheses at once (lane = hypothesis against lane - 1; hypothesis 0 always loads): one
            // shuffle + compare + ballot per view replace a scalar compare / select chain per sample - the walk is instruction
            // -issue bound (with every load removed it still needed 1.24 ms at stage 0), most of it scalar bookkeeping.
            const int prev_off = __shfl_up(off, 1);
            const unsigned long long newcell = __ballot(lane == 0 || off != prev_off);
            // byte offsets fit 32 bits (one padded source map), so a load is scalar base + 32-bit vector offset
            const char* vbase = reinterpret_cast<const char*>(fmap2 + (long)v * P2C);
            float sdot = 0.f;
            // samples are walked in groups of 8: first all (needed) texel loads of the group are issued,
            // then consumed - 8 loads in flight per wave instead of one dependent load per sample
#pragma unroll
            for (int g = 0; g < NACC / 8; ++g) {
                if (g * 8 < nk) {                          // wave-uniform
                    float4 t[8][NQ];
                    const float4 wa = *reinterpret_cast<const float4*>(wl + corner * 64 + g * 8);
                    const float4 wb = *reinterpret_cast<const float4*>(wl + corner * 64 + g * 8 + 4);
                    const float wgt[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                    const unsigned bits = (unsigned)(newcell >> (g * 8)) & 0xFFu;        // scalar
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (bits & (1u << i)) {
                            const unsigned bo = (unsigned)(__builtin_amdgcn_readlane(off, g * 8 + i) + loff) * 4u;
#pragma unroll
                            for (int q = 0; q < NQ; ++q) t[i][q] = cer_ld4(reinterpret_cast<const float*>(vbase + bo) + 64 * q);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (bits & (1u << i)) {
                            sdot = 0.f;
#pragma unroll
                            for (int q = 0; q < NQ; ++q) sdot = cer_dot4(f1q[q], t[i][q], sdot);
                        }
                        acc[g * 8 + i] = fmaf(sdot, wgt[i], acc[g * 8 + i]);
                    }
                }
            }
        }
        // ---- butterfly: after the six exchange steps lane l holds the wave total of acc[l]
#pragma unroll
        for (int hb = 32; hb >= 1; hb >>= 1) {
            const bool up = (lane & hb) != 0;
#pragma unroll
            for (int i = 0; i < hb; ++i) {
                if (i >= NACC) continue;                   // (compile time) hypotheses beyond NACC do not exist
                const float hi = (i + hb < NACC) ? acc[i + hb] : 0.f;
                const float send = up ? acc[i] : hi;
                const float keep = up ? hi : acc[i];
                acc[i] = keep + __shfl_xor(send, hb);
            }
        }
        if (levels >= 1) {
            // fused epilogue (host guarantees SUM, !accumulate, D <= 64): level 0 = view sum * scale (levels == 1: that is all - the
            // compact rows of round 5, same meaning as in cer_cost_lines_reduce_f32), then avg-pool pairs
            // (core/corr.py:94-97) with one exchange per level: element j of level l lives on lane j << l
            float cur = acc[0] * scale;
            if (lane < nk) orow[lane] = cur;
            int off = 0, n = D;
            for (int l = 1; l < levels; ++l) {
                const int m = n / 2;
                const float other = __shfl_xor(cur, 1 << (l - 1));
                cur = (cur + other) * 0.5f;
                off += n;
                if ((lane & ((1 << l) - 1)) == 0 && (lane >> l) < m) orow[off + (lane >> l)] = cur;
                n = m;
            }
        } else if (lane < nk) {
            float* o = orow + kb + lane;
            *o = accumulate ? (*o + acc[0]) : acc[0];
        }
    }
}

template <int NQ>
static int launch_build(const float* f1, const float* f2, const float* Pij, const float* disp_in, float* vol, float* origin_out, int V,
                        int h1, int w1, int h2, int w2, int C, int D, int rs, double incre_d, int shift, int mode, int y0, int levels, float scale,
                        hipStream_t st) {
    const long P = (long)h1 * w1;
    const float lim = (float)((D / 2) * incre_d);
    const float incre = (float)incre_d;
    const unsigned gx = (unsigned)((P + 3) / 4);
    if (mode == 0)
        hipLaunchKernelGGL((cost_build_kernel<NQ, false, 64>), dim3(gx, (unsigned)V), dim3(256), 0, st, f1, f2, Pij, disp_in, vol, origin_out, V,
                           h1, w1, h2, w2, C, D, rs, incre, lim, shift, 0, y0, 0, 1.0f);
    else if (D <= 48)
        hipLaunchKernelGGL((cost_build_kernel<NQ, true, 48>), dim3(gx, 1), dim3(256), 0, st, f1, f2, Pij, disp_in, vol, origin_out, V, h1, w1,
                           h2, w2, C, D, rs, incre, lim, shift, mode == 2 ? 1 : 0, y0, levels, scale);
    else
        hipLaunchKernelGGL((cost_build_kernel<NQ, true, 64>), dim3(gx, 1), dim3(256), 0, st, f1, f2, Pij, disp_in, vol, origin_out, V, h1, w1,
                           h2, w2, C, D, rs, incre, lim, shift, mode == 2 ? 1 : 0, y0, levels, scale);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

static int g_cost_build_algo = 0;
extern "C" int cer_cost_build_algo(int algo) {             // 0: automatic (host layer prefers cost_lines.hip), 1: always this file's walk
    const int prev = g_cost_build_algo;
    if (algo >= 0 && algo <= 1) g_cost_build_algo = algo;
    return prev;
}

extern "C" int cer_cost_build_f32(const float* fmap1, const float* fmap2, const float* Pij, const float* disp_in, float* vol,
                                  float* origin_out, int V, int h1, int w1, int h2, int w2, int C, int D, int row_stride, double incre,
                                  int shift, int mode, int y0, int fuse_levels, float fuse_scale, void* stream) {
    if (!fmap1 || !fmap2 || !Pij || !disp_in || !vol) return CER_EINVAL;
    if (fuse_levels >= 1) {                                // fused epilogue (1: scale only): view-sum fold of a single 64-hypothesis block only
        if (mode != 1 || D > 64) return CER_EINVAL;
        int need = 0, n = D;
        for (int l = 0; l < fuse_levels; ++l) { need += n; n /= 2; }
        if (row_stride < need || fuse_levels > 6) return CER_ESHAPE;
    }
    if (V <= 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || C <= 0 || D <= 0 || row_stride < D || mode < 0 || mode > 2) return CER_EINVAL;
    if (C % 64 != 0 || C > 256 || V > 65535) return CER_ESHAPE;
    if ((long)(h2 + 4) * (w2 + 4) * C >= (1L << 30)) return CER_ESHAPE;      // the walk forms 32-bit BYTE offsets into one padded map
    if (!cer_aligned16(fmap1) || !cer_aligned16(fmap2)) return CER_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    switch (C / 64) {
        case 1: return launch_build<1>(fmap1, fmap2, Pij, disp_in, vol, origin_out, V, h1, w1, h2, w2, C, D, row_stride, incre, shift, mode, y0, fuse_levels, fuse_scale, st);
        case 2: return launch_build<2>(fmap1, fmap2, Pij, disp_in, vol, origin_out, V, h1, w1, h2, w2, C, D, row_stride, incre, shift, mode, y0, fuse_levels, fuse_scale, st);
        case 3: return launch_build<3>(fmap1, fmap2, Pij, disp_in, vol, origin_out, V, h1, w1, h2, w2, C, D, row_stride, incre, shift, mode, y0, fuse_levels, fuse_scale, st);
        default: return launch_build<4>(fmap1, fmap2, Pij, disp_in, vol, origin_out, V, h1, w1, h2, w2, C, D, row_stride, incre, shift, mode, y0, fuse_levels, fuse_scale, st);
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

// Feature maps for the cost build in one pass: NCHW [N,C,h,w] -> channels-last [N,(h+2b)*(w+2b),C] * scale with a
// b-texel zero border (replaces torch's divide + pad + permute + contiguous: 4 passes over 300 MB at 1600x1184).
__global__ __launch_bounds__(256) void nchw_to_nhwc_border_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int h, int w, int b,
                                                                  float scale) {
    __shared__ float tile[64][65];
    const int wp = w + 2 * b, hp = h + 2 * b;
    const long Pp = (long)hp * wp, P = (long)h * w;
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const float* s = src + (long)blockIdx.z * C * P;
    float* d = dst + (long)blockIdx.z * Pp * C;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    {
        const long pp = p0 + tx;
        const int y = (int)(pp / wp) - b, x = (int)(pp % wp) - b;
        const bool inside = pp < Pp && y >= 0 && y < h && x >= 0 && x < w;
        for (int i = ty; i < 64; i += 4) {
            const int c = c0 + i;
            tile[i][tx] = (inside && c < C) ? s[(long)c * P + (long)y * w + x] * scale : 0.f;
        }
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const long pp = p0 + i;
        const int c = c0 + tx;
        if (pp < Pp && c < C) d[pp * C + c] = tile[tx][i];
    }
}

extern "C" int cer_nchw_to_nhwc_border_f32(const float* src, float* dst, int N, int C, int h, int w, int border, float scale, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || h <= 0 || w <= 0 || border < 0) return CER_EINVAL;
    if (N > 65535) return CER_ESHAPE;
    const long Pp = (long)(h + 2 * border) * (w + 2 * border);
    hipLaunchKernelGGL(nchw_to_nhwc_border_kernel, dim3((unsigned)((Pp + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, src, dst, C, h, w, border, scale);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// ---- multi-segment copy (row-slab halo pack / refresh): blockIdx.y = segment, grid-stride over the segment
struct CopySegs {
    const float* src[CER_COPY_MAX_SEG];
    float* dst[CER_COPY_MAX_SEG];
    long n[CER_COPY_MAX_SEG];
};

__global__ __launch_bounds__(256) void copy_segments_kernel(const CopySegs a) {
    const int s = blockIdx.y;
    const float* __restrict__ src = a.src[s];
    float* __restrict__ dst = a.dst[s];
    const long n = a.n[s];
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const long stride = (long)gridDim.x * 256;
    if (vec) {
        const long n4 = n >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    }
}

extern "C" int cer_copy_segments_f32(const cer_copy_segments* seg, void* stream) {
    if (!seg) return CER_EINVAL;
    CopySegs a;
    long nmax = 0;
    for (int i = 0; i < CER_COPY_MAX_SEG; ++i) {
        if (seg->n[i] < 0 || (seg->n[i] > 0 && (!seg->src[i] || !seg->dst[i]))) return CER_EINVAL;
        a.src[i] = seg->src[i];
        a.dst[i] = seg->dst[i];
        a.n[i] = seg->n[i];
        nmax = seg->n[i] > nmax ? seg->n[i] : nmax;
    }
    if (nmax == 0) return CER_OK;
    const long blocks = (nmax / 4 + 255) / 256;
    dim3 grid((unsigned)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks)), CER_COPY_MAX_SEG);
    hipLaunchKernelGGL(copy_segments_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
