// K1, round 2: epipolar cost-volume build (view-sum fold), one cascade stage
// (reference: core/corr.py:56-91 + utils/projective_ops.py:5-28 + core/corr.py:28-43 +
//  alt_cuda_corr/correlation_kernel.cu:18-119, pyramid core/corr.py:94-97) as "band GEMM + 4-tap gather".
//
// Round 1 walked every sample with its own 2x2 x 256-B footprint: each source texel went through L1 ~250 times (46 GB of gather
// traffic at 1600x1184, texture-address units 78 % busy).  The correlation is bilinear in the TEXEL dot products:
//     <f1(p), bilerp(f2)(u, w)>  =  bilerp over the 4 texels of  <f1(p), f2(texel)>,
// and neighbouring reference pixels walk nearly the same epipolar band.  So, per tile of 8 x 4 reference pixels, source view and
// chunk of consecutive hypotheses:
//   1. every thread projects its (pixel, hypothesis) samples (same fp32 expressions as round 1) and the block reduces the
//      bounding box of the cells they touch; the chunk is halved until the box holds <= CG_NMAX texels (and grows again when
//      the box has room) - at 1600x1184 stage 0 settles at 8 hypotheses (~31 x 6 texels);
//   2. the box's texels are read ONCE, 32 at a time, straight into MFMA B fragments (fp32 -> hi|lo f16 in registers; the tile's
//      f1 rows are the A fragments, held for the whole tile) and  dots[texel][pixel]  goes to LDS: 12 MFMAs per 32 texels
//      (split-f16, three terms into one fp32 accumulator: fp32-class, see conv_s16.hip);
//   3. every sample gathers its 4 dots from LDS, applies the bilinear weights and adds into the tile's [32][D] accumulator in LDS.
// After the last view the tile's rows leave with the view-mean scale and the avg-pooled pyramid levels (fused, as round 1).
// A chunk whose box does not fit even for one hypothesis (projections blown apart near Z = 0, or per-pixel origins that
// scatter the tile's segments - a rough stage-1 disparity) takes a direct per-sample path: correct, slow.  The host therefore
// uses this kernel for the stage whose origins are uniform (shift = 1: every origin is clamped to the same value when the
// incoming disparity is 0) and keeps the round-1 walk (cost_build.hip) for the re-centred stages.
#include "common.hpp"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CG_TW 8
#define CG_TH 4
#define CG_NMAX 256                         // texels of a box (LDS: CG_NMAX x 36 floats; 52 KB per block with the row accumulator: 3 blocks per CU)
#define CG_DS 36                            // floats per texel row of the dot matrix (32 pixels + pad: conflict-free 16-B writes)
#define CG_LOG2S 6                          // operand scale of the split: features (already / 8) saturate at 65504 / 64

__device__ __forceinline__ void cg_split8(const float4 a, const float4 b, half8& hi, half8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    cer_h2 h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cer_f2 x = (cer_f2){v[2 * i], v[2 * i + 1]} * (float)(1 << CG_LOG2S);
        x = __builtin_elementwise_min(__builtin_elementwise_max(x, (cer_f2){-65504.0f, -65504.0f}), (cer_f2){65504.0f, 65504.0f});
        h[i] = __builtin_convertvector(x, cer_h2);
        l[i] = __builtin_convertvector(x - __builtin_convertvector(h[i], cer_f2), cer_h2);
    }
    hi = (half8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
    lo = (half8){l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
}

__global__ __launch_bounds__(256, 3) void cost_gemm_kernel(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                                                           const float* __restrict__ Pij, const float* __restrict__ disp_in,
                                                           float* __restrict__ vol, float* __restrict__ origin_out, int V, int h1, int w1,
                                                           int h2, int w2, int D, int rs, float incre, float lim, int shift, int accumulate,
                                                           int y0, int levels, float scale, int tiles_x) {
    constexpr int C = 64;
    extern __shared__ __attribute__((aligned(16))) float cg_smem[];
    float* dots = cg_smem;                                  // [CG_NMAX][CG_DS]
    float* accL = dots + CG_NMAX * CG_DS;                   // [32][Dp]   view sums (then the scaled row with its pooled levels)
    const int Dp = rs + 1;                                  // odd-ish pitch; rows hold level 0 | level 1 | ...
    float* pxd = accL + 32 * Dp;                            // per pixel: px, py, origin, valid
    int* box = reinterpret_cast<int*>(pxd + 32 * 4);        // [0..3] wave partials x 4, then the result

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kg = lane >> 5;
    // XCD-aware tile order (blocks are dealt round-robin over the 8 XCDs): each XCD gets a contiguous range of tiles
    int tile;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int ty0 = (tile / tiles_x) * CG_TH, tx0 = (tile % tiles_x) * CG_TW;
    const int half = D / 2;
    const int wp = w2 + 4;
    const long P2C = (long)(h2 + 4) * wp * C;

    // ---- per pixel: coordinates, origin (core/corr.py:59-62), validity; the view-sum accumulator starts at zero
    if (tid < 32) {
        const int y = ty0 + (tid >> 3), x = tx0 + (tid & 7);
        const bool ok = y < h1 && x < w1;
        const long p = (long)min(y, h1 - 1) * w1 + min(x, w1 - 1);
        float origin = disp_in[p];
        if (shift && origin < lim) origin = lim;
        if (ok && origin_out) origin_out[p] = origin;
        pxd[tid * 4 + 0] = (float)x;
        pxd[tid * 4 + 1] = (float)(y + y0);                // y0: first image row of a row slab (multi-GPU)
        pxd[tid * 4 + 2] = origin;
        pxd[tid * 4 + 3] = ok ? 1.f : 0.f;
    }
    for (int i = tid; i < 32 * Dp; i += 256) accL[i] = 0.f;
    // A fragments: f1 rows of the tile's pixels (lane: pixel li, channels 16 ks + 8 kg .. + 7), split once
    half8 ah[4], al[4];
    {
        const int y = ty0 + (li >> 3), x = tx0 + (li & 7);
        const bool ok = y < h1 && x < w1;
        const float* f1p = fmap1 + ((long)min(y, h1 - 1) * w1 + min(x, w1 - 1)) * C + 8 * kg;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float4 a = cer_ld4(f1p + 16 * ks), b = cer_ld4(f1p + 16 * ks + 4);
            if (!ok) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
            cg_split8(a, b, ah[ks], al[ks]);
        }
    }
    __syncthreads();
    const int spx = tid & 31, sg = tid >> 5;                // this thread's samples: pixel spx, hypotheses k0 + sg + 8 j
    const float px = pxd[spx * 4 + 0], py = pxd[spx * 4 + 1], origin = pxd[spx * 4 + 2];
    const float invS = 1.0f / (float)(1 << (2 * CG_LOG2S));

    for (int v = 0; v < V; ++v) {
        const float* m = Pij + v * 16;
        const float* f2v = fmap2 + (long)v * P2C;
        // ray of the pixel in the source view (utils/projective_ops.py:26-28): X = a0 + m3 * hyp, ...
        const float a0 = fmaf(m[1], py, m[0] * px) + m[2], a1 = fmaf(m[5], py, m[4] * px) + m[6], a2 = fmaf(m[9], py, m[8] * px) + m[10];
        int hint = 8;                                      // chunk size to try first: what worked last (doubled when the box had room)
        for (int k0 = 0; k0 < D;) {
            int HC = hint;
            while (HC > D - k0) HC >>= 1;                  // a power of two that fits the remaining hypotheses
            float du[4], dw[4];
            int ciu[4], ciw[4];
            bool slow = false;
            int bx0 = 0, by0 = 0, bw = 1, bh = 1;
            for (;;) {
                // ---- 1. project (same fp32 expressions as cost_build.hip / the reference) and reduce the box of the cells
                int mnu = 1 << 30, mxu = -(1 << 30), mnw = 1 << 30, mxw = -(1 << 30);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int hl = sg + 8 * j;
                    if (hl < HC) {
                        const float hyp = __fadd_rn(__fmul_rn((float)(k0 + hl - half), incre), origin);
                        const float X = fmaf(m[3], hyp, a0), Y = fmaf(m[7], hyp, a1), Z = fmaf(m[11], hyp, a2);
                        float u = X / Z, w = Y / Z;
                        const bool ok = (u == u) && (w == w);          // 0/0 samples nothing
                        u = fminf(fmaxf(u, -1e4f), 1e4f);
                        w = fminf(fmaxf(w, -1e4f), 1e4f);
                        const float fu = floorf(u), fw = floorf(w);
                        du[j] = ok ? u - fu : 0.f;
                        dw[j] = ok ? w - fw : 0.f;
                        ciu[j] = ok ? min(max((int)fu, -2), w2) : -2;
                        ciw[j] = ok ? min(max((int)fw, -2), h2) : -2;
                        mnu = min(mnu, ciu[j]); mxu = max(mxu, ciu[j]);
                        mnw = min(mnw, ciw[j]); mxw = max(mxw, ciw[j]);
                    }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    mnu = min(mnu, __shfl_xor(mnu, o)); mxu = max(mxu, __shfl_xor(mxu, o));
                    mnw = min(mnw, __shfl_xor(mnw, o)); mxw = max(mxw, __shfl_xor(mxw, o));
                }
                __syncthreads();                           // the previous chunk's readers of `box` / `dots` are done
                if (lane == 0) { box[wave * 4 + 0] = mnu; box[wave * 4 + 1] = mxu; box[wave * 4 + 2] = mnw; box[wave * 4 + 3] = mxw; }
                __syncthreads();
                bx0 = min(min(box[0], box[4]), min(box[8], box[12]));
                by0 = min(min(box[2], box[6]), min(box[10], box[14]));
                bw = max(max(box[1], box[5]), max(box[9], box[13])) - bx0 + 2;
                bh = max(max(box[3], box[7]), max(box[11], box[15])) - by0 + 2;
                if ((long)bw * bh <= CG_NMAX) break;
                if (HC == 1) { slow = true; break; }
                HC >>= 1;
            }
            const int N = bw * bh;
            hint = (2 * N <= CG_NMAX) ? min(2 * HC, 32) : HC;
            if (!slow) {
                // ---- 2. dots[texel][pixel] for the box: 32 texels per MFMA group, groups dealt to the 4 waves; the next group's
                // texels are requested before the current group is multiplied
                const int ngroups = (N + 31) >> 5;
                float4 t[8], tn[8];
                auto fetch = [&](float4 (&dst)[8], int gi) {
                    const int n = min(gi * 32 + li, N - 1);
                    const int tyb = n / bw, txb = n - tyb * bw;
                    const float* tp = f2v + ((long)(by0 + tyb + 2) * wp + (bx0 + txb + 2)) * C + 8 * kg;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) { dst[2 * ks] = cer_ld4(tp + 16 * ks); dst[2 * ks + 1] = cer_ld4(tp + 16 * ks + 4); }
                };
                if (wave < ngroups) fetch(tn, wave);
                for (int gi = wave; gi < ngroups; gi += 4) {
                    const int n = min(gi * 32 + li, N - 1);
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = tn[q];
                    if (gi + 4 < ngroups) fetch(tn, gi + 4);
                    floatx16 d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        half8 bh_, bl_;
                        cg_split8(t[2 * ks], t[2 * ks + 1], bh_, bl_);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh_, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl_, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh_, d, 0, 0, 0);
                    }
                    // d[r]: pixel (r&3) + 8(r>>2) + 4kg (row), texel n (column)
                    if (gi * 32 + li < N) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(dots + n * CG_DS + 8 * q + 4 * kg) =
                                make_float4(d[4 * q] * invS, d[4 * q + 1] * invS, d[4 * q + 2] * invS, d[4 * q + 3] * invS);
                    }
                }
                __syncthreads();
            }
            // ---- 3. samples: 4 dots, bilinear weights (correlation_kernel.cu:97-100), into the view sum
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hl = sg + 8 * j;
                if (hl < HC) {
                    float d00, d01, d10, d11;
                    if (!slow) {
                        const float* dp = dots + ((ciw[j] - by0) * bw + (ciu[j] - bx0)) * CG_DS + spx;
                        d00 = dp[0]; d01 = dp[CG_DS]; d10 = dp[bw * CG_DS]; d11 = dp[(bw + 1) * CG_DS];
                    } else {
                        // direct path: the four texel dots of this sample from global memory (exact fp32 fmaf chains)
                        const int yy = ty0 + (spx >> 3), xx = tx0 + (spx & 7);
                        const float* f1p = fmap1 + ((long)min(yy, h1 - 1) * w1 + min(xx, w1 - 1)) * C;
                        const float* tp = f2v + ((long)(ciw[j] + 2) * wp + (ciu[j] + 2)) * C;
                        d00 = d01 = d10 = d11 = 0.f;
                        for (int c = 0; c < C; ++c) {
                            const float f = f1p[c];
                            d00 = fmaf(f, tp[c], d00); d01 = fmaf(f, tp[C + c], d01);
                            d10 = fmaf(f, tp[(long)wp * C + c], d10); d11 = fmaf(f, tp[(long)wp * C + C + c], d11);
                        }
                    }
                    const float wy1 = dw[j], wy0 = 1.0f - dw[j], wx1 = du[j], wx0 = 1.0f - du[j];
                    const float val = d00 * (wy0 * wx0) + d01 * (wy0 * wx1) + d10 * (wy1 * wx0) + d11 * (wy1 * wx1);
                    accL[spx * Dp + k0 + hl] += val;
                }
            }
            k0 += HC;
        }
    }
    __syncthreads();
    // ---- rows out.  Fused pyramid (levels > 1; host guarantees !accumulate): level 0 = view sum * scale, then avg-pool pairs
    // (core/corr.py:94-97: (a + b) * 0.5, level by level); else the plain (or accumulated) view sum.
    if (levels > 1) {
        for (int i = tid; i < 32 * D; i += 256) accL[(i / D) * Dp + i % D] *= scale;
        int off = 0, n = D;
        for (int l = 1; l < levels; ++l) {
            __syncthreads();
            const int mlen = n / 2;
            for (int i = tid; i < 32 * mlen; i += 256) {
                const int pxi = i / mlen, e = i - pxi * mlen;
                const float* src = accL + pxi * Dp + off;
                accL[pxi * Dp + off + n + e] = (src[2 * e] + src[2 * e + 1]) * 0.5f;
            }
            off += n;
            n = mlen;
        }
        __syncthreads();
        const int used = off + n;
        for (int i = tid; i < 32 * used; i += 256) {
            const int pxi = i / used, e = i - pxi * used;
            if (pxd[pxi * 4 + 3] != 0.f) {
                const int y = ty0 + (pxi >> 3), x = tx0 + (pxi & 7);
                vol[((long)y * w1 + x) * rs + e] = accL[pxi * Dp + e];
            }
        }
    } else {
        for (int i = tid; i < 32 * D; i += 256) {
            const int pxi = i / D, e = i - pxi * D;
            if (pxd[pxi * 4 + 3] != 0.f) {
                const int y = ty0 + (pxi >> 3), x = tx0 + (pxi & 7);
                float* o = vol + ((long)y * w1 + x) * rs + e;
                *o = accumulate ? (*o + accL[pxi * Dp + e]) : accL[pxi * Dp + e];
            }
        }
    }
}

// fold modes (1: write, 2: accumulate) of cer_cost_build_f32 at C = 64; returns CER_ESHAPE if this kernel does not apply
int cer_cost_gemm_launch(const float* f1, const float* f2, const float* Pij, const float* disp_in, float* vol, float* origin_out, int V, int h1,
                         int w1, int h2, int w2, int C, int D, int rs, double incre_d, int shift, int mode, int y0, int levels, float scale,
                         hipStream_t st) {
    if (C != 64 || (mode != 1 && mode != 2) || D > 192) return CER_ESHAPE;
    const int tiles_x = (w1 + CG_TW - 1) / CG_TW, tiles_y = (h1 + CG_TH - 1) / CG_TH;
    const size_t smem = sizeof(float) * (CG_NMAX * CG_DS + 32 * (rs + 1) + 32 * 4) + sizeof(int) * 16;
    if (smem > 53 * 1024) return CER_ESHAPE;
    const float lim = (float)((D / 2) * incre_d);
    hipLaunchKernelGGL(cost_gemm_kernel, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), smem, st, f1, f2, Pij, disp_in, vol, origin_out, V, h1, w1, h2,
                       w2, D, rs, (float)incre_d, lim, shift, mode == 2 ? 1 : 0, y0, levels, scale, tiles_x);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
