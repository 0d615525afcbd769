// K1, round 3: epipolar cost-volume build on EPIPOLAR-LINE TILES
// (reference: core/corr.py:56-91 + utils/projective_ops.py:5-28 + core/corr.py:28-43 +
//  alt_cuda_corr/correlation_kernel.cu:18-119; pyramid core/corr.py:94-97).
//
// The round-1/2 walk (cost_build.hip) fetches a 2 x 2 x 256-B footprint per sample: 1 KiB through the L1 per (pixel, view,
// hypothesis) - 46 GB per launch at 1600x1184, which IS its run time (64 B/clk/CU).  The correlation is bilinear in the
// texel dot products,
//     <f1(p), bilerp(f2)(u, w)>  =  bilerp over the 4 texels of  <f1(p), f2(texel)>,
// and all samples of a reference pixel lie on ONE line of the source image (its epipolar line); reference pixels on the same
// reference epipolar line share that source line.  So the reference image is partitioned, per source view, into tiles of 32
// pixels that follow the view's epipolar direction (digital lines: a shear of the pixel grid, chosen per view from Pij by
// cost_lines_setup_kernel), and per (view, tile) - one 256-thread block:
//   1. the tile's samples live in a thin band of the source map: columns along the band's major axis x R texels across
//      (R = 4..6 when the shear fits; measured 3.9 / 3.6 on the bench scene).  Wave 0 derives the band (major axis, line,
//      R, column range, travel direction) from the end points of every pixel's segment; the band is walked in chunks of
//      CL_T = 128 texels (measured: 6.9 chunks per tile at stage 0, 2.8 at stage 1);
//   2. per chunk, dots[texel][pixel] = <f2(texel), f1(pixel)> for ALL 128 x 32 pairs: each wave one MFMA tile of 32 texels x
//      32 pixels x 64 channels (split-f16: three f16 MFMAs into one fp32 accumulator, fp32-class - conv_s16.hip).  Operands
//      are pre-split hi|lo f16 PLANES (cer_feat_split_f16) read straight from global memory in fragment order, the next
//      chunk's while the current one is gathered: every band texel is read once per tile, nothing is staged;
//   3. every sample was projected ONCE, before the chunk loop, by straight-line code (two IEEE divisions, cell, fractions,
//      band rows packed into 16 B of LDS per sample); in the loop lane (pixel, hypothesis phase) walks its samples in order
//      (a cursor) and, for each one whose cell lies in the chunk, reads its 4 dots from LDS (16 B instead of 1 KiB),
//      applies the bilinear weights and stores the value; samples outside the source image are zeros (the walk's zero border);
//   4. anything the band analysis did not cover (projections blown apart near Z = 0, epipoles inside the image, rough
//      per-pixel origins that break the cursor's monotonic order, bands wider than 29 texels) takes a per-sample direct path:
//      correct, slow, rare.
// Views are independent (their tiles differ), so each (view, tile) writes its [32 px][D] result to a per-view partial
// volume; cost_lines_reduce_kernel sums the partials in view order (deterministic), applies the view-mean scale and emits
// the pooled pyramid levels - the fused epilogue of the walk.
//
// The coordinate arithmetic (hypothesis, projection, IEEE divisions, clamps, floor, fractions) is expression-for-expression
// that of cost_build.hip, so cells and weights are bit-identical to the walk; only the 64-channel dot differs in rounding
// (4e-8 relative L1 on the bench scene).  Measured, what shaped it and what is left: DESIGN.md section 3e.
#include "common.hpp"
#include <atomic>
#include <string.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CL_T 128                            // texels per chunk (one 32-texel MFMA row tile per wave)
#define CL_DP 33                            // pitch (float2) of the [hypothesis][pixel] sample / output tile
#define CL_LOG2S 6                          // operand scale of the split: features (already / 8) saturate at 65504 / 64
#define CL_LG 16                            // lines per tile-order group (4 .. 64 measured alike)
#define CL_RMAX 32                          // widest band (texels across) that still goes through the MFMA path

// ---- fp32 rows -> split-f16 operand planes.  Per block of `bt` texels (one source view, or the reference map):
//   8 planes p = hl * 4 + ks (hl: 0 = hi, 1 = lo half of x * 2^CL_LOG2S; ks: 16-channel group), each [bt][16 halves]:
//   halves of (block b, texel t, plane p, channel 16 ks + c) at  ((b * 8 + p) * bt + t) * 16 + c.
// One MFMA fragment load (fixed plane) of a wave then reads 32 B per texel, and x-neighbouring texels are contiguous: a band chunk
// of 8 columns x 4 rows touches 8 cache lines per load instruction instead of 32 with texel-major 256-B rows - the L1's line
// rate, not its byte rate, bounded the kernel (1.65 -> see DESIGN.md).
__global__ __launch_bounds__(256) void feat_split_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long bt, long n2, int* __restrict__ flag) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;    // (texel of the whole tensor, kg): a wave reads 32 whole 256-B rows
    if (i >= n2) return;
    const int kg = (int)(i & 1);
    const long texel = i >> 1, blk = texel / bt, t = texel - blk * bt;
    const float* sp = src + texel * 64 + kg * 8;
    _Float16* dp = dst + (blk * 8 * bt + t) * 16 + kg * 8;
    bool sat = false;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = cer_ld4(sp + ks * 16), b = cer_ld4(sp + ks * 16 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        cer_h2 h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cer_f2 x = (cer_f2){v[2 * j], v[2 * j + 1]} * (float)(1 << CL_LOG2S);
            sat |= !(fabsf(x.x) <= 65504.0f) || !(fabsf(x.y) <= 65504.0f);
            x = __builtin_elementwise_min(__builtin_elementwise_max(x, (cer_f2){-65504.0f, -65504.0f}), (cer_f2){65504.0f, 65504.0f});
            h[j] = __builtin_convertvector(x, cer_h2);
            l[j] = __builtin_convertvector(x - __builtin_convertvector(h[j], cer_f2), cer_h2);
        }
        *reinterpret_cast<half8*>(dp + ks * bt * 16) = (half8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
        *reinterpret_cast<half8*>(dp + (4 + ks) * bt * 16) = (half8){l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
    }
    if (sat && flag) atomicOr(flag, 1);                     // sticky: a feature beyond +-1023 was clamped (or is not finite)
}

extern "C" int cer_feat_split_f16(const float* src, void* dst, long blocks, long block_texels, int C, int* overflow_flag, void* stream) {
    if (!src || !dst || blocks <= 0 || block_texels <= 0) return CER_EINVAL;
    if (C != 64) return CER_ESHAPE;
    if (!cer_aligned16(src) || !cer_aligned16(dst)) return CER_EALIGN;
    const long n2 = blocks * block_texels * 2;
    hipLaunchKernelGGL(feat_split_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst,
                       block_texels, n2, overflow_flag);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// ---- per view: the tile partition of the reference image.  params[v] = {axis, shear, 0, 0}: tiles run along x (axis 0) or
// y (axis 1); pixel `a` along the axis of line j sits at b = j + rint(shear * (a - centre)) across it.
// Reference epipole e_h = adj(A) t with A = Pij[:3,:3], t = Pij[:3,3] (A e ~ t: the pixel whose ray passes through the
// source camera); the epipolar direction at the grid centre c is e_h.xy - c * e_h.z (finite or not).
__global__ void cost_lines_setup_kernel(const float* __restrict__ Pij, float* __restrict__ params, int v0, int nv, int h1, int w1, int y0) {
    const int v = v0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= v0 + nv) return;
    const float* m = Pij + v * 16;
    const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
    const double t0 = m[3], t1 = m[7], t2 = m[11];
    const double c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const double c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const double ex = c00 * t0 + c01 * t1 + c02 * t2, ey = c10 * t0 + c11 * t1 + c12 * t2, ez = c20 * t0 + c21 * t1 + c22 * t2;
    const double cx = 0.5 * w1, cy = y0 + 0.5 * h1;
    double dx = ex - cx * ez, dy = ey - cy * ez;
    if (!(dx == dx) || !(dy == dy) || fabs(dx) > 1e300 || fabs(dy) > 1e300) { dx = 1.0; dy = 0.0; }
    int axis = 0;
    double s = 0.0;
    if (fabs(dx) >= fabs(dy)) {
        axis = 0;
        s = dx != 0.0 ? dy / dx : 0.0;
    } else {
        axis = 1;
        s = dx / dy;
    }
    if (!(s == s)) s = 0.0;
    s = fmin(1.0, fmax(-1.0, s));
    params[v * 4 + 0] = (float)axis;
    params[v * 4 + 1] = (float)s;
    params[v * 4 + 2] = 0.f;
    params[v * 4 + 3] = 0.f;
}

// wave-wide min / max: 4 DPP permutes within each 16-lane row, then the four row results through SGPRs (uniform result)
__device__ __forceinline__ float cl_wmin(float x) {
    x = fminf(x, cer_dpp<0xB1>(x));
    x = fminf(x, cer_dpp<0x4E>(x));
    x = fminf(x, cer_dpp<0x141>(x));
    x = fminf(x, cer_dpp<0x140>(x));
    const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)), s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)), s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return fminf(fminf(s0, s1), fminf(s2, s3));
}
__device__ __forceinline__ float cl_wmax(float x) {
    x = fmaxf(x, cer_dpp<0xB1>(x));
    x = fmaxf(x, cer_dpp<0x4E>(x));
    x = fmaxf(x, cer_dpp<0x141>(x));
    x = fmaxf(x, cer_dpp<0x140>(x));
    const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)), s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)), s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
}

// direct path: the four texel dots of one sample from the split planes in global memory (scaled by 2^(2 CL_LOG2S), like the MFMA path).
// f1t / t00: the (block, texel) base of plane 0 (hi, channels 0-15); ps1 / ps2: plane strides in halves of the two maps
__device__ __noinline__ float cl_direct(const _Float16* __restrict__ f1t, long ps1, const _Float16* __restrict__ t00, long ps2, int smajS, int sminS,
                                        float wm0, float wm1, float wn0, float wn1, bool f2lo = true) {      // f2lo = false: the two-term form (source texels as f16)
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    const _Float16* tp[4] = {t00, t00 + (long)smajS * 16, t00 + (long)sminS * 16, t00 + (long)(smajS + sminS) * 16};
    for (int c8 = 0; c8 < 8; ++c8) {
        const int ks = c8 >> 1, kg = c8 & 1;
        const half8 ah = *reinterpret_cast<const half8*>(f1t + ks * ps1 + kg * 8), al = *reinterpret_cast<const half8*>(f1t + (4 + ks) * ps1 + kg * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const half8 bh = *reinterpret_cast<const half8*>(tp[q] + ks * ps2 + kg * 8), bl = *reinterpret_cast<const half8*>(tp[q] + (4 + ks) * ps2 + kg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[q] = fmaf((float)ah[e] + (float)al[e], f2lo ? (float)bh[e] + (float)bl[e] : (float)bh[e], d[q]);
        }
    }
    return d[0] * (wn0 * wm0) + d[1] * (wn0 * wm1) + d[2] * (wn1 * wm0) + d[3] * (wn1 * wm1);
}

// ---- cycle statistics of the one-line kernel (variant build -DCL_STATS=1: tools/archive/r05/mkvariant.sh clstats cost_lines.hip -DCL_STATS=1;
// tools/archive/r05/stats_cost_lines1.py).  Wave 0's view of a tile's phases, summed over all tiles with atomics.
#ifndef CL_STATS
#define CL_STATS 0
#endif

#if CL_STATS
#define CL_SLOTS 4096
__device__ unsigned long long cl_stats[CL_SLOTS][16];      // hashed by block: no contention on the counters; the host sums the slots
#define CL_CLK() __builtin_readcyclecounter()
// (durations are collected in registers of thread 0 and added ONCE per tile, behind its last store: atomics inside the chunk loop would sit in
// the same vmcnt queue as the fragment loads)
#define CL_STAT(i_, v_) do { if (threadIdx.x == 0) cl_acc[i_] += (unsigned long long)(v_); } while (0)
#define CL_FLUSH() do { if (threadIdx.x == 0) for (int q_ = 0; q_ < 16; ++q_) if (cl_acc[q_]) atomicAdd(&cl_stats[blockIdx.x & (CL_SLOTS - 1)][q_], cl_acc[q_]); } while (0)
extern "C" int cer_cost_lines1_stats(unsigned long long* out, int reset) {
    if (!out) return CER_EINVAL;
    static unsigned long long hostbuf[CL_SLOTS][16];
    hipError_t e = hipMemcpyFromSymbol(hostbuf, HIP_SYMBOL(cl_stats), sizeof(hostbuf));
    if (e != hipSuccess) return (int)e;
    for (int q = 0; q < 16; ++q) { out[q] = 0; for (int sl = 0; sl < CL_SLOTS; ++sl) out[q] += hostbuf[sl][q]; }
    if (reset) {
        memset(hostbuf, 0, sizeof(hostbuf));
        e = hipMemcpyToSymbol(HIP_SYMBOL(cl_stats), hostbuf, sizeof(hostbuf));
        if (e != hipSuccess) return (int)e;
    }
    return CER_OK;
}
#else
#define CL_CLK() 0ull
#define CL_STAT(i_, v_) do { (void)sizeof(v_); } while (0)
#define CL_FLUSH() do { } while (0)
#endif

// The band of one (view, segment, line) tile: what cl_tile's chunk loop, projection and gather need to know about where the tile's samples
// lie in the source map.  Round 6: computed by cost_lines_bands_kernel in a pass of its own (one WAVE per tile) instead of by wave 0 of
// every tile block while its other three waves sat at a barrier (profiles/r05_cost_lines_phases.txt: 15.3 k of a tile's 68 k cycles).
struct ClBand {
    int smaj, nchunks, R, Wc, cmin, cmax, dir;      // nchunks < 0: the tile has no pixel (the tile block returns at once)
    float bm, bl0;                                    // base row of band column c: floor(bl0 + bm * c)
    int pad[3];
};
static_assert(sizeof(ClBand) == 48, "ClBand is read as three 16-byte pieces");

struct ClArgs {
    const _Float16* f1s;      // [8 planes][P][16]   reference map of this call's pixel grid (cer_feat_split_f16 layout)
    const _Float16* f2s;      // [V][8 planes][(h2+4)*(w2+4)][16]   zero border included
    const float* Pij;         // [V][16]
    const float* params;      // [V][4]  cost_lines_setup_kernel
    const float* disp_in;     // [P]
    float* part;              // [V][P][D]  per-view partial volume, scaled by 2^(2 CL_LOG2S)
    const int* slot;          // [V] or null: view v's rows are block slot[v] of f2s (sharded forward: the gathered layout)
    int V, v0, h1, w1, h2, w2, D;       // V: views in the workspace; this launch builds views v0 .. v0 + gridDim.x / tpv - 1
    float incre, lim;
    int shift, y0, tpv;
    int tpv8;                 // blocks per view of cost_lines8_kernel (8 lines x one segment each)
    unsigned long long* todo; // [0]: count, then (view, segment, line) of the lines cost_lines8_kernel left to the one-line form
    ClBand* bands;            // [V][tpv] band records, indexed by (absolute view, tile order inside the view)
};

// ---- what a lane knows about its pixel slot of a (view, segment, line) tile: the pixel, its hypothesis origin (core/corr.py:59-62) and its
// ray (utils/projective_ops.py:26-28).  Shared by the band pass and the tile kernel: the same expressions, so the same bits.
struct ClLane {
    bool valid;
    int x_me, y_me;
    long p_me;
    float origin, a0, a1, a2, m3, m7, m11, incre;
    int half;
    // same fp32 expressions as cost_build.hip / the reference
    __device__ __forceinline__ bool project(int k, float& u, float& w) const {
        const float hyp = __fadd_rn(__fmul_rn((float)(k - half), incre), origin);
        const float X = fmaf(m3, hyp, a0), Y = fmaf(m7, hyp, a1), Z = fmaf(m11, hyp, a2);
        u = X / Z;
        w = Y / Z;
        const bool ok = (u == u) && (w == w);               // 0/0 samples nothing
        u = fminf(fmaxf(u, -1e4f), 1e4f);
        w = fminf(fmaxf(w, -1e4f), 1e4f);
        return ok;
    }
};
// the pixel of slot li (lanes l and l + 32 share it); false: the tile has no pixel at all (wave-uniform, and the same in every wave)
__device__ __forceinline__ bool cl_lane_pixel(const ClArgs& A, int v, int seg, int jj, int li, ClLane& L) {
    const int h1 = A.h1, w1 = A.w1;
    const int axis = (int)A.params[v * 4 + 0];
    const float shear = A.params[v * 4 + 1];
    const int La = axis ? h1 : w1, Hm = axis ? w1 : h1;     // extent along / across the tile axis
    const int njm = Hm + 32;
    if (jj >= njm) return false;
    const float cm = 0.5f * (float)La;
    const int a_first = seg * 32, a_last = min(seg * 32 + 31, La - 1);
    const int sh_f = (int)rintf(shear * ((float)a_first - cm)), sh_l = (int)rintf(shear * ((float)a_last - cm));
    const int sh_lo = min(sh_f, sh_l), sh_hi = max(sh_f, sh_l);
    const int j = jj - sh_hi;
    if (j > Hm - 1 - sh_lo) return false;
    const int a_me = seg * 32 + li;
    const int b_me = j + (int)rintf(shear * ((float)a_me - cm));
    L.valid = a_me < La && b_me >= 0 && b_me < Hm;
    if (__ballot(L.valid) == 0ull) return false;
    L.x_me = axis ? b_me : a_me;
    L.y_me = axis ? a_me : b_me;
    L.p_me = (long)min(max(L.y_me, 0), h1 - 1) * w1 + min(max(L.x_me, 0), w1 - 1);
    return true;
}
__device__ __forceinline__ void cl_lane_ray(const ClArgs& A, int v, ClLane& L) {
    const float* m = A.Pij + v * 16;
    const float px = (float)L.x_me, py = (float)(L.y_me + A.y0);
    float origin = A.disp_in[L.p_me];
    if (A.shift && origin < A.lim) origin = A.lim;
    L.origin = origin;
    L.a0 = fmaf(m[1], py, m[0] * px) + m[2];
    L.a1 = fmaf(m[5], py, m[4] * px) + m[6];
    L.a2 = fmaf(m[9], py, m[8] * px) + m[10];
    L.m3 = m[3]; L.m7 = m[7]; L.m11 = m[11];
    L.half = A.D / 2;
    L.incre = A.incre;
}

// (view, tile order inside the view) of the o-th tile of a launch that builds views v0 ..: groups of CL_LG lines x all segments, so that
// the ~100 tiles an XCD works on at a time cover a compact patch (16 lines x 6 segments: ~2 MB of band rows) instead of 100 lines of one
// segment (6 MB: its 4 MB L2 thrashed)
__device__ __forceinline__ void cl_decode(const ClArgs& A, unsigned o, int& v, int& rem, int& seg, int& jj) {
    v = A.v0 + (int)(o / (unsigned)A.tpv);
    rem = (int)(o % (unsigned)A.tpv);
    const int axis = (int)A.params[v * 4 + 0];
    const int La = axis ? A.h1 : A.w1;
    const int nseg = (La + 31) >> 5;
    const int lg = (int)((unsigned)rem / (unsigned)(nseg * CL_LG)), rem2 = rem - lg * nseg * CL_LG;
    seg = rem2 / CL_LG;
    jj = lg * CL_LG + (rem2 - seg * CL_LG);
}

// tile order inside its view of (segment, line) - the inverse of cl_decode (the hand-over list of the experimental multi-line form)
__device__ __forceinline__ int cl_rem(const ClArgs& A, int v, int seg, int jj) {
    const int axis = (int)A.params[v * 4 + 0];
    const int nseg = ((axis ? A.h1 : A.w1) + 31) >> 5;
    const int lg = jj / CL_LG;
    return lg * nseg * CL_LG + seg * CL_LG + (jj - lg * CL_LG);
}

// ---- band pass (round 6): one WAVE per tile derives the tile's band - major axis, reference line, rows across, column range, travel
// direction - from the end points of every pixel's segment: lanes with kg = 0 project hypothesis 0, lanes with kg = 1 hypothesis D - 1,
// and the halves swap; eight DPP wave reductions; one 48-byte record per tile.  43 680 tiles at the bench size: ~600 instructions per
// wave, no barrier, nothing waits for anybody.
__global__ __launch_bounds__(256) void cost_lines_bands_kernel(const ClArgs A, unsigned ntiles) {
    const unsigned o = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (o >= ntiles) return;
    const int lane = threadIdx.x & 63, li = lane & 31, kg = lane >> 5;
    int v, rem, seg, jj;
    cl_decode(A, o, v, rem, seg, jj);
    ClBand* rec = A.bands + (long)v * A.tpv + rem;
    ClLane L;
    if (!cl_lane_pixel(A, v, seg, jj, li, L)) {
        if (lane == 0) rec->nchunks = -1;
        return;
    }
    cl_lane_ray(A, v, L);
    const int h2 = A.h2, w2 = A.w2, D = A.D;
    const bool valid = L.valid;
    int smaj = 0, nchunks = 0, R = 1, Wc = 1, cmin = 0, cmax = 0, dir = 1;
    float bm = 0.f, bl0 = 0.f;
    float ua, wa, ub, wb;
    bool part0;
    {
        float ue, we;
        const bool oke = L.project(kg ? D - 1 : 0, ue, we);
        const float uo = __shfl_xor(ue, 32), wo = __shfl_xor(we, 32);
        const bool oko = ((__ballot(oke) >> (lane ^ 32)) & 1ull) != 0ull;
        ua = kg ? uo : ue; wa = kg ? wo : we;
        ub = kg ? ue : uo; wb = kg ? we : wo;
        part0 = valid && oke && oko;
    }
    const unsigned long long pm0 = __ballot(part0);
    if (pm0) {                                               // smaj 0: band runs along u (x of the source map), 1: along w
        const float INF = 3e38f;
        const float mnu = cl_wmin(part0 ? fminf(ua, ub) : INF), mxu = cl_wmax(part0 ? fmaxf(ua, ub) : -INF);
        const float mnw = cl_wmin(part0 ? fminf(wa, wb) : INF), mxw = cl_wmax(part0 ? fmaxf(wa, wb) : -INF);
        smaj = (mxw - mnw) > (mxu - mnu) ? 1 : 0;
    }
    smaj = __builtin_amdgcn_readfirstlane(smaj);
    const int Wmaj = smaj ? h2 : w2;
    if (pm0) {
        const float ma = smaj ? wa : ua, na = smaj ? ua : wa, mb = smaj ? wb : ub, nb = smaj ? ub : wb;
        // clip the segment to the columns of the (padded) map: what lies beyond samples zeros and needs no band
        const float lo = -2.0f, hi = (float)Wmaj + 1.0f;
        const float dm = mb - ma, dn = nb - na;
        float t0 = 0.f, t1 = 1.f;
        bool inside;
        if (fabsf(dm) < 1e-6f) {
            inside = ma >= lo && ma <= hi;
        } else {
            const float ta = (lo - ma) / dm, tb = (hi - ma) / dm;
            t0 = fmaxf(0.f, fminf(ta, tb));
            t1 = fminf(1.f, fmaxf(ta, tb));
            inside = t0 <= t1;
        }
        const bool part1 = part0 && inside;
        const unsigned long long pm1 = __ballot(part1);
        if (pm1) {
            const float INF = 3e38f;
            const float ma1 = fmaf(t0, dm, ma), na1 = fmaf(t0, dn, na), mb1 = fmaf(t1, dm, ma), nb1 = fmaf(t1, dn, na);
            const bool aFirst = ma1 <= mb1;
            const float lmin = part1 ? (aFirst ? ma1 : mb1) : INF, lminN = aFirst ? na1 : nb1;
            const float lmax = part1 ? (aFirst ? mb1 : ma1) : -INF, lmaxN = aFirst ? nb1 : na1;
            const float gmin = cl_wmin(lmin), gmax = cl_wmax(lmax);
            const int l0 = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(part1 && lmin == gmin)) - 1);
            const int l1 = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(part1 && lmax == gmax)) - 1);
            const float q0n = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lminN), l0));
            const float q1n = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lmaxN), l1));
            bm = (gmax - gmin > 1e-3f) ? (q1n - q0n) / (gmax - gmin) : 0.f;
            bm = fminf(fmaxf(bm, -4.f), 4.f);
            const float oa = na1 - fmaf(bm, ma1 - gmin, q0n), ob = nb1 - fmaf(bm, mb1 - gmin, q0n);
            const float omin = cl_wmin(part1 ? fminf(oa, ob) : INF), omax = cl_wmax(part1 ? fmaxf(oa, ob) : -INF);
            const float am = fabsf(bm);
            const float spread = omax - omin + 2.f * am + 0.02f;
            cmin = min(max((int)floorf(gmin), -2), Wmaj + 1);
            cmax = min(max((int)floorf(gmax) + 1, -2), Wmaj + 1);
            if (spread < (float)(CL_RMAX - 3) && cmax > cmin) {
                R = (int)floorf(spread) + 3;
                Wc = CL_T / R;
                bl0 = q0n - bm * gmin + omin - am - 0.01f;
                nchunks = (cmax - cmin + Wc - 2) / (Wc - 1);
                const unsigned long long fw = __ballot(part1 && mb1 > ma1), bw = __ballot(part1 && mb1 < ma1);
                dir = __popcll(fw) >= __popcll(bw) ? 1 : -1;
            }
        }
    }
    if (lane == 0) {
        ClBand r;
        r.smaj = smaj; r.nchunks = nchunks; r.R = R; r.Wc = Wc; r.cmin = cmin; r.cmax = cmax; r.dir = dir;
        r.bm = bm; r.bl0 = bl0;
        r.pad[0] = r.pad[1] = r.pad[2] = 0;
        *rec = r;
    }
}

// One (view, segment, line) tile by a 256-thread block; every early return is block-uniform.  `rem`: the tile's order inside its view
// (its band record is A.bands[v * tpv + rem]).
// F2LO (round 6): true = three-term dots  f1h*f2h + f1l*f2h + f1h*f2l  (fp32-class: rounds 3-5); false = "two-term": the band texels' lo planes
// are NOT READ - the source features enter as f16 (the reference rows keep both halves) - which halves the bytes of the fragment stream that IS this
// kernel's run time: stage 0 1 580 -> 1 020 us, 1.2e-5 relative L1 on the volume, 5e-6 on the final disparity (costed on the oracle first).
template <bool F2LO>
__device__ __forceinline__ void cl_tile(const ClArgs& A, const int v, const int rem, const int seg, const int jj) {
    __shared__ __attribute__((aligned(16))) float prod[CL_T * 32];      // dots[texel of the chunk][pixel of the tile]
    // per (hypothesis, pixel): {packed cell, fraction along the band, fraction across it, value}
    //   packed: bits 0-15 band column of the cell + 4; bits 30-31 kind: 0 = samples through the band (bits 16-20 / 21-25: band row
    //   of the cell in its own / the next column), 1 = zero (outside the map / non-finite), 2 = direct path (bits 16-29: cell row + 4)
    extern __shared__ __attribute__((aligned(16))) float desc[];          // [D][CL_DP][4]
    __shared__ int pidx[32];                                             // pixel index of tile slot i, or -1

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kg = lane >> 5;
    const int h1 = A.h1, w1 = A.w1, h2 = A.h2, w2 = A.w2, D = A.D;
    const unsigned long long cl_t0 = CL_CLK();
#if CL_STATS
    unsigned long long cl_acc[16] = {0};
#endif
    // ---- the tile's band (cost_lines_bands_kernel): a uniform address - scalar loads, requested before anything else
    const ClBand* rec = A.bands + (long)v * A.tpv + rem;
    int smaj = rec->smaj, nchunks = rec->nchunks, R = rec->R, Wc = rec->Wc, cmin = rec->cmin, cmax = rec->cmax, dir = rec->dir;
    float bm = rec->bm, bl0 = rec->bl0;
    ClLane L;
    if (!cl_lane_pixel(A, v, seg, jj, li, L)) return;       // (lanes l and l + 32 share pixel slot li: they own different hypotheses)
    const bool valid = L.valid;
    const long p_me = L.p_me;
    if (wave == 0 && lane < 32) pidx[lane] = valid ? (int)p_me : -1;

    // ---- first hop, all in one round trip: the pixel's origin (the projections below wait for it), the B fragments - the tile's 32
    // reference rows (lane: pixel slot li, channels 16 ks + 8 kg .. + 7), held for the whole tile - and the first chunk's A fragments
    cl_lane_ray(A, v, L);
    const long ps1 = (long)h1 * w1 * 16, ps2 = (long)(h2 + 4) * (w2 + 4) * 16;      // plane strides (halves) of the reference / source maps
    const _Float16* f1t = A.f1s + p_me * 16;
    half8 bh[4], bl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bh[ks] = *reinterpret_cast<const half8*>(f1t + ks * ps1 + 8 * kg);
        bl[ks] = *reinterpret_cast<const half8*>(f1t + (4 + ks) * ps1 + 8 * kg);
    }
    auto project = [&](int k, float& u, float& w) -> bool { return L.project(k, u, w); };

    const unsigned long long cl_t1 = CL_CLK();
    CL_STAT(0, 1); CL_STAT(3, cl_t1 - cl_t0);
    smaj = __builtin_amdgcn_readfirstlane(smaj);
    const int Wmaj = smaj ? h2 : w2, Wmin = smaj ? w2 : h2;
    const int wp = w2 + 4;
    const int smajS = smaj ? wp : 1, sminS = smaj ? 1 : wp;      // texel strides of the padded source map along / across the band
    nchunks = __builtin_amdgcn_readfirstlane(nchunks);
    R = __builtin_amdgcn_readfirstlane(R);
    Wc = __builtin_amdgcn_readfirstlane(Wc);
    cmin = __builtin_amdgcn_readfirstlane(cmin);
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    dir = __builtin_amdgcn_readfirstlane(dir);
    bm = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bm)));
    bl0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bl0)));

    const _Float16* f2v = A.f2s + (long)(A.slot ? A.slot[v] : v) * 8 * ps2;

    // ---- A fragments of a chunk: wave `wave` owns band texels 32 wave .. + 31 of the chunk (lane: texel li, channels as above)
    half8 ahf[4], alf[4];
    const int t_me = wave * 32 + li;
    const int colo = t_me / R, rowo = t_me - colo * R;
    const int cstep = dir * (Wc - 1), cb0 = dir > 0 ? cmin : cmax - (Wc - 1);      // chunk n covers band columns cb0 + n cstep .. + Wc - 1
    auto loadA = [&](int n) {
        const int col = cb0 + n * cstep + colo;
        const int row = (int)floorf(fmaf(bm, (float)col, bl0)) + rowo;
        const int cc = min(max(col, -2), Wmaj + 1), rc = min(max(row, -2), Wmin + 1);
        const _Float16* tp = f2v + (long)((cc + 2) * smajS + (rc + 2) * sminS) * 16 + 8 * kg;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            ahf[ks] = *reinterpret_cast<const half8*>(tp + ks * ps2);
            if (F2LO) alf[ks] = *reinterpret_cast<const half8*>(tp + (4 + ks) * ps2);
        }
    };
    if (nchunks > 0) loadA(0);                              // arrives under the projections below

    // ---- the samples of this lane: hypotheses k0, k0 + 8, ... (k0 = 2 wave + kg) of pixel slot li.  Straight-line code, two IEEE
    // divisions per sample, nothing divergent: cell, fractions and band rows go to LDS; the chunk loop only looks things up
    const int k0 = 2 * wave + kg;
#pragma unroll 2
    for (int k = k0; k < D; k += 8) {
        float u, w;
        const bool ok = project(k, u, w);
        const float fu = floorf(u), fw = floorf(w);
        const float du = ok ? u - fu : 0.f, dw = ok ? w - fw : 0.f;
        const int iu = ok ? min(max((int)fu, -2), w2) : -2, iw = ok ? min(max((int)fw, -2), h2) : -2;
        const int sc = smaj ? iw : iu, sr = smaj ? iu : iw;
        // cells without a corner inside the map read only border zeros (cost_build.hip clamps them into the zero border)
        const bool zero = iu < -1 || iu > w2 - 1 || iw < -1 || iw > h2 - 1;
        const int b0 = (int)floorf(fmaf(bm, (float)sc, bl0)), b1 = (int)floorf(fmaf(bm, (float)(sc + 1), bl0));
        const int r0 = sr - b0, r1 = sr - b1;
        const bool fits = nchunks > 0 && sc >= cmin && sc < cmax && r0 >= 0 && r0 + 1 < R && r1 >= 0 && r1 + 1 < R;
        const int hi14 = fits ? (r0 + r1 * 32) : (sr + 4);
        const int kind2 = zero ? 1 : (fits ? 0 : 2);
        const unsigned packed = (unsigned)(sc + 4) + ((unsigned)(hi14 & 0x3FFF) << 16) + ((unsigned)kind2 << 30);
        *reinterpret_cast<float4*>(desc + (k * CL_DP + li) * 4) = make_float4(__uint_as_float(packed), smaj ? dw : du, smaj ? du : dw, 0.f);
    }

    const unsigned long long cl_t2 = CL_CLK();
    CL_STAT(4, cl_t2 - cl_t1);
    // ---- per-lane sample cursor
    int k = valid ? k0 : D;
    unsigned pk = 3u << 30;                                 // kind 3: done
    float fm = 0.f, fn = 0.f;
    auto load_sample = [&]() {                              // (a lane only ever reads descriptors it wrote itself)
        if (k >= D) { pk = 3u << 30; return; }
        const float4 d = *reinterpret_cast<const float4*>(desc + (k * CL_DP + li) * 4);
        pk = __float_as_uint(d.x);
        fm = d.y;
        fn = d.z;
    };
    load_sample();
    auto direct_value = [&]() -> float {
        const int sc = (int)(pk & 0xFFFFu) - 4, sr = (int)((pk >> 16) & 0x3FFFu) - 4;
        const _Float16* t00 = f2v + (long)((sc + 2) * smajS + (sr + 2) * sminS) * 16;
        return cl_direct(f1t, ps1, t00, ps2, smajS, sminS, 1.0f - fm, fm, 1.0f - fn, fn, F2LO);
    };

    for (int n = 0; n < nchunks; ++n) {
        const int cb = cb0 + n * cstep;
        const unsigned long long cl_ta = CL_CLK();
        // ---- dots of this wave's 32 texels with the 32 pixels: 12 MFMAs
        floatx16 acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahf[ks], bh[ks], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahf[ks], bl[ks], acc0, 0, 0, 0);
            if (F2LO) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alf[ks], bh[ks], acc0, 0, 0, 0);
        }
        if (n + 1 < nchunks) loadA(n + 1);                  // in flight during the gather below
        // acc0[r]: texel row (r & 3) + 8 (r >> 2) + 4 kg of this wave's tile, pixel column li
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            prod[row * 32 + li] = acc0[r];
        }
        const unsigned long long cl_tb = CL_CLK();
        __syncthreads();
        const unsigned long long cl_tc = CL_CLK();
        // ---- gather: every lane consumes its samples whose cell lies in this chunk; zeros and direct-path samples as they come
        const int cbe = cb + Wc - 2, cbR = (cb - 4) * R;    // last cell column of the chunk; band index of (column c, row r) = c R + r - cb R
        for (;;) {
            const unsigned kind = pk >> 30;
            const int scp = (int)(pk & 0xFFFFu);            // cell column + 4
            const bool in = kind == 0 && scp >= cb + 4 && scp <= cbe + 4;
            const bool behind = kind == 0 && (dir > 0 ? scp < cb + 4 : scp > cbe + 4);
            const bool consume = in || behind || kind == 1 || kind == 2;
            if (__ballot(consume) == 0ull) break;
            float val = 0.f;
            if (in) {
                const int t0 = scp * R + (int)((pk >> 16) & 31u) - cbR - 8 * R;      // (scp - 4 - cb) R + r0  [cbR = (cb - 4) R]
                const int t1 = t0 + R + (int)((pk >> 21) & 31u) - (int)((pk >> 16) & 31u);
                const float* d0 = prod + t0 * 32 + li;
                const float* d1 = prod + t1 * 32 + li;
                const float wm1 = fm, wm0 = 1.0f - fm, wn1 = fn, wn0 = 1.0f - fn;
                val = d0[0] * (wn0 * wm0) + d1[0] * (wn0 * wm1) + d0[32] * (wn1 * wm0) + d1[32] * (wn1 * wm1);
            }
            if (__ballot(behind || kind == 2) != 0ull) {    // rare: the wave-level test keeps the call off the hot path
                if (behind || kind == 2) {
                    if (kind == 0) {                        // re-pack a band sample as a direct one: its cell row from the band row
                        const int sc = scp - 4;
                        const int sr = (int)floorf(fmaf(bm, (float)sc, bl0)) + (int)((pk >> 16) & 31u);
                        pk = (pk & 0xFFFFu) | ((unsigned)(sr + 4) << 16) | (2u << 30);
                    }
                    val = direct_value();
                }
            }
            if (consume) {
                desc[(k * CL_DP + li) * 4 + 3] = val;
                k += 8;
                load_sample();
            }
        }
        const unsigned long long cl_td = CL_CLK();
        __syncthreads();
        CL_STAT(1, 1); CL_STAT(5, cl_tb - cl_ta); CL_STAT(6, cl_tc - cl_tb); CL_STAT(7, cl_td - cl_tc); CL_STAT(8, CL_CLK() - cl_td);
    }
    const unsigned long long cl_t3 = CL_CLK();
    // ---- what the chunks did not cover (no band, samples out of order): direct path
    while (__ballot((pk >> 30) != 3u) != 0ull) {
        if ((pk >> 30) != 3u) {
            float val = 0.f;
            if ((pk >> 30) != 1u) {
                if ((pk >> 30) == 0u) {
                    const int sc = (int)(pk & 0xFFFFu) - 4;
                    const int sr = (int)floorf(fmaf(bm, (float)sc, bl0)) + (int)((pk >> 16) & 31u);
                    pk = (pk & 0xFFFFu) | ((unsigned)(sr + 4) << 16) | (2u << 30);
                }
                val = direct_value();
            }
            desc[(k * CL_DP + li) * 4 + 3] = val;
            k += 8;
            load_sample();
        }
    }
    __syncthreads();
    const unsigned long long cl_t4 = CL_CLK();
    CL_STAT(9, cl_t4 - cl_t3);
    // ---- rows out: wave w writes pixel slots w, w + 4, ...; lane = hypothesis (one coalesced D-float row per store)
    float* pv = A.part + (long)v * ((long)h1 * w1) * D;
    for (int i = wave; i < 32; i += 4) {
        const int p = pidx[i];
        if (p >= 0 && lane < D) pv[(long)p * D + lane] = desc[(lane * CL_DP + i) * 4 + 3];
    }
    CL_STAT(10, CL_CLK() - cl_t4); CL_STAT(2, CL_CLK() - cl_t0);
    CL_FLUSH();
}

// XCD-aware order (blocks are dealt round-robin over the 8 XCDs): each XCD gets a contiguous range of (view, segment, line):
// neighbouring lines share most of their band, which then stays in that XCD's L2
#ifndef CL_CU_ADJ
#define CL_CU_ADJ 0      // experiment (round 6): N > 0 - the k-th block of an XCD takes, inside groups of 3 N consecutive tiles, tile (k % N) * 3 + k / N:
                         // if the dispatcher deals an XCD's blocks round-robin over N CUs, the three resident blocks of a CU then hold ADJACENT lines
#endif
__device__ __forceinline__ unsigned cl_xcd_order() {
    const unsigned nblk = gridDim.x, bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    unsigned k = bid >> 3;
#if CL_CU_ADJ > 0
    {
        const unsigned len = xcd < r ? q + 1 : q, G = 3u * CL_CU_ADJ, g = k / G, j = k - g * G;
        if ((g + 1) * G <= len) k = g * G + (j % CL_CU_ADJ) * 3u + j / CL_CU_ADJ;
    }
#endif
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// OCC = waves per SIMD the register allocation aims at: 3 (LDS: 16 KiB of dots + D x 33 descriptors <= 50 KiB per block; 134 VGPRs).
// D <= 44 would fit 4 blocks per CU, but at 128 VGPRs the kernel spills and was measured slower (1.14 vs 1.04 ms)
template <int OCC, bool F2LO = true>
__global__ __launch_bounds__(256, OCC) void cost_lines_kernel(const ClArgs A) {
    int v, rem, seg, jj;
    cl_decode(A, cl_xcd_order(), v, rem, seg, jj);
    cl_tile<F2LO>(A, v, rem, seg, jj);
}

#ifndef CER_WITH_LINES8
#define CER_WITH_LINES8 0
#endif
#if CER_WITH_LINES8
#include "experimental/cost_lines8.inc"
#endif

// ---- sum of the per-view partials (view order: deterministic) * scale, origin, pooled levels: wave per pixel, lane = hypothesis
__global__ __launch_bounds__(256) void cost_lines_reduce_kernel(const float* __restrict__ part, const float* __restrict__ disp_in,
                                                               float* __restrict__ vol, float* __restrict__ origin_out, int V, long P, int D,
                                                               int rs, int levels, float scale, int accumulate, float lim, int shift) {
    const int lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    if (origin_out && lane == 0) {
        float origin = disp_in[p];
        if (shift && origin < lim) origin = lim;
        origin_out[p] = origin;
    }
    float s = 0.f;
    if (lane < D)
        for (int v = 0; v < V; ++v) s += part[((long)v * P + p) * D + lane];
    float* orow = vol + p * rs;
    float cur = s * scale;
    if (levels >= 1) {
        if (lane < D) orow[lane] = cur;
        int off = 0, n = D;
        for (int l = 1; l < levels; ++l) {                  // core/corr.py:94-97: (a + b) * 0.5 level by level; element j of level l on lane j << l
            const int mlen = n / 2;
            const float other = __shfl_xor(cur, 1 << (l - 1));
            cur = (cur + other) * 0.5f;
            off += n;
            if ((lane & ((1 << l) - 1)) == 0 && (lane >> l) < mlen) orow[off + (lane >> l)] = cur;
            n = mlen;
        }
    } else if (lane < D) {
        orow[lane] = accumulate ? (orow[lane] + cur) : cur;
    }
}

static long cl_tiles_per_view(int h1, int w1) {               // one-line tiles of a view (either axis), in CL_LG-line groups
    const long tx = (long)((w1 + 31) / 32) * ((h1 + 32 + CL_LG - 1) / CL_LG * CL_LG), ty = (long)((h1 + 31) / 32) * ((w1 + 32 + CL_LG - 1) / CL_LG * CL_LG);
    return tx > ty ? tx : ty;
}

extern "C" long cer_cost_lines_workspace(int V, int h1, int w1, int D) {
    if (V <= 0 || h1 <= 0 || w1 <= 0 || D <= 0) return CER_EINVAL;
    // per-view partial volumes | tile parameters | per view: a count + one entry per one-line tile (the lines the eight-line kernel hands over)
    return (long)V * h1 * w1 * D * 4 + (long)V * 16 + 256 + (long)V * (cl_tiles_per_view(h1, w1) + 1) * 8 + 64
           + (long)V * cl_tiles_per_view(h1, w1) * (long)sizeof(ClBand) + 64;      // (round 6: + one band record per tile)
}

#if CER_WITH_LINES8
// Which kernel builds the per-view partial volumes: 0 (default) the one-line form of round 3 (cost_lines_kernel), 1 the multi-line
// form of round 4 (cost_lines8_kernel: measured slower - DESIGN.md 3k - a variant-library experiment).  Process-wide like
// cer_cost_build_algo; < 0 queries.  CER_COST_LINES_FORM in the environment sets the initial value.
static std::atomic<int> g_lines_form{-1};
extern "C" int cer_cost_lines_form(int form) {
    int cur = g_lines_form.load();
    if (cur < 0) {
        const char* e = getenv("CER_COST_LINES_FORM");
        g_lines_form.compare_exchange_strong(cur, (e && e[0] == '1') ? 1 : 0);
    }
    const int prev = g_lines_form.load();
    if (form == 0 || form == 1) g_lines_form.store(form);
    return prev;
}
#endif

static int cl_check(int V, int h1, int w1, int h2, int w2, int C, int D) {
    if (V <= 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || D <= 0) return CER_EINVAL;
    if (C != 64 || D > 64 || V > 4096) return CER_ESHAPE;
    if ((long)(h2 + 4) * (w2 + 4) >= (1L << 24) || (long)h1 * w1 >= (1L << 24) || h2 > 16000 || w2 > 16000) return CER_ESHAPE;
    return CER_OK;
}
static float* cl_params(void* workspace, int V, long P, int D) {
    float* params = (float*)workspace + (long)V * P * D;
    return (float*)(((uintptr_t)params + 63) & ~(uintptr_t)63);
}

// views v0 .. v0 + nv - 1 of a V-view workspace: tile parameters + per-view partial volumes (no reduction)
extern "C" int cer_cost_lines_views_f32(const void* fmap1_split, const void* fmap2_split, const int* view_slot, const float* Pij,
                                        const float* disp_in, void* workspace, int V, int v0, int nv, int h1, int w1, int h2, int w2, int C, int D,
                                        double incre_d, int shift, int y0, int two_term, void* stream) {
    if (!fmap1_split || !fmap2_split || !Pij || !disp_in || !workspace) return CER_EINVAL;
    if (two_term != 0 && two_term != 1) return CER_EINVAL;
    const int rc = cl_check(V, h1, w1, h2, w2, C, D);
    if (rc != CER_OK) return rc;
    if (v0 < 0 || nv <= 0 || v0 + nv > V) return CER_EINVAL;
    if (!cer_aligned16(fmap1_split) || !cer_aligned16(fmap2_split) || !cer_aligned16(workspace)) return CER_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const long P = (long)h1 * w1;
    float* params = cl_params(workspace, V, P, D);
    hipLaunchKernelGGL(cost_lines_setup_kernel, dim3((unsigned)((nv + 63) / 64)), dim3(64), 0, st, Pij, params, v0, nv, h1, w1, y0);
    CER_RETURN_IF_LAUNCH_FAILED();
    ClArgs a;
    a.f1s = (const _Float16*)fmap1_split;
    a.f2s = (const _Float16*)fmap2_split;
    a.Pij = Pij;
    a.params = params;
    a.disp_in = disp_in;
    a.part = (float*)workspace;
    a.slot = view_slot;
    a.V = V; a.v0 = v0; a.h1 = h1; a.w1 = w1; a.h2 = h2; a.w2 = w2; a.D = D;
    a.incre = (float)incre_d;
    a.lim = (float)((D / 2) * incre_d);
    a.shift = shift;
    a.y0 = y0;
    a.tpv = (int)cl_tiles_per_view(h1, w1);
    const long nblk = (long)nv * a.tpv;
    if (nblk >= (1L << 31)) return CER_ESHAPE;
    const size_t dyn = (size_t)D * CL_DP * 16;
    {
        uintptr_t t = (uintptr_t)(params + (long)V * 4);
        t = (t + 63) & ~(uintptr_t)63;
        a.todo = (unsigned long long*)t + (long)v0 * (a.tpv + 1);     // this call's views own this stretch of the list
    }
    {   // band records behind the list, [V][tpv]: indexed by the ABSOLUTE view, so launches for different view ranges never collide
        uintptr_t t = (uintptr_t)((unsigned long long*)(((uintptr_t)(params + (long)V * 4) + 63) & ~(uintptr_t)63) + (long)V * (a.tpv + 1));
        a.bands = (ClBand*)((t + 63) & ~(uintptr_t)63);
    }
    a.tpv8 = 0;
    hipLaunchKernelGGL(cost_lines_bands_kernel, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, st, a, (unsigned)nblk);
    CER_RETURN_IF_LAUNCH_FAILED();
#if CER_WITH_LINES8
    if (cer_cost_lines_form(-1) == 1) {
        // NW lines x one segment per block; lines whose bands do not fit its window are listed and done in the one-line form
        static int nw = 0, outd = 1;                         // lines per block: 4 (measured best) or 8 (CER_COST_LINES_NW=8); CER_COST_LINES_OUTD=0: values staged in LDS
        if (!nw) {
            const char* e = getenv("CER_COST_LINES_NW");
            nw = (e && e[0] == '8') ? 8 : 4;
            e = getenv("CER_COST_LINES_OUTD");
            outd = (e && e[0] == '0') ? 0 : 1;
        }
        const int bpg = 16 / nw;
        const long tx8 = (long)((w1 + 31) / 32) * bpg * ((h1 + 32 + 15) / 16), ty8 = (long)((h1 + 31) / 32) * bpg * ((w1 + 32 + 15) / 16);
        a.tpv8 = (int)(tx8 > ty8 ? tx8 : ty8);
        int dev = -1;
        static bool raised[4][64];
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        const int kid = (nw == 8 ? 2 : 0) + outd;
        void (*fn)(const ClArgs) = kid == 0 ? cost_lines8_kernel<4, false> : kid == 1 ? cost_lines8_kernel<4, true>
                                   : kid == 2 ? cost_lines8_kernel<8, false> : cost_lines8_kernel<8, true>;
        const int smem = kid == 0 ? C8Cfg<4, false>::SMEM : kid == 1 ? C8Cfg<4, true>::SMEM : kid == 2 ? C8Cfg<8, false>::SMEM : C8Cfg<8, true>::SMEM;
        if (dev < 0 || dev >= 64 || !raised[kid][dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) raised[kid][dev] = true;
        }
        hipError_t e = hipMemsetAsync(a.todo, 0, 8, st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(fn, dim3((unsigned)((long)nv * a.tpv8)), dim3(64 * nw), smem, st, a);
        CER_RETURN_IF_LAUNCH_FAILED();
        hipLaunchKernelGGL(cost_lines_todo_kernel, dim3(768), dim3(256), dyn, st, a);
        CER_RETURN_IF_LAUNCH_FAILED();
        return CER_OK;
    }
#endif
    if (two_term) hipLaunchKernelGGL((cost_lines_kernel<3, false>), dim3((unsigned)nblk), dim3(256), dyn, st, a);
    else hipLaunchKernelGGL(cost_lines_kernel<3>, dim3((unsigned)nblk), dim3(256), dyn, st, a);      // (<4> at D <= 44: 128 VGPRs with 16 spilled dwords - 895 against 795 us, re-measured in round 5)
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// sum of the V partial volumes of a workspace (view order) -> vol rows (+ origin, pooled levels): the second half of cer_cost_lines_f32
extern "C" int cer_cost_lines_reduce_f32(const void* workspace, const float* disp_in, float* vol, float* origin_out, int V, int h1, int w1, int D,
                                         int row_stride, double incre_d, int shift, int mode, int fuse_levels, float fuse_scale, void* stream) {
    if (!workspace || !disp_in || !vol) return CER_EINVAL;
    if (V <= 0 || h1 <= 0 || w1 <= 0 || D <= 0 || D > 64 || row_stride < D || (mode != 1 && mode != 2)) return CER_EINVAL;
    if (fuse_levels >= 1) {                                 // (1: level 0 only, scaled - the compact rows of round 5; the lookup pools on the fly)
        if (mode != 1) return CER_EINVAL;
        int need = 0, n = D;
        for (int l = 0; l < fuse_levels; ++l) { need += n; n /= 2; }
        if (row_stride < need || fuse_levels > 6) return CER_ESHAPE;
    }
    const long P = (long)h1 * w1;
    const float scale = (fuse_levels >= 1 ? fuse_scale : 1.0f) / (float)(1 << (2 * CL_LOG2S));
    hipLaunchKernelGGL(cost_lines_reduce_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, disp_in, vol,
                       origin_out, V, P, D, row_stride, fuse_levels, scale, mode == 2 ? 1 : 0, (float)((D / 2) * incre_d), shift);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_cost_lines_f32(const void* fmap1_split, const void* fmap2_split, const int* view_slot, const float* Pij, const float* disp_in,
                                  float* vol, float* origin_out, void* workspace, int V, int h1, int w1, int h2, int w2, int C, int D, int row_stride,
                                  double incre_d, int shift, int mode, int y0, int fuse_levels, float fuse_scale, int two_term, void* stream) {
    if (!vol || row_stride < D || (mode != 1 && mode != 2)) return CER_EINVAL;
    int rc = cer_cost_lines_views_f32(fmap1_split, fmap2_split, view_slot, Pij, disp_in, workspace, V, 0, V, h1, w1, h2, w2, C, D, incre_d, shift, y0,
                                      two_term, stream);
    if (rc != CER_OK) return rc;
    return cer_cost_lines_reduce_f32(workspace, disp_in, vol, origin_out, V, h1, w1, D, row_stride, incre_d, shift, mode, fuse_levels, fuse_scale,
                                     stream);
}
