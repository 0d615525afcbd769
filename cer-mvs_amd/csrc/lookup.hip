// K2: multi-level correlation lookup (reference: CorrBlock.__call__ core/corr.py:102-143 and
// bilinear_sampler1 utils/bilinear_sampler.py:6-25 - 528 grid_sample launches per GRU iteration in
// the reference), optionally fused with the view mean (core/update.py:103) and the first
// corr_encoder layer (core/update.py:61-62: Conv2d(33,64,1) + ReLU).
//
// HBM-bound: per pixel one volume row (level0|level1|level2, e.g. 112 floats) is read and
// L*(2r+1) = 33 floats (or 64 encoded floats) are written.  A block stages 64 consecutive rows
// into LDS with coalesced 16-B loads; each thread then owns (pixel, level-slice) windows.
#include "common.hpp"

#define LK_PIX 64          // pixels per block
#define LK_MAX_ROW 256     // max row_stride (floats)
#define LK_MAX_TAPS 64     // max L*(2r+1)
#ifndef LK_ABL
#define LK_ABL 0           // timing ablations (variant builds, results wrong): 1 no 1x1 conv loop, 2 no windows, 4 no row loads, 8 no output stores
#endif
#ifndef LK_OCC4
#define LK_OCC4 5          // blocks per CU the 4-prefetch-register instantiation (level-0-only rows) is compiled for
#endif

struct LevelInfo {
    int off[8];
    int len[8];
    int pool;              // 1: the rows hold level 0 only; level l is formed on the fly (round 5, see lk_elem)
};

// LDS pitch of a staged row (floats): rs + 4 or + 8 so that pitch / 4 is odd - the 16-byte accesses of 8 consecutive lanes (one row each,
// same column) then fall into 8 different 16-byte bank groups
__host__ __device__ __forceinline__ int lk_pitch(int rs) { return rs + (((rs >> 2) & 1) ? 8 : 4); }

// Element i of pyramid level lv formed from a level-0 row (core/corr.py:94-97: F.avg_pool2d([1,2]) level by level = (a + b) * 0.5, in the
// association the build's fused epilogue / cer_pyramid_f32 use, so the value is BIT-IDENTICAL to the stored level).  The row is 16-byte
// aligned and 2^lv * i is a multiple of 2^lv: one 8- / 16-byte LDS read per element.  Since round 5 the folded volume of RAFT.forward keeps
// level 0 only: rows of D instead of 1.75 D floats (-43 % of the lookup's read bytes and of the build's volume write).
__device__ __forceinline__ float lk_elem(const float* __restrict__ row, int lv, int i) {
    if (lv == 0) return row[i];
    if (lv == 1) {
        const float2 v = *reinterpret_cast<const float2*>(row + 2 * i);
        return (v.x + v.y) * 0.5f;
    }
    if (lv == 2) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * i);
        return ((v.x + v.y) * 0.5f + (v.z + v.w) * 0.5f) * 0.5f;
    }
    const float4 u = *reinterpret_cast<const float4*>(row + 8 * i), v = *reinterpret_cast<const float4*>(row + 8 * i + 4);      // lv == 3 (at most 4 levels)
    return (((u.x + u.y) * 0.5f + (u.z + u.w) * 0.5f) * 0.5f + ((v.x + v.y) * 0.5f + (v.z + v.w) * 0.5f) * 0.5f) * 0.5f;
}

__device__ __forceinline__ float lk_index(float disp, float origin, float incre, int D) {
    // core/corr.py:107 - true division then + D//2, lower clamp only
    const float c = __fadd_rn(__fdiv_rn(__fsub_rn(disp, origin), incre), (float)(D / 2));
    return fmaxf(c, 0.0f);    // NaN -> 0 like torch.maximum? (torch.maximum propagates NaN; disp is never NaN on this path)
}

// one window of 2r+1 taps on level `lv` of the LDS row
// (pool: the row holds level 0 only and `lv` says which level to form; else `off` is the level's offset in the row)
__device__ __forceinline__ void lk_window(const float* __restrict__ row, int off, int len, float x, int r, float* __restrict__ o, int pool = 0, int lv = 0) {
    // x = c / 2^lv (exact); taps at x + dx, dx = -r..r; zero outside [0, len-1] (grid_sample zeros padding, align_corners)
    const float fx = floorf(x);
    const float w = x - fx;
    const bool in_range = fx < (float)(len + r + 1);         // else every tap is outside
    const int i0 = in_range ? (int)fx - r : 0;
    float prev = 0.f;
    {
        const int i = i0;
        prev = (in_range && i >= 0 && i < len) ? (pool ? lk_elem(row, lv, i) : row[off + i]) : 0.f;
    }
    for (int j = 0; j < 2 * r + 1; ++j) {
        const int i = i0 + j + 1;
        const float next = (in_range && i >= 0 && i < len) ? (pool ? lk_elem(row, lv, i) : row[off + i]) : 0.f;
        o[j] = prev * (1.0f - w) + next * w;
        prev = next;
    }
}

// out [nv, L*(2r+1), P] planar
__global__ __launch_bounds__(256) void lookup_kernel(const float* __restrict__ vol, const float* __restrict__ origin,
                                                     const float* __restrict__ disp, long dvs, float* __restrict__ out, long P, int D,
                                                     int rs, float incre, int L, int r, LevelInfo li) {
    extern __shared__ __attribute__((aligned(16))) float lk_smem[];
    float* rows = lk_smem;                                   // [LK_PIX][rs + 4]
    const int v = blockIdx.y;
    const long p0 = (long)blockIdx.x * LK_PIX;
    const int npix = (int)min((long)LK_PIX, P - p0);
    const int rsp = lk_pitch(rs);                            // padded LDS stride
    // stage rows: rs/4 float4 per row
    const float* src = vol + ((long)v * P + p0) * rs;
    const int n4 = rs / 4;
    for (int t = threadIdx.x; t < npix * n4; t += 256) {
        const int pr = t / n4, q = t - pr * n4;
        const float4 val = cer_ld4(src + (long)pr * rs + 4 * q);
        *reinterpret_cast<float4*>(&rows[pr * rsp + 4 * q]) = val;
    }
    __syncthreads();
    const int taps = 2 * r + 1;
    // thread -> (pixel = tid & 63, level = tid >> 6 ...) : levels strided over the 4 waves
    const int pix = threadIdx.x & 63;
    if (pix >= npix) return;
    const long p = p0 + pix;
    const float c = lk_index(disp[(long)v * dvs + p], origin[p], incre, D);
    for (int lv = threadIdx.x >> 6; lv < L; lv += 4) {
        float o[32];
        const float x = c / (float)(1 << lv);
        lk_window(&rows[pix * rsp], li.off[lv], li.len[lv], x, r, o, li.pool, lv);
        float* dst = out + ((long)v * L * taps + (long)lv * taps) * P + p;
        for (int j = 0; j < taps; ++j) dst[(long)j * P] = o[j];
    }
}

// fused: lookup on the folded volume + 1x1 conv (taps_total -> 64) + bias + ReLU, out [P,64] NHWC (or split32 / frag16).
// HBM-bound (one 4*rs-byte volume row in, 256 bytes out per pixel), so the kernel is built around keeping loads in flight:
// persistent blocks (3 per CU) walk over 64-pixel tiles; the NEXT tile's rows are requested into registers (16-byte loads, <= 16
// per thread) right after the current tile's have been written to LDS, and arrive while the block does the windows and the 1x1
// conv of the current tile.  Two barriers per tile: rows -> [B1] -> windows (one level per wave) into a double-buffered feature
// tile -> [B2] -> conv (1x1 weights through the scalar cache: wave-uniform).
// (A variant with the 3 x 11 windows and the 33-deep conv unrolled at compile time was 1 us faster stand-alone, no faster in the
// pipeline, and produced intermittently wrong features when a second process shared the GPU - 17 of 70 two-process runs against
// 0 of 130 for this form; the cause was narrowed to the unrolled window code but not found.  It was removed.)
// Happens-before: `rows` is written in front of [B1](t) and read (windows) between [B1](t) and [B2](t);
// the next tile's writes follow [B2](t).  `feats` (ONE tile since round 4: the second half bought nothing) is written between [B1](t) and
// [B2](t) and read by the conv behind [B2](t); it is written again behind [B1](t+1), which a wave passes only after EVERY wave has arrived
// there - i.e. has finished its conv of tile t (the conv precedes the next tile's row stores and [B1](t+1) in every wave's program order).
// The schedule-fuzz build (common.hpp CER_FUZZ) reproduces the first launch over 2 000 launches.
#define LK_MAX_PRE 16      // float4 per thread of one 64-row tile: 64 * (LK_MAX_ROW / 4) / 256
// Round 5: the PREVIOUS iteration's disparity update (core/update.py:114, core/raft.py:101: disp += 0.01 * delta, the 18-tap gather of
// cer_delta_sum_f32) can ride on this launch: with `dT` given, wave 3 - idle while waves 0-2 form the three levels' windows - requests the
// tap planes of a tile's pixels together with the tile's rows, forms  d' = d + 0.01 * (bias + sum of the taps)  in cer_delta_sum_f32's
// order (bit-identical), writes it back to `disp` (in place: every pixel is read and written by exactly one thread of one block) and
// publishes it to the block through LDS (`dnew`: written in front of [B1](t), read behind it, written again behind [B2](t)).
struct LkDelta {
    const float* T;        // [nhalf][9][P] tap planes of the fused delta head, or null: `disp` is used as it is
    float* disp_rw;        // the disparity, updated in place
    int nhalf, h;
    float bias;
    int* flag;             // sticky overflow flag of the device (cer_overflow_flag) or null: bit 4 = a frag16 output value had to be clamped
};
// MAXPRE: float4 registers of the next tile's rows per thread (4: level-0-only rows up to 64 floats - the model's since round 5; 8: rows up to
// 128 floats - the stored pyramid's 112; 16: any row the entry point accepts)
template <int MAXPRE>
__global__ __launch_bounds__(256, MAXPRE <= 4 ? LK_OCC4 : MAXPRE <= 8 ? 4 : 3) void lookup_encode_kernel(const float* __restrict__ vol, const float* __restrict__ origin,
                                                            const float* __restrict__ disp, const float* __restrict__ wgt,
                                                            const float* __restrict__ bias, float* __restrict__ out, long P, int D, int rs,
                                                            float incre, int L, int r, LevelInfo li, int out_split, float out_scale, int img_w,
                                                            int ntiles, LkDelta dl) {
    extern __shared__ __attribute__((aligned(16))) float lk_smem[];
    const int rsp = lk_pitch(rs);
    const int taps = 2 * r + 1, K = L * taps, FS = K | 1;                         // (odd pixel stride: conflict-free columns)
    float* rows = lk_smem;                                   // [LK_PIX][rsp]
    float* feats = lk_smem + LK_PIX * rsp;                   // [LK_PIX][FS]  (one tile: see the happens-before note above)
    float* dnew = feats + LK_PIX * FS;                       // [LK_PIX]  updated disparities of the tile (dl.T given)
    const int n4 = rs / 4, npre = (LK_PIX * n4 + 255) / 256;
    const int pix = threadIdx.x & 63;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave id: lookup level, then output-channel group
    float4 pre[MAXPRE];
    float pre_d = 0.f, pre_o = 0.f;
    float tp[18];                                            // wave 3: the 2 x 9 taps of its pixel (zeros outside the image)
    const bool fuse_delta = dl.T != nullptr;                 // (uniform)
    CER_FUZZ_INIT();
    // float4 number t = tid + 256 i of the tile is (row pr, quad q) = (t / n4, t % n4): walked incrementally (no division per item)
    const int pr0 = threadIdx.x / n4, q0 = threadIdx.x - pr0 * n4, dpr = 256 / n4, dq = 256 - dpr * n4;
    auto request = [&](int tile) {                           // rows of `tile` -> registers (zeros past the last pixel)
        const long p0 = (long)tile * LK_PIX;
        const int npix = (int)min((long)LK_PIX, P - p0);
        const float* src = vol + p0 * rs;
        if (pix < npix) {                                    // ... and this thread's pixel's window position
            pre_o = origin[p0 + pix];
            if (!fuse_delta) {
                pre_d = disp[p0 + pix];
            } else if (grp == 3) {                           // (the other waves take the updated value from `dnew`)
                pre_d = disp[p0 + pix];
                const unsigned p = (unsigned)(p0 + pix);
                const int y = (int)(p / (unsigned)img_w), x = (int)(p - (unsigned)y * (unsigned)img_w);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                        const bool ok = hf < dl.nhalf && yy >= 0 && yy < dl.h && xx >= 0 && xx < img_w;
                        tp[hf * 9 + tap] = ok ? dl.T[((long)hf * 9 + tap) * P + (long)yy * img_w + xx] : 0.f;
                    }
            }
        }
        int pr = pr0, q = q0;
#pragma unroll
        for (int i = 0; i < MAXPRE; ++i) {                // (predicated, not `break`: pre[] must stay in registers)
            if (i < npre) {
                pre[i] = (pr < npix && !(LK_ABL & 4)) ? cer_ld4(src + (long)(threadIdx.x + 256 * i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                pr += dpr; q += dq;
                if (q >= n4) { q -= n4; ++pr; }
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) request(tile);
    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        const long p0 = (long)tile * LK_PIX;
        const int npix = (int)min((long)LK_PIX, P - p0);
        const bool active = pix < npix;
        float c = (active && !fuse_delta) ? lk_index(pre_d, pre_o, incre, D) : 0.f;
        const float po = pre_o;                              // (request() below overwrites pre_o with the next tile's)
        CER_FUZZ_POINT();                                    // (in front of the LDS write phase)
        if (fuse_delta && grp == 3 && active) {
            float s = 0.f;                                   // cer_delta_sum_f32's order: half-major, taps in order (a tap outside the image adds + 0)
#pragma unroll
            for (int i = 0; i < 18; ++i) s += tp[i];
            const float dn = pre_d + 0.01f * (s + dl.bias);
            dl.disp_rw[p0 + pix] = dn;
            dnew[pix] = dn;
        }
        {
            int pr = pr0, q = q0;
#pragma unroll
            for (int i = 0; i < MAXPRE; ++i) {
                if (i < npre) {
                    if (pr < LK_PIX) *reinterpret_cast<float4*>(&rows[pr * rsp + 4 * q]) = pre[i];
                    pr += dpr; q += dq;
                    if (q >= n4) { q -= n4; ++pr; }
                }
            }
        }
        __syncthreads();                                     // [B1] rows complete; the previous tile's windows were read before its [B2]
        CER_FUZZ_POINT();
        if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
        float* ft = feats;
        if (fuse_delta && active) c = lk_index(dnew[pix], po, incre, D);
        if (active && !(LK_ABL & 2))
            for (int lv = grp; lv < L; lv += 4)            // (straight into the feature tile: no per-thread array)
                lk_window(&rows[pix * rsp], li.off[lv], li.len[lv], c / (float)(1 << lv), r, &ft[pix * FS + lv * taps], li.pool, lv);
        __syncthreads();                                     // [B2] features complete; rows free for the next tile
        CER_FUZZ_POINT();
        if (!active) continue;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias[grp * 16 + j];
#pragma unroll 3
        for (int k = 0; k < ((LK_ABL & 1) ? 1 : K); ++k) {
            const float f = ft[pix * FS + k];
            const float* wr = wgt + k * 64 + grp * 16;       // wave-uniform: scalar loads
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(f, wr[j], acc[j]);
        }
        if (LK_ABL & 8) { if (acc[0] == 123.456f) out[p0 + pix] = acc[1]; continue; }
        if (out_split == 2) {
            // frag16 layout (cer_mvs.h, conv_s16.hip): this thread's 16 channels are group `grp` of its pixel's m-tile: four 16-byte
            // pieces (hi | lo planes x channel octets) of relu(acc) * out_scale
            const unsigned p = (unsigned)(p0 + pix);             // (P < 2^31 is checked by the launcher: 32-bit division)
            const unsigned y = p / (unsigned)img_w, x = p - y * (unsigned)img_w;
            const long mt = (long)(y >> 1) * ((img_w + 15) >> 4) + (x >> 4);
            char* dst = reinterpret_cast<char*>(out) + ((mt * 4 + grp) * 2) * 1024 + (((y & 1) << 4) | (x & 15)) * 16;
            // saturation is never silent: the largest value of this thread's 16 outputs is checked against the clamp here -
            // round 5: the tensor's buffer is reused for r * h later in the iteration, so nothing can scan it after the fact
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) amax = fmaxf(amax, acc[j]);
            if (dl.flag && __ballot(!(amax * out_scale <= 65504.0f)) != 0ull && (threadIdx.x & 63) == 0) atomicOr(dl.flag, 4);
#pragma unroll
            for (int j = 0; j < 16; j += 8) {
                cer_h2 h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                // packed: relu, scale, clamp, hi = f16(xs), lo = f16(xs - hi)
                    cer_f2 xs = __builtin_elementwise_max((cer_f2){acc[j + 2 * e], acc[j + 2 * e + 1]}, (cer_f2){0.f, 0.f}) * out_scale;
                    xs = __builtin_elementwise_min(xs, (cer_f2){65504.0f, 65504.0f});
                    h[e] = __builtin_convertvector(xs, cer_h2);
                    l[e] = __builtin_convertvector(xs - __builtin_convertvector(h[e], cer_f2), cer_h2);
                }
                *reinterpret_cast<cer_h8*>(dst + (j >> 3) * 512) = (cer_h8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
                *reinterpret_cast<cer_h8*>(dst + 1024 + (j >> 3) * 512) = (cer_h8){l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
            }
        } else if (out_split) {
            // split32 layout (cer_mvs.h): per pixel and 32-channel chunk 32 hi halves | 32 lo halves - the corr2 conv then stages
            // this tensor with plain copies
            char* dst = reinterpret_cast<char*>(out + (p0 + pix) * 64) + (grp >> 1) * 128 + (grp & 1) * 32;
#pragma unroll
            for (int j = 0; j < 16; j += 8) {
                const float v[8] = {fmaxf(acc[j], 0.f), fmaxf(acc[j + 1], 0.f), fmaxf(acc[j + 2], 0.f), fmaxf(acc[j + 3], 0.f),
                                    fmaxf(acc[j + 4], 0.f), fmaxf(acc[j + 5], 0.f), fmaxf(acc[j + 6], 0.f), fmaxf(acc[j + 7], 0.f)};
                cer_h8 hi, lo;
                cer_split8(v, hi, lo);
                *reinterpret_cast<cer_h8*>(dst + j * 2) = hi;
                *reinterpret_cast<cer_h8*>(dst + 64 + j * 2) = lo;
            }
        } else {
            float* dst = out + (p0 + pix) * 64 + grp * 16;
#pragma unroll
            for (int j = 0; j < 16; j += 4)
                *reinterpret_cast<float4*>(dst + j) =
                    make_float4(fmaxf(acc[j], 0.f), fmaxf(acc[j + 1], 0.f), fmaxf(acc[j + 2], 0.f), fmaxf(acc[j + 3], 0.f));
        }
    }
}

// level0_only (round 6, ADVICE r5: explicit, no longer inferred from the stride): the rows hold level 0 only and the pooled levels are formed
// on the fly (lk_elem: at most 4 levels); otherwise the rows hold the whole pyramid [level0 | level1 | ...] and must be long enough for it
static int level_info(int D, int rs, int L, int r, int level0_only, LevelInfo* li) {
    if (L <= 0 || L > 8 || r < 0 || r > 15 || L * (2 * r + 1) > LK_MAX_TAPS) return CER_ESHAPE;
    if (rs % 4 != 0 || rs > LK_MAX_ROW || rs < D) return CER_ESHAPE;
    if (level0_only != 0 && level0_only != 1) return CER_EINVAL;
    int off = 0, n = D;
    for (int l = 0; l < L; ++l) {
        li->off[l] = off;
        li->len[l] = n;
        off += n;
        n /= 2;
    }
    li->pool = level0_only;
    if (level0_only ? L > 4 : off > rs) return CER_ESHAPE;
    return CER_OK;
}

extern "C" int cer_corr_lookup_f32(const float* vol, const float* origin, const float* disp, long disp_view_stride, float* out, int nv,
                                   long P, int D, int row_stride, double incre, int num_levels, int radius, int level0_only, void* stream) {
    if (!vol || !origin || !disp || !out || nv <= 0 || P <= 0 || D <= 0) return CER_EINVAL;
    if (!cer_aligned16(vol)) return CER_EALIGN;
    LevelInfo li;
    int rc = level_info(D, row_stride, num_levels, radius, level0_only, &li);
    if (rc) return rc;
    hipLaunchKernelGGL(lookup_kernel, dim3((unsigned)((P + LK_PIX - 1) / LK_PIX), (unsigned)nv), dim3(256),
                       sizeof(float) * LK_PIX * lk_pitch(row_stride), (hipStream_t)stream, vol, origin,
                       disp, disp_view_stride, out, P, D, row_stride, (float)incre, num_levels, radius, li);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_lookup_encode_f32(const float* vol, const float* origin, float* disp, const float* w, const float* b, float* out,
                                     long P, int D, int row_stride, double incre, int num_levels, int radius, int Cout, int out_split,
                                     int log2s_out, int img_w, const float* delta_taps, int delta_nhalf, float delta_bias, int level0_only, void* stream) {
    if (!vol || !origin || !disp || !w || !b || !out || P <= 0 || D <= 0) return CER_EINVAL;
    if (Cout != 64 || num_levels > 4) return CER_ESHAPE;
    if ((out_split == 2 || delta_taps) && (img_w <= 0 || P % img_w != 0 || P >= (1L << 31))) return CER_ESHAPE;
    if (delta_taps && (delta_nhalf < 1 || delta_nhalf > 2)) return CER_ESHAPE;
    if (!cer_aligned16(vol) || !cer_aligned16(out)) return CER_EALIGN;
    LevelInfo li;
    int rc = level_info(D, row_stride, num_levels, radius, level0_only, &li);
    if (rc) return rc;
    const int K = num_levels * (2 * radius + 1);
    const long ntiles = (P + LK_PIX - 1) / LK_PIX;
    if (ntiles >= (1L << 30)) return CER_ESHAPE;
    const size_t smem = sizeof(float) * LK_PIX * ((size_t)lk_pitch(row_stride) + (K | 1) + 1);
    const int ncu = cer_num_cus();
    LkDelta dl;
    dl.T = delta_taps; dl.disp_rw = disp; dl.nhalf = delta_nhalf; dl.h = delta_taps ? (int)(P / img_w) : 0; dl.bias = delta_bias;
    dl.flag = out_split == 2 ? cer_overflow_flag_get() : nullptr;
    const int pre = row_stride <= 64 ? 4 : row_stride <= 128 ? 8 : LK_MAX_PRE;      // prefetch registers per thread: which instantiation
    const int by_regs = pre == 4 ? LK_OCC4 : pre == 8 ? 4 : 3;
    const int by_lds = (int)((long)cer_lds_per_cu() / (long)(smem + 256));
    const long resident = (long)ncu * (by_lds < 1 ? 1 : by_lds < by_regs ? by_lds : by_regs);
    const unsigned grid = (unsigned)(ntiles < resident ? ntiles : resident);
#define LK_LAUNCH(N) hipLaunchKernelGGL(lookup_encode_kernel<N>, dim3(grid), dim3(256), smem, (hipStream_t)stream, vol, origin, disp, w, b, out, P, D, row_stride, \
                                        (float)incre, num_levels, radius, li, out_split, ldexpf(1.0f, log2s_out), img_w, (int)ntiles, dl)
    if (pre == 4) LK_LAUNCH(4);
    else if (pre == 8) LK_LAUNCH(8);
    else LK_LAUNCH(LK_MAX_PRE);
#undef LK_LAUNCH
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// mean over views of planar features + 1x1 conv (K -> 64) + bias + ReLU -> NHWC [P,64]
__global__ __launch_bounds__(256) void corr_encode_kernel(const float* __restrict__ feats, const float* __restrict__ wgt,
                                                          const float* __restrict__ bias, float* __restrict__ out, int nv, int K, long P) {
    extern __shared__ __attribute__((aligned(16))) float lk_smem[];
    float* wsm = lk_smem;                        // [K][64]
    float* fm = wsm + K * 64;                    // [K][LK_PIX]
    const long p0 = (long)blockIdx.x * LK_PIX;
    const int npix = (int)min((long)LK_PIX, P - p0);
    for (int t = threadIdx.x; t < K * 64; t += 256) wsm[t] = wgt[t];
    const float inv = 1.0f / (float)nv;
    for (int t = threadIdx.x; t < K * LK_PIX; t += 256) {
        const int k = t / LK_PIX, pix = t - k * LK_PIX;
        float s = 0.f;
        if (pix < npix)
            for (int v = 0; v < nv; ++v) s += feats[((long)v * K + k) * P + p0 + pix];
        fm[t] = nv == 1 ? s : s * inv;
    }
    __syncthreads();
    const int pix = threadIdx.x & 63, grp = threadIdx.x >> 6;
    if (pix >= npix) return;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[grp * 16 + j];
    for (int k = 0; k < K; ++k) {
        const float f = fm[k * LK_PIX + pix];
        const float* wr = &wsm[k * 64 + grp * 16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = fmaf(f, wr[j], acc[j]);
    }
    float* dst = out + (p0 + pix) * 64 + grp * 16;
#pragma unroll
    for (int j = 0; j < 16; j += 4)
        *reinterpret_cast<float4*>(dst + j) =
            make_float4(fmaxf(acc[j], 0.f), fmaxf(acc[j + 1], 0.f), fmaxf(acc[j + 2], 0.f), fmaxf(acc[j + 3], 0.f));
}

extern "C" int cer_corr_encode_f32(const float* feats, const float* w, const float* b, float* out, int nv, int K, long P, int Cout,
                                   void* stream) {
    if (!feats || !w || !b || !out || nv <= 0 || K <= 0 || P <= 0) return CER_EINVAL;
    if (Cout != 64 || K > 256) return CER_ESHAPE;
    if (!cer_aligned16(out)) return CER_EALIGN;
    hipLaunchKernelGGL(corr_encode_kernel, dim3((unsigned)((P + LK_PIX - 1) / LK_PIX)), dim3(256), sizeof(float) * (K * 64 + K * LK_PIX),
                       (hipStream_t)stream, feats, w, b, out, nv, K, P);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
