// EXPERIMENT (not part of libcermvs.so): the lookup kernel with the round-2 compile-time specialisation (3 x 11 windows and the 33-deep 1x1 conv
// unrolled, packed FMAs) that was removed in round 2 after it produced intermittently wrong features when a second process shared the GPU.
// Restored from the repository's own history (commit f55ed20^) in round 4 to reproduce and root-cause that failure:
//   make -C cer-mvs_amd/csrc variants/libcermvs_lkspec.so ; tools/archive/repro_lookup_spec.sh
// K2: multi-level correlation lookup (reference: CorrBlock.__call__ core/corr.py:102-143 and
// bilinear_sampler1 utils/bilinear_sampler.py:6-25 - 528 grid_sample launches per GRU iteration in
// the reference), optionally fused with the view mean (core/update.py:103) and the first
// corr_encoder layer (core/update.py:61-62: Conv2d(33,64,1) + ReLU).
//
// HBM-bound: per pixel one volume row (level0|level1|level2, e.g. 112 floats) is read and
// L*(2r+1) = 33 floats (or 64 encoded floats) are written.  A block stages 64 consecutive rows
// into LDS with coalesced 16-B loads; each thread then owns (pixel, level-slice) windows.
#include "../common.hpp"
#ifndef LKX
#define LKX 0
#endif

#define LK_PIX 64          // pixels per block
#define LK_MAX_ROW 256     // max row_stride (floats)
#define LK_MAX_TAPS 64     // max L*(2r+1)

struct LevelInfo {
    int off[8];
    int len[8];
};

__device__ __forceinline__ float lk_index(float disp, float origin, float incre, int D) {
    // core/corr.py:107 - true division then + D//2, lower clamp only
    const float c = __fadd_rn(__fdiv_rn(__fsub_rn(disp, origin), incre), (float)(D / 2));
    return fmaxf(c, 0.0f);    // NaN -> 0 like torch.maximum? (torch.maximum propagates NaN; disp is never NaN on this path)
}

// one window of 2r+1 taps on level `lv` of the LDS row
__device__ __forceinline__ void lk_window(const float* __restrict__ row, int off, int len, float x, int r, float* __restrict__ o) {
    // x = c / 2^lv (exact); taps at x + dx, dx = -r..r; zero outside [0, len-1] (grid_sample zeros padding, align_corners)
    const float fx = floorf(x);
    const float w = x - fx;
    const bool in_range = fx < (float)(len + r + 1);         // else every tap is outside
    const int i0 = in_range ? (int)fx - r : 0;
    float prev = 0.f;
    {
        const int i = i0;
        prev = (in_range && i >= 0 && i < len) ? row[off + i] : 0.f;
    }
    for (int j = 0; j < 2 * r + 1; ++j) {
        const int i = i0 + j + 1;
        const float next = (in_range && i >= 0 && i < len) ? row[off + i] : 0.f;
        o[j] = prev * (1.0f - w) + next * w;
        prev = next;
    }
}

// the same with a compile-time radius: the 2r + 2 row reads are issued back to back, o has stride 1
template <int R_>
__device__ __forceinline__ void lk_window_ct(const float* __restrict__ row, int off, int len, float x, float* __restrict__ o) {
    const float fx = floorf(x);
    const float w = x - fx;
    const bool in_range = fx < (float)(len + R_ + 1);
    const int i0 = in_range ? (int)fx - R_ : 0;
    float v[2 * R_ + 2];
#pragma unroll
    for (int j = 0; j < 2 * R_ + 2; ++j) {
        const int i = i0 + j;
#if LKX & 2                                                  // (investigation) unconditional reads of a clamped address + select
        const float t = row[off + min(max(i, 0), len - 1)];
        v[j] = (in_range && i >= 0 && i < len) ? t : 0.f;
#else
        v[j] = (in_range && i >= 0 && i < len) ? row[off + i] : 0.f;
#endif
    }
#if LKX & 1                                                  // (investigation) no packed math: every tap's products pinned to scalar instructions
#pragma unroll
    for (int j = 0; j < 2 * R_ + 1; ++j) {
        float a = v[j] * (1.0f - w), bq = v[j + 1] * w;
        asm volatile("" : "+v"(a), "+v"(bq));
        o[j] = a + bq;
    }
#elif LKX & 16                                               // (investigation) packed math, but every register copy is made first and followed by wait states
    {
        static_assert(R_ == 5, "");
        cer_f2 hi[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) hi[q] = (cer_f2){v[2 * q + 2], v[2 * q]};
#pragma unroll
        for (int q = 0; q < 5; ++q) asm volatile("" : "+v"(hi[q]));
        asm volatile("s_nop 7\n s_nop 7");
        const cer_f2 ww = (cer_f2){w, 1.0f - w};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const cer_f2 a = (cer_f2){v[2 * q + 1], v[2 * q + 1]} * ww, bq = hi[q] * ww;
            o[2 * q] = a.x + bq.y;
            o[2 * q + 1] = a.y + bq.x;
        }
        o[10] = v[10] * (1.0f - w) + v[11] * w;
    }
#elif LKX & (512 | 1024 | 2048 | 4096 | 8192 | 16384)                       // (investigation) the compiler's packed sequence written out, with knobs
    {
        static_assert(R_ == 5, "");
        const cer_f2 ww = (cer_f2){w, 1.0f - w};
        cer_f2 pa[5], hi[5];
#pragma unroll
#if LKX & 16384
        for (int q = 0; q < 5; ++q) { pa[q] = (cer_f2){v[2 * q + 1], v[2 * q + 1]}; hi[q] = (cer_f2){v[2 * q], v[2 * q + 2]}; }
#else
        for (int q = 0; q < 5; ++q) { pa[q] = (cer_f2){v[2 * q + 1], v[2 * q + 1]}; hi[q] = (cer_f2){v[2 * q + 2], v[2 * q]}; }
#endif
        const cer_f2 wws = (cer_f2){1.0f - w, w};
        const unsigned addr = (unsigned)(size_t)o;           // LDS byte address of o[0]
#if LKX & 1024
#define LKX_NOPS "s_nop 0\n"
#elif LKX & 2048
#define LKX_NOPS "s_nop 3\n"
#else
#define LKX_NOPS ""
#endif
#if LKX & 4096                                               // alternate result registers: the store data is not overwritten by the next instruction
#define LKX_R1 "v[204:205]"
#define LKX_R1A "v204"
#define LKX_R1B "v205"
#else
#define LKX_R1 "v[200:201]"
#define LKX_R1A "v200"
#define LKX_R1B "v201"
#endif
#if LKX & 8192
#define LKX_MID "s_nop 3\n"
#else
#define LKX_MID "s_nop 0\n"
#endif
#if LKX & 16384                                              // both products laid out so that the sum needs no cross-half selection: HI = (v[2q], v[2q+2]) * (1-w, w)
#define LKX_PAIR(RD, RA, RB, PA, HI, O0, O1)                                                            \
        "v_pk_mul_f32 " RD ", %[ww], " PA " op_sel_hi:[1,0]\n"                                         \
        "v_pk_mul_f32 v[202:203], %[wws], " HI "\n"                                                    \
        "s_nop 0\n"                                                                                    \
        "v_pk_add_f32 " RD ", " RD ", v[202:203]\n" LKX_NOPS                                            \
        "ds_write2_b32 %[addr], " RA ", " RB " offset0:" O0 " offset1:" O1 "\n"
#else
#define LKX_PAIR(RD, RA, RB, PA, HI, O0, O1)                                                            \
        "v_pk_mul_f32 " RD ", %[ww], " PA " op_sel_hi:[1,0]\n"                                         \
        "v_pk_mul_f32 v[202:203], %[ww], " HI "\n" LKX_MID                                             \
        "v_pk_add_f32 " RD ", " RD ", v[202:203] op_sel:[0,1] op_sel_hi:[1,0]\n" LKX_NOPS               \
        "ds_write2_b32 %[addr], " RA ", " RB " offset0:" O0 " offset1:" O1 "\n"
#endif
        asm volatile(
            LKX_PAIR("v[200:201]", "v200", "v201", "%[pa0]", "%[hi0]", "0", "1")
            LKX_PAIR(LKX_R1, LKX_R1A, LKX_R1B, "%[pa1]", "%[hi1]", "2", "3")
            LKX_PAIR("v[200:201]", "v200", "v201", "%[pa2]", "%[hi2]", "4", "5")
            LKX_PAIR(LKX_R1, LKX_R1A, LKX_R1B, "%[pa3]", "%[hi3]", "6", "7")
            LKX_PAIR("v[200:201]", "v200", "v201", "%[pa4]", "%[hi4]", "8", "9")
            :
            : [ww] "v"(ww), [wws] "v"(wws), [addr] "v"(addr), [pa0] "v"(pa[0]), [hi0] "v"(hi[0]), [pa1] "v"(pa[1]), [hi1] "v"(hi[1]), [pa2] "v"(pa[2]),
              [hi2] "v"(hi[2]), [pa3] "v"(pa[3]), [hi3] "v"(hi[3]), [pa4] "v"(pa[4]), [hi4] "v"(hi[4])
            : "v200", "v201", "v202", "v203", "v204", "v205", "memory");
        o[10] = v[10] * (1.0f - w) + v[11] * w;
    }
#elif LKX & (64 | 128 | 256)                                 // (investigation) N wait states between each pair's packed sum and its LDS store
#pragma unroll
    for (int j = 0; j + 1 < 2 * R_ + 1; j += 2) {
        float a = v[j] * (1.0f - w) + v[j + 1] * w, bq = v[j + 1] * (1.0f - w) + v[j + 2] * w;
#if LKX & 64
        asm volatile("s_nop 0" : "+v"(a), "+v"(bq));
#elif LKX & 256
        asm volatile("s_nop 3" : "+v"(a), "+v"(bq));
#else
        asm volatile("" : "+v"(a), "+v"(bq));               // (control: the same code motion barrier without a wait state)
#endif
        o[j] = a;
        o[j + 1] = bq;
    }
    o[2 * R_] = v[2 * R_] * (1.0f - w) + v[2 * R_ + 1] * w;
#elif LKX & 32                                               // (investigation) packed math as the compiler writes it, all LDS stores after the last of it
    {
        float rr[2 * R_ + 1];
#pragma unroll
        for (int j = 0; j < 2 * R_ + 1; ++j) rr[j] = v[j] * (1.0f - w) + v[j + 1] * w;
#pragma unroll
        for (int j = 0; j < 2 * R_ + 1; ++j) asm volatile("" : "+v"(rr[j]));
#pragma unroll
        for (int j = 0; j < 2 * R_ + 1; ++j) o[j] = rr[j];
    }
#elif LKX & 8                                                // (investigation) wait states behind every register copy the packed form needs
#pragma unroll
    for (int j = 0; j < 2 * R_ + 2; ++j) asm volatile("s_nop 4" : "+v"(v[j]));
#pragma unroll
    for (int j = 0; j < 2 * R_ + 1; ++j) o[j] = v[j] * (1.0f - w) + v[j + 1] * w;
#else
#pragma unroll
    for (int j = 0; j < 2 * R_ + 1; ++j) o[j] = v[j] * (1.0f - w) + v[j + 1] * w;
#endif
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
    const int rsp = rs + 4;                                  // padded LDS stride: rs % 32 == 16 or 0 -> +4 breaks the conflict pattern
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
        lk_window(&rows[pix * rsp], li.off[lv], li.len[lv], x, r, o);
        float* dst = out + ((long)v * L * taps + (long)lv * taps) * P + p;
        for (int j = 0; j < taps; ++j) dst[(long)j * P] = o[j];
    }
}

// fused: lookup on the folded volume + 1x1 conv (taps_total -> 64) + bias + ReLU, out [P,64] NHWC (or split32 / frag16).
// HBM-bound (one 4*rs-byte volume row in, 256 bytes out per pixel; 3.6 us of VALU work for the 1x1 conv at 296 x 400), so the
// kernel is built around keeping loads in flight: persistent blocks (3 per CU) walk over 64-pixel tiles; the NEXT tile's rows are
// requested into registers (16-byte loads, <= 16 per thread) right after the current tile's have been written to LDS, and arrive
// while the block does the windows and the 1x1 conv of the current tile.  Two barriers per tile: rows -> [B1] -> windows (one
// level per wave) into a double-buffered feature tile -> [B2] -> conv (1x1 weights through the scalar cache: wave-uniform).
// <L_, R_> = <3, 5> (the model's 3 levels x 11 taps) compiles the windows and the 33-deep 1x1 conv fully unrolled - the feature
// vector sits in registers and the scalar weight loads are issued ahead of the FMAs that use them; <0, 0> is the generic form.
#define LK_MAX_PRE 16      // float4 per thread of one 64-row tile: 64 * (LK_MAX_ROW / 4) / 256
template <int L_, int R_>
__global__ __launch_bounds__(256) void lookup_encode_kernel(const float* __restrict__ vol, const float* __restrict__ origin,
                                                            const float* __restrict__ disp, const float* __restrict__ wgt,
                                                            const float* __restrict__ bias, float* __restrict__ out, long P, int D, int rs,
                                                            float incre, int L, int r, LevelInfo li, int out_split, float out_scale, int img_w,
                                                            int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lk_smem[];
    const int rsp = rs + 4;
    const int taps = 2 * r + 1, K = L * taps, FS = K | 1, fstride = LK_PIX * FS;     // (odd pixel stride: conflict-free columns)
    float* rows = lk_smem;                                   // [LK_PIX][rs + 4]
    float* feats = lk_smem + LK_PIX * rsp;                   // [2][LK_PIX][FS]
    const int n4 = rs / 4, npre = (LK_PIX * n4 + 255) / 256;
    const int pix = threadIdx.x & 63;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave id: lookup level, then output-channel group
    float4 pre[LK_MAX_PRE];
    float pre_d = 0.f, pre_o = 0.f;
    // float4 number t = tid + 256 i of the tile is (row pr, quad q) = (t / n4, t % n4): walked incrementally (no division per item)
    const int pr0 = threadIdx.x / n4, q0 = threadIdx.x - pr0 * n4, dpr = 256 / n4, dq = 256 - dpr * n4;
    auto request = [&](int tile) {                           // rows of `tile` -> registers (zeros past the last pixel)
        const long p0 = (long)tile * LK_PIX;
        const int npix = (int)min((long)LK_PIX, P - p0);
        const float* src = vol + p0 * rs;
        if (pix < npix) { pre_d = disp[p0 + pix]; pre_o = origin[p0 + pix]; }    // ... and this thread's pixel's window position
        int pr = pr0, q = q0;
#pragma unroll
        for (int i = 0; i < LK_MAX_PRE; ++i) {            // (predicated, not `break`: pre[] must stay in registers)
            if (i < npre) {
                pre[i] = pr < npix ? cer_ld4(src + (long)(threadIdx.x + 256 * i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        const float c = active ? lk_index(pre_d, pre_o, incre, D) : 0.f;
        {
            int pr = pr0, q = q0;
#pragma unroll
            for (int i = 0; i < LK_MAX_PRE; ++i) {
                if (i < npre) {
                    if (pr < LK_PIX) *reinterpret_cast<float4*>(&rows[pr * rsp + 4 * q]) = pre[i];
                    pr += dpr; q += dq;
                    if (q >= n4) { q -= n4; ++pr; }
                }
            }
        }
        __syncthreads();                                     // [B1] rows complete; the previous tile's windows were read before its [B2]
#if !(LKX & 4)
        if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
#endif
        float* ft = feats + (it & 1) * fstride;
        if (active) {
            if constexpr (L_ > 0) {
                if (grp < L_) lk_window_ct<R_>(&rows[pix * rsp], li.off[grp], li.len[grp], c / (float)(1 << grp), &ft[pix * FS + grp * taps]);
            } else {
                for (int lv = grp; lv < L; lv += 4)        // (straight into the feature tile: no per-thread array)
                    lk_window(&rows[pix * rsp], li.off[lv], li.len[lv], c / (float)(1 << lv), r, &ft[pix * FS + lv * taps]);
            }
        }
        __syncthreads();                                     // [B2] features complete; rows free for the next tile
#if LKX & 4                                                  // (investigation) the next tile's rows are requested AFTER the windows
        if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
#endif
        if (!active) continue;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias[grp * 16 + j];
        if constexpr (L_ > 0) {
            constexpr int KT = L_ * (2 * R_ + 1);
            // packed fp32 FMAs (v_pk_fma_f32: two channels per instruction, each component an exact fma like fmaf) - a plain
            // v_fma_f32 takes 4 cycles per wave and the 33 x 16 of them made this phase the longest of the kernel
            float f[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) f[k] = ft[pix * FS + k];
            cer_f2 a2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a2[j] = (cer_f2){acc[2 * j], acc[2 * j + 1]};
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float* wr = wgt + k * 64 + grp * 16;   // wave-uniform: scalar loads
                const cer_f2 fk = (cer_f2){f[k], f[k]};
#pragma unroll
                for (int j = 0; j < 8; ++j) a2[j] = __builtin_elementwise_fma(fk, (cer_f2){wr[2 * j], wr[2 * j + 1]}, a2[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[2 * j] = a2[j].x; acc[2 * j + 1] = a2[j].y; }
        } else {
#pragma unroll 3
            for (int k = 0; k < K; ++k) {
                const float f = ft[pix * FS + k];
                const float* wr = wgt + k * 64 + grp * 16;   // wave-uniform: scalar loads
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(f, wr[j], acc[j]);
            }
        }
        if (out_split == 2) {
            // frag16 layout (cer_mvs.h, conv_s16.hip): this thread's 16 channels are group `grp` of its pixel's m-tile: four 16-byte
            // pieces (hi | lo planes x channel octets) of relu(acc) * out_scale
            const unsigned p = (unsigned)(p0 + pix);             // (P < 2^31 is checked by the launcher: 32-bit division)
            const unsigned y = p / (unsigned)img_w, x = p - y * (unsigned)img_w;
            const long mt = (long)(y >> 1) * ((img_w + 15) >> 4) + (x >> 4);
            char* dst = reinterpret_cast<char*>(out) + ((mt * 4 + grp) * 2) * 1024 + (((y & 1) << 4) | (x & 15)) * 16;
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

static int level_info(int D, int rs, int L, int r, LevelInfo* li) {
    if (L <= 0 || L > 8 || r < 0 || r > 15 || L * (2 * r + 1) > LK_MAX_TAPS) return CER_ESHAPE;
    if (rs % 4 != 0 || rs > LK_MAX_ROW) return CER_ESHAPE;
    int off = 0, n = D;
    for (int l = 0; l < L; ++l) {
        li->off[l] = off;
        li->len[l] = n;
        off += n;
        n /= 2;
    }
    if (off > rs) return CER_ESHAPE;
    return CER_OK;
}

extern "C" int cer_corr_lookup_f32(const float* vol, const float* origin, const float* disp, long disp_view_stride, float* out, int nv,
                                   long P, int D, int row_stride, double incre, int num_levels, int radius, void* stream) {
    if (!vol || !origin || !disp || !out || nv <= 0 || P <= 0 || D <= 0) return CER_EINVAL;
    if (!cer_aligned16(vol)) return CER_EALIGN;
    LevelInfo li;
    int rc = level_info(D, row_stride, num_levels, radius, &li);
    if (rc) return rc;
    hipLaunchKernelGGL(lookup_kernel, dim3((unsigned)((P + LK_PIX - 1) / LK_PIX), (unsigned)nv), dim3(256),
                       sizeof(float) * LK_PIX * (row_stride + 4), (hipStream_t)stream, vol, origin,
                       disp, disp_view_stride, out, P, D, row_stride, (float)incre, num_levels, radius, li);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_lookup_encode_f32(const float* vol, const float* origin, const float* disp, const float* w, const float* b, float* out,
                                     long P, int D, int row_stride, double incre, int num_levels, int radius, int Cout, int out_split,
                                     int log2s_out, int img_w, void* stream) {
    if (!vol || !origin || !disp || !w || !b || !out || P <= 0 || D <= 0) return CER_EINVAL;
    if (Cout != 64 || num_levels > 4) return CER_ESHAPE;
    if (out_split == 2 && (img_w <= 0 || P % img_w != 0 || P >= (1L << 31))) return CER_ESHAPE;
    if (!cer_aligned16(vol) || !cer_aligned16(out)) return CER_EALIGN;
    LevelInfo li;
    int rc = level_info(D, row_stride, num_levels, radius, &li);
    if (rc) return rc;
    const int generic_req = out_split & 0x100;               // (investigation) out_split | 0x100: the runtime-loop form for THIS call
    out_split &= 0xFF;
    const int K = num_levels * (2 * radius + 1);
    const long ntiles = (P + LK_PIX - 1) / LK_PIX;
    if (ntiles >= (1L << 30)) return CER_ESHAPE;
    const size_t smem = sizeof(float) * LK_PIX * ((size_t)(row_stride + 4) + 2 * (K | 1));
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const long resident = (long)ncu * (smem <= 50 * 1024 ? 3 : smem <= 76 * 1024 ? 2 : 1);
    const unsigned grid = (unsigned)(ntiles < resident ? ntiles : resident);
    static int force_generic = -1;                           // (investigation) CER_LK_GENERIC=1: always the runtime-loop form
    if (force_generic < 0) { const char* e = getenv("CER_LK_GENERIC"); force_generic = (e && e[0] == '1') ? 1 : 0; }
    if (num_levels == 3 && radius == 5 && !force_generic && !generic_req)
        hipLaunchKernelGGL((lookup_encode_kernel<3, 5>), dim3(grid), dim3(256), smem, (hipStream_t)stream, vol, origin, disp, w, b, out, P, D, row_stride,
                           (float)incre, num_levels, radius, li, out_split, ldexpf(1.0f, log2s_out), img_w, (int)ntiles);
    else
        hipLaunchKernelGGL((lookup_encode_kernel<0, 0>), dim3(grid), dim3(256), smem, (hipStream_t)stream, vol, origin, disp, w, b, out, P, D, row_stride,
                           (float)incre, num_levels, radius, li, out_split, ldexpf(1.0f, log2s_out), img_w, (int)ntiles);
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
