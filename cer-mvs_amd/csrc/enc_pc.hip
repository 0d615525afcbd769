// Encoder convolutions, round 4: producer / consumer wave specialisation (reference: core/extractor.py:49-57,143-155).
//
// The round-2/3 kernels (enc_conv.hip) run every phase of a tile in every wave - load the halo, normalise + ReLU + split to
// hi|lo f16, write LDS, barrier, MFMAs, barrier, epilogue - and by their own cycle trace spend 30 % of a block's life preparing
// operands (pure VALU), 34 % in the matrix phase and the rest in epilogue / waits / barriers: the phases ADD.  Here a 512-thread
// block (one per CU, persistent) is split by role:
//   * waves 0-3, PRODUCERS: fetch the next unit's halo (one unit = one tile x one 32-channel chunk) into registers, apply the
//     producer layer's instance norm + ReLU - or form the residual merge relu(fa(A) + fb(B)) of TWO tensors on the fly, which
//     removes enc_merge_kernel as a pass - split to hi|lo f16 and write one of two LDS buffers; optionally the merged activation
//     is written back once (core pixels only) for the residual branch that needs it later;
//   * waves 4-7, CONSUMERS: fragments by ds_read_b128, weights resident in registers (<= 18 k16-steps) or streamed from L2 two
//     steps ahead, 3 f16 MFMAs per product term set (x*w = xh*wh + 2^-11 (xh*wl' + xl'*wh), two fp32 accumulators - the
//     arithmetic of enc_conv.hip, bit for bit), epilogue straight from the accumulators (a lane owns one output channel of 16
//     pixels: 128-byte contiguous dword stores per half wave, per-lane statistics without any shuffle).
// One s_barrier per unit couples the two groups: the producers are always one unit ahead (buffer k+1 is filled while buffer k
// is multiplied), so the vector work of operand preparation runs UNDER the matrix work of the same SIMD instead of in front of it
// (MI355X_MICROARCH.md: the MFMA and VALU pipes of a SIMD run concurrently for two different waves).
// Happens-before for the LDS buffers (unit k lives in buffer k & 1; barrier #k is the k-th s_barrier every wave executes):
//   producer: commit(k) -> lgkmcnt(0) -> barrier #k           consumer: barrier #k -> ds_reads of unit k -> (MFMAs wait for them)
//   buffer k & 1 is rewritten by commit(k + 2), which the producers start after barrier #k+1; a consumer arrives at barrier
//   #k+1 only after every fragment of unit k has landed in its registers (each MFMA waits for its operands), so no ds_read of
//   unit k is outstanding when the first ds_write of unit k + 2 issues.
// Statistics partials: consumers leave [row group][channel] (sum, sum of squares) in a double-buffered LDS patch; producer wave 0
// adds the row groups in fixed order after the next barrier and writes the tile's record: deterministic, one record per tile
// whatever block processed it (batch invariance, tests/test_hip_parity.py::test_encoder_engine_batch_invariance_at_tnt_size).
//
// Round 6, the FP6-correction form (flags & 8; weights from cer_enc_conv_pack_f6): the two correction terms of a 32-channel tap -
// xh*wl' + xl'*wh for both k16-steps, 2 x 2 f16 MFMAs = 128 matrix-pipe cycles per m-tile - become ONE v_mfma_scale_f32_32x32x64_f8f6f4 with
// both operands in e2m3 (8 passes = 32 cycles), accumulated into the main term's accumulator: 96 instead of 192 cycles per tap and m-tile.
// K block of lane (pixel or output channel, kg): 32 six-bit fields, field i at bit 6 i = [xh (8) | xl' (8)] of channels 8kg..8kg+7 of k16-step 0,
// then of k16-step 1 (weights: [wl' | wh] in the same positions), all divided by one power of two per block - activations: s = 2^(e - 2), e the
// exponent of the block's largest |xh| (|xl'| <= |xh| element by element: nothing saturates), formed by the producers (packed 16-bit maxima of
// the two items of a block - lanes tid and tid ^ 2 - one DPP exchange, ONE v_cvt_scalef32_pk32_fp6_f16 per item); weights: from the block
// maximum on the host, times the 2^-11 that puts both terms on the accumulator's scale.  LDS: the 24 bytes + E8M0 byte of a block take the
// place of the block's lo halves (dwords 0-3 where lo of k16-step 0 was, dwords 4-5 | scale where lo of step 1 was): the consumers' fragment
// addresses do not change.  Costed on the oracle first (tools/experiments/encoder_corr_numerics.py: 7e-6 end to end), its matrix time and the
// producers' slack measured with ablation builds (profiles/r06_encoder_fp6_upper_bound.txt) before the kernels were written.
#include "common.hpp"
#include <math.h>
#include <string.h>
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int pc_u32x4 __attribute__((ext_vector_type(4)));
typedef int pc_intx8 __attribute__((ext_vector_type(8)));

#ifndef PC_ABL
#define PC_ABL 0                       // profiling ablations (variant builds only): 1 no MFMAs, 2 no output stores, 4 no operand transform (halves of the raw bits),
                                       // 8 / 16: the MATRIX TIME of an fp8- / FP6-correction form (2 of 3 / 3 of 6 MFMAs issued; results wrong) - upper bounds
#endif
#ifndef PC_EXP24
#define PC_EXP24 0
#endif
#ifndef PC_PAD
#define PC_PAD 0
#endif
#ifndef PC_TRACE
#define PC_TRACE 0                     // debug build: per-wave phase cycle sums written to a.out2 [blocks][8 waves][8] (u64); tools/archive/trace_pc.py
#endif
#if PC_TRACE
#define PC_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tsum[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define PC_T(k) do { } while (0)
#endif
#define PC_AS 144                      // LDS bytes per halo pixel and 32-channel chunk: 32 hi | 32 lo | 16 pad (conflict-free ds_read_b128)
#define PC_EPI_RAW 0
#define PC_EPI_FMAP 1
#define PC_EPI_CTX 2
#define PC_EPI_FSPLIT 3                // feature maps straight into the cost volume's split-f16 operand planes (cost_lines.hip: feat_split_kernel)

struct PcArgs {
    const float* srcA;                 // [N, h*w, CIN] channels-last
    const float* tfA;                  // statistics [N*CIN][2] (mean, rstd) or NULL
    const float* srcB;                 // second tensor of a residual merge (same geometry) or NULL
    const float* tfB;
    int flags;                         // 1 ReLU on the A term, 2 ReLU on the B term, 4 ReLU on the sum (enc_merge_kernel's flags)
    float* mout;                       // merged activation written back [N, h*w, CIN] or NULL (stride 1 only)
    const _Float16* wpk;               // cer_enc_conv_pack order
    const float* bias;
    float* out;
    float* out2;
    float* part;                       // statistics partials [N][nblk][COUT][2] or NULL
    int h, w, ho, wo;
    int tiles_x, nblk, total_tiles;
    int out_border;
    float out_scale;
};

template <int CIN, int COUT, int STRIDE, int TAPS>
struct PcCfg {
    static constexpr int NCH = CIN / 32, NT = COUT / 32;
    static constexpr int PAD = TAPS == 9 ? 1 : 0;
    static constexpr int WN = NT >= 4 ? 4 : NT, WM = 4 / WN;                    // consumer waves along channels / rows
    static constexpr int TH = (TAPS == 9 && STRIDE == 2) ? 2 : (NT >= 4 ? 4 : 8);
    static constexpr int RPW = TH / WM;                                         // output rows (m-tiles) per consumer wave
    static constexpr int HH = TAPS == 9 ? (TH - 1) * STRIDE + 3 : TH;
    static constexpr int HW = TAPS == 9 ? 31 * STRIDE + 3 : 32;
    static constexpr int ROWS = HH * HW;                                        // halo pixels of a unit
    static constexpr int ITEMS = (ROWS + 63) / 64;                              // (pixel, 8-channel group) items per producer lane
    static constexpr int BUF = ROWS * PC_AS;
    static constexpr int NS = TAPS * 2;                                         // k16-steps per unit
    static constexpr bool WRES = NCH * NS <= 18;                                // weights resident in registers
    static constexpr int RED = 2 * WM * COUT * 2 * 4;                           // bytes of the statistics patches
    static constexpr int PATCH = 4 * 32 * 36 * 4;                               // epilogue transpose: 32 pixels x 32 channels (pitch 36) per consumer wave
    static constexpr int SMEM = 2 * BUF + RED + PATCH;
};

__device__ __forceinline__ void pc_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");            // (the bare builtin does not order LDS accesses for the compiler)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS pixel index of halo pixel (hy, hx): stride-2 3x3 tiles keep even and odd columns in separate planes of a row, so that the
// consumers' fragment reads (lane = output column) stay unit-stride
template <int STRIDE, int TAPS, int HW>
__device__ __forceinline__ int pc_lds_pixel(int hy, int hx) {
    if (TAPS == 9 && STRIDE == 2) return hy * HW + (hx & 1) * ((HW + 1) / 2) + (hx >> 1);
    return hy * HW + hx;
}

// Work order.  Block b runs on XCD b % 8 (observed, not promised: used for speed only), and every XCD has its own L2.  The j-th work
// item of block b is index j * G + (b % 8) * (G / 8) + b / 8 of a COLUMN-major enumeration of an image's tiles: at any time the 32
// blocks of an XCD work on 32 vertically adjacent tiles, so the two halo rows a tile shares with the tile above / below (25 % extra
// reads with 8-row tiles) are fetched into that XCD's L2 once instead of crossing the fabric twice.  The tile's identity (statistics
// record, output position) stays row-major.
#ifndef PC_XCD_ORDER
#define PC_XCD_ORDER 1
#endif
__device__ __forceinline__ int pc_work_offset() {
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    return (PC_XCD_ORDER && G % 8 == 0) ? (b % 8) * (G / 8) + b / 8 : b;
}
__device__ __forceinline__ void pc_work_tile(const PcArgs& a, int widx, int& img, int& tile, int& ty, int& tx) {
    img = widx / a.nblk;
    const int c = widx - img * a.nblk;
    if (PC_XCD_ORDER) {
        const int tiles_y = a.nblk / a.tiles_x;
        tx = c / tiles_y;
        ty = c - tx * tiles_y;
    } else {
        ty = c / a.tiles_x;
        tx = c - ty * a.tiles_x;
    }
    tile = ty * a.tiles_x + tx;
}

// ------------------------------------------------------------------------------------------------------------------ producers
// Split of eight values given as y = 2048 x (the factor rides on rstd: a power of two, exact): hi = f16(y 2^-11) = f16(x),
// lo = f16(y - 2048 hi) = f16((x - hi) 2^11) - the operands of enc_conv.hip bit for bit, in two v_fma_mix instructions per value
// (the conversions, the subtraction and the scaling of cer_split8 cost 5; VALU issue is what the producers are bound by)
__device__ __forceinline__ void pc_split8_scaled(const float (&y)[8], half8& hi, half8& lo) {
    const float c_dn = 1.0f / 2048.0f, c_up = -2048.0f;
    unsigned h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[k]) : "v"(y[2 * k]), "s"(c_dn));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[k]) : "v"(y[2 * k + 1]), "s"(c_dn));
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l[k]) : "v"(h[k]), "s"(c_up), "v"(y[2 * k]));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[k]) : "v"(h[k]), "s"(c_up), "v"(y[2 * k + 1]));
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(half8, (u32x4){h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(half8, (u32x4){l[0], l[1], l[2], l[3]});
}

#if PC_ABL & 1
__device__ __forceinline__ floatx16 pc_keep(half8 a_, half8 b_, floatx16 c_) { asm volatile("" :: "v"(a_), "v"(b_)); return c_; }
#define PC_MFMA(a_, b_, c_) pc_keep(a_, b_, c_)
#else
#define PC_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)
#endif
#if PC_ABL & 24
__device__ __forceinline__ floatx16 pc_keep2(half8 a_, half8 b_, floatx16 c_) { asm volatile("" :: "v"(a_), "v"(b_)); return c_; }
#define PC_MFMA_C3(a_, b_, c_, s_) pc_keep2(a_, b_, c_)                                        // lo x hi: never issued
#define PC_MFMA_C2(a_, b_, c_, s_) (((PC_ABL & 16) && ((s_) & 1)) ? pc_keep2(a_, b_, c_) : PC_MFMA(a_, b_, c_))   // hi x lo: FP6 time = every other step
#else
#define PC_MFMA_C3(a_, b_, c_, s_) PC_MFMA(a_, b_, c_)
#define PC_MFMA_C2(a_, b_, c_, s_) PC_MFMA(a_, b_, c_)
#endif
#define PC_YMAX 134152192.0f           // 65504 * 2048: what the f16 hi half can hold, in the scaled domain

// FP6 form: the item's 16 values [hi (8) | lo' (8)] -> 16 e2m3 fields (three dwords) under the block's scale; `sb` = its E8M0 byte
__device__ __forceinline__ void pc_fp6_item(const half8 hi, const half8 lo, unsigned (&f)[3], unsigned& sb) {
    typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
    union U2 { unsigned u; ushort2_t s; };
    const pc_u32x4 hb = __builtin_bit_cast(pc_u32x4, hi);
    U2 a0, a1, a2, a3;                                                           // f16 bit patterns without their sign order like the magnitudes
    a0.u = hb.x & 0x7FFF7FFFu; a1.u = hb.y & 0x7FFF7FFFu; a2.u = hb.z & 0x7FFF7FFFu; a3.u = hb.w & 0x7FFF7FFFu;
    a0.s = __builtin_elementwise_max(a0.s, a1.s);
    a2.s = __builtin_elementwise_max(a2.s, a3.s);
    a0.s = __builtin_elementwise_max(a0.s, a2.s);
    unsigned mm = max(a0.u & 0xFFFFu, a0.u >> 16);
    mm = max(mm, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mm, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]: the block's other k16-step
    sb = ((mm + 0x40u) >> 10) + 110u;                                            // E8M0 of s = 2^(bexp - 15 - 2); + 0x40: mantissas >= 1.9375 go one up
    const float sc = __uint_as_float(sb << 23);
    typedef _Float16 half32_t __attribute__((ext_vector_type(32)));
    typedef _Float16 half16_t __attribute__((ext_vector_type(16)));
    typedef int intx6_t __attribute__((ext_vector_type(6)));
    const half16_t in = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const half32_t in32 = __builtin_shufflevector(in, in, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                                  -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const intx6_t f6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(in32, sc);     // (divides by its scale operand, rounds to nearest even, saturates at 7.5)
    f[0] = (unsigned)f6[0]; f[1] = (unsigned)f6[1]; f[2] = (unsigned)f6[2];
}

template <int CIN, int COUT, int STRIDE, int TAPS, bool DUAL, bool F6>
__device__ __forceinline__ void pc_producer(const PcArgs& a, char* __restrict__ lds, const float* __restrict__ red, int tid) {
    using C = PcCfg<CIN, COUT, STRIDE, TAPS>;
    constexpr int PS = TAPS == 9 ? 1 : STRIDE;                                   // pixel step of the halo in the source (1x1 stride 2: every other pixel)
    const int g = tid & 3, prow0 = (tid >> 2) & 63;
    int hyv[C::ITEMS], hxv[C::ITEMS], lpix[C::ITEMS], ioff[C::ITEMS];
#pragma unroll
    for (int i = 0; i < C::ITEMS; ++i) {
        const int row = min(prow0 + 64 * i, C::ROWS - 1);
        hyv[i] = row / C::HW;
        hxv[i] = row - hyv[i] * C::HW;
        lpix[i] = pc_lds_pixel<STRIDE, TAPS, C::HW>(hyv[i], hxv[i]) * PC_AS + g * 16;
        ioff[i] = (hyv[i] * PS * a.w + hxv[i] * PS) * CIN + 8 * g;              // element offset from the halo's first pixel (interior units)
    }
    // clamp bounds in the scaled domain: ReLU = lower bound 0; without it the lower bound is the f16 range (cer_split2's clamp)
    const float lowA = (a.flags & 1) ? 0.f : -PC_YMAX, lowB = (a.flags & 2) ? 0.f : -PC_YMAX, lowS = (a.flags & 4) ? 0.f : -PC_YMAX;
    const bool bplain = DUAL && !a.tfB && !(a.flags & 2);                        // the second tensor enters as it is (an already merged activation)
    float4 rawA[C::ITEMS][2], rawB[DUAL ? C::ITEMS : 1][2];
    const int G = (int)gridDim.x;
    const int woff = pc_work_offset();
    const int ntile = max(0, (a.total_tiles - woff + G - 1) / G);                // tiles of this block
    const int U = ntile * C::NCH;

    auto geom = [&](int u, int& img, int& tile, int& ty0, int& tx0, int& c0) {
        int ty, tx;
        pc_work_tile(a, woff + (u / C::NCH) * G, img, tile, ty, tx);
        ty0 = ty * C::TH;
        tx0 = tx * 32;
        c0 = (u % C::NCH) * 32;
    };
    // whole halo inside the image (and, for the write-back / 1x1 forms, the whole tile inside the output): no clamps, no masks
    auto is_interior = [&](int ty0, int tx0) -> bool {
        const int y0 = ty0 * STRIDE - C::PAD, x0 = tx0 * STRIDE - C::PAD;
        return y0 >= 0 && x0 >= 0 && y0 + (C::HH - 1) * PS < a.h && x0 + (C::HW - 1) * PS < a.w;
    };
    // the NEXT unit's halo loads are issued item by item, each right behind the commit of the same item of the current unit (a burst
    // of 12-24 KiB-sized loads per wave at the end of a commit queued in front of everything else the CU had to fetch or store)
    // (branch-free: with scalar branches inside the item loop hipcc falls back to vmcnt(0) at the top of a commit - every load of
    // the unit waited for at once; straight-line code gets counted waits, one item at a time.  Past the block's last unit the
    // current unit is simply requested again.)
    bool nx_interior = false;
    const float *nx_pa = nullptr, *nx_pb = nullptr;
    int nx_y0 = 0, nx_x0 = 0;
    auto prepare = [&](int u) {                                                  // addressing of unit u's loads (uniform)
        int img, tile, ty0, tx0, c0;
        geom(min(u, U - 1), img, tile, ty0, tx0, c0);
        const long ibase = (long)img * a.h * a.w * CIN + c0;
        nx_interior = is_interior(ty0, tx0);
        nx_y0 = ty0 * STRIDE - C::PAD;
        nx_x0 = tx0 * STRIDE - C::PAD;
        nx_pa = a.srcA + ibase;
        nx_pb = DUAL ? a.srcB + ibase : nullptr;
    };
    auto issue_item = [&](int i) {
        // interior units: the precomputed offset from the halo's first pixel; border units: clamped coordinates (masked at commit)
        const int gy = min(max(nx_y0 + hyv[i] * PS, 0), a.h - 1), gx = min(max(nx_x0 + hxv[i] * PS, 0), a.w - 1);
        const int off = nx_interior ? (nx_y0 * a.w + nx_x0) * CIN + ioff[i] : (gy * a.w + gx) * CIN + 8 * g;
        rawA[i][0] = cer_ld4(nx_pa + off);
#if PC_EXP24       // timing experiment (WRONG results): 24 instead of 32 bytes per item in, three of four 16-byte stores out - the traffic of a 3-byte format
        { const float2 t2 = *reinterpret_cast<const float2*>(nx_pa + off + 4); rawA[i][1] = make_float4(t2.x, t2.y, t2.x, t2.y); }
#else
        rawA[i][1] = cer_ld4(nx_pa + off + 4);
#endif
        if (DUAL) {
            rawB[DUAL ? i : 0][0] = cer_ld4(nx_pb + off);
#if PC_EXP24
            { const float2 t2 = *reinterpret_cast<const float2*>(nx_pb + off + 4); rawB[DUAL ? i : 0][1] = make_float4(t2.x, t2.y, t2.x, t2.y); }
#else
            rawB[DUAL ? i : 0][1] = cer_ld4(nx_pb + off + 4);
#endif
        }
    };
    float muA[8], rsA[8], muB[8], rsB[8];                                        // rs = 2048 rstd
    int cur_key = -1;
    auto commit_items = [&](auto border_tag, auto bplain_tag, int img, int ty0, int tx0, int c0, char* buf) {
        constexpr bool BORDER = decltype(border_tag)::value, BPLAIN = decltype(bplain_tag)::value;
#pragma unroll
        for (int i = 0; i < C::ITEMS; ++i) {
            // (lanes whose item index lies beyond ROWS hold the clamped last halo pixel: they repeat its owner's writes - same
            // address, same value - instead of branching around the item)
            {
                const int sy = ty0 * STRIDE + hyv[i] * PS - C::PAD, sx = tx0 * STRIDE + hxv[i] * PS - C::PAD;
                const bool inside = !BORDER || (sy >= 0 && sy < a.h && sx >= 0 && sx < a.w);
                const float va[8] = {rawA[i][0].x, rawA[i][0].y, rawA[i][0].z, rawA[i][0].w, rawA[i][1].x, rawA[i][1].y, rawA[i][1].z, rawA[i][1].w};
                constexpr int ib = DUAL ? 1 : 0;                                 // (rawB has one row when there is no second tensor)
                const float4 b0 = rawB[ib * i][0], b1 = rawB[ib * i][1];
                const float vb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float yv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y;
                    if (!DUAL) {
                        y = __builtin_amdgcn_fmed3f((va[e] - muA[e]) * rsA[e], lowA, PC_YMAX);
                    } else {
                        const float ya = fmaxf((va[e] - muA[e]) * rsA[e], lowA);
                        const float yb = BPLAIN ? vb[e] * 2048.0f : fmaxf((vb[e] - muB[e]) * rsB[e], lowB);
                        y = __builtin_amdgcn_fmed3f(ya + yb, lowS, PC_YMAX);
                    }
                    yv[e] = (BORDER && !inside) ? 0.f : y;                       // zero padding of the (normalised) activation
                }
                half8 hi, lo;
                if (PC_ABL & 4) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    hi = __builtin_bit_cast(half8, (f32x4){va[0], va[1], va[2], va[3]});
                    lo = __builtin_bit_cast(half8, (f32x4){va[4], va[5], va[6], va[7]});
                } else
                pc_split8_scaled(yv, hi, lo);
#if PC_PAD          // timing experiment: PC_PAD extra VALU instructions per 8-value item (what a narrower correction format's conversion would add)
                {
                    unsigned pad_ = __builtin_bit_cast(unsigned, yv[0]);
#pragma unroll
                    for (int q_ = 0; q_ < PC_PAD; ++q_) asm volatile("v_add_u32 %0, %0, %1" : "+v"(pad_) : "v"(q_ + 1));
                    asm volatile("" :: "v"(pad_));
                }
#endif
                *reinterpret_cast<half8*>(buf + lpix[i]) = hi;
                if constexpr (F6) {
                    // block (pixel, kg = g & 1): dwords 0-3 at +64 + 16 kg, dwords 4-5 | scale at +96 + 16 kg; this item (k16-step g >> 1) owns
                    // dwords 0-2 (step 0) or 3-5 (step 1); the scale is the same in both lanes of the block
                    unsigned f[3], sb;
                    pc_fp6_item(hi, lo, f, sb);
                    char* q = buf + lpix[i] + (64 - 16 * g) + 16 * (g & 1);
                    char* q0 = q + ((g >> 1) ? 12 : 0);
                    char* q1 = q + ((g >> 1) ? 32 : 4);
                    *reinterpret_cast<unsigned*>(q0) = f[0];
                    *reinterpret_cast<unsigned*>(q1) = f[1];
                    *reinterpret_cast<unsigned*>(q1 + 4) = f[2];
                    *reinterpret_cast<unsigned*>(q + 40) = sb;
                } else
                *reinterpret_cast<half8*>(buf + lpix[i] + 64) = lo;
                if (DUAL && STRIDE == 1 && a.mout) {                             // merged activation, once per pixel: the tile's core
                    const bool core = TAPS == 9 ? (hyv[i] >= 1 && hyv[i] <= C::TH && hxv[i] >= 1 && hxv[i] <= 32) : true;
                    if (core && inside) {
                        float* m = a.mout + (long)img * a.h * a.w * CIN + ((long)sy * a.w + sx) * CIN + c0 + 8 * g;
                        constexpr float dn = 1.0f / 2048.0f;
                        *reinterpret_cast<float4*>(m) = make_float4(yv[0] * dn, yv[1] * dn, yv[2] * dn, yv[3] * dn);
                        *reinterpret_cast<float4*>(m + 4) = make_float4(yv[4] * dn, yv[5] * dn, yv[6] * dn, yv[7] * dn);
                    }
                }
            }
            issue_item(i);
        }
    };
    auto commit = [&](int u) {
        int img, tile, ty0, tx0, c0;
        geom(u, img, tile, ty0, tx0, c0);
        const int key = img * C::NCH + (u % C::NCH);
        if (key != cur_key) {                                                    // statistics of this (image, chunk): block-uniform branch
            cur_key = key;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long s = 2 * ((long)img * CIN + c0 + 8 * g + e);
                muA[e] = a.tfA ? a.tfA[s] : 0.f;
                rsA[e] = (a.tfA ? a.tfA[s + 1] : 1.f) * 2048.0f;
                muB[e] = (DUAL && a.tfB) ? a.tfB[s] : 0.f;
                rsB[e] = ((DUAL && a.tfB) ? a.tfB[s + 1] : 1.f) * 2048.0f;
            }
        }
        char* buf = lds + (u & 1) * C::BUF;
        prepare(u + 1);
        const bool interior = is_interior(ty0, tx0) && ty0 + C::TH <= a.ho && tx0 + 32 <= a.wo;
        if (interior) {
            if (bplain) commit_items(std::false_type{}, std::true_type{}, img, ty0, tx0, c0, buf);
            else commit_items(std::false_type{}, std::false_type{}, img, ty0, tx0, c0, buf);
        } else {
            commit_items(std::true_type{}, std::false_type{}, img, ty0, tx0, c0, buf);
        }
    };
    auto finalize = [&](int u) {                                                 // statistics record of the tile whose last unit was u
        if (!a.part || (u % C::NCH) != C::NCH - 1 || tid >= COUT) return;
        int img, tile, ty0, tx0, c0;
        geom(u, img, tile, ty0, tx0, c0);
        const float* r = red + ((u / C::NCH) & 1) * (C::WM * COUT * 2);
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int m = 0; m < C::WM; ++m) {
            s += r[(m * COUT + tid) * 2 + 0];
            q += r[(m * COUT + tid) * 2 + 1];
        }
        float* dst = a.part + (((long)img * a.nblk + tile) * COUT + tid) * 2;
        dst[0] = s;
        dst[1] = q;
    };

#if PC_TRACE
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif
    if (U > 0) {
        prepare(0);
#pragma unroll
        for (int i = 0; i < C::ITEMS; ++i) issue_item(i);
    }
    for (int u = 0; u < U; ++u) {
        PC_T(0);
        commit(u);                                                               // (per item: wait for its loads, transform, LDS, request the next unit's)
        PC_T(1);
        PC_T(2);
        pc_barrier();                                                            // #u: buffer u & 1 is complete
        PC_T(3);
        if (u >= 1) finalize(u - 1);
        PC_T(4);
    }
    pc_barrier();                                                                // #U: the consumers' last epilogue is done
    if (U >= 1) finalize(U - 1);
#if PC_TRACE
    if ((tid & 63) == 0 && a.out2) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.out2) + ((long)blockIdx.x * 8 + (tid >> 6)) * 8;
        for (int k = 0; k < 5; ++k) o[k] = tsum[k];
        o[6] = __builtin_readcyclecounter() - tstart;
        o[7] = (unsigned long long)U;
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------ consumers
template <int CIN, int COUT, int STRIDE, int TAPS, int EPI, bool F6>
__device__ __forceinline__ void pc_consumer(const PcArgs& a, const char* __restrict__ lds, float* __restrict__ red, float* __restrict__ patch, int cw,
                                            int lane) {
    using C = PcCfg<CIN, COUT, STRIDE, TAPS>;
    const int li = lane & 31, kg = lane >> 5;
    const int wn = cw % C::WN, wm = cw / C::WN;
    const int G = (int)gridDim.x;
    const int woff = pc_work_offset();
    const int ntile = max(0, (a.total_tiles - woff + G - 1) / G);
    // this wave's slice of the packed weights: [chunk][tap][ntile][k16-step][hi|lo][lane][8].  Buffer loads with the step's
    // offset in an SGPR: with flat addresses hipcc hoists one 64-bit pointer per step out of the tile loop and spills them
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wpk), 0, C::NCH * TAPS * C::NT * 4096, 0x00020000);
    const int wvoff = lane * 16, wsoff = wn * 4096;
    auto ldw = [&](int chtap, int ks, int hl) -> half8 {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff + chtap * (C::NT * 4096) + ks * 2048 + hl * 1024, 0);
        return __builtin_bit_cast(half8, v);
    };
    half8 wres[C::WRES ? C::NCH * C::NS : 1][2];
    if (C::WRES) {
#pragma unroll
        for (int s = 0; s < C::NCH * C::NS; ++s) {
            wres[s][0] = ldw(s >> 1, s & 1, 0);
            wres[s][1] = ldw(s >> 1, s & 1, 1);
        }
    }
    const float bias = a.bias ? a.bias[wn * 32 + li] : 0.f;
    // byte offset of this lane's fragment for output row 0 of the wave, tap (0, 0), k16-step 0
    int abase;
    if (TAPS == 9 && STRIDE == 2) abase = ((wm * C::RPW * 2) * C::HW + li) * PC_AS + kg * 16;
    else abase = ((wm * C::RPW) * C::HW + li) * PC_AS + kg * 16;
    auto aoff = [&](int m, int s) -> int {                                       // compile-time offset of (row m, step s) from abase
        const int tap = s >> 1, ks = s & 1, dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
        int pix;
        if (TAPS == 9 && STRIDE == 2) pix = (2 * m + dy) * C::HW + (dx & 1) * ((C::HW + 1) / 2) + (dx >> 1);
        else pix = (m + dy) * C::HW + dx;
        return pix * PC_AS + ks * 32;
    };
#if PC_TRACE
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif
    constexpr bool ROLL = C::RPW >= 2;          // fragments rolled in place one step ahead (RPW = 1: triple-buffered two steps ahead)
    constexpr int NBUF = ROLL ? 1 : 3;
    static_assert(!F6 || ROLL || C::WRES, "FP6 form at one m-tile per wave: resident weights only");
    // FP6 form: both correction terms of a tap (two k16-steps) in one e2m3 MFMA; the block's dwords 0-3 / 4-5 | scale sit where the lo halves of
    // step 0 / step 1 were - in LDS and in the packed weights alike - so `a0` / `b0` are a step-0 "lo" piece and `a1` / `b1` a step-1 one
    auto mfma6 = [&](pc_u32x4 a0, pc_u32x4 a1, half8 b0h, half8 b1h, floatx16 c) -> floatx16 {
        const pc_u32x4 b0 = __builtin_bit_cast(pc_u32x4, b0h), b1 = __builtin_bit_cast(pc_u32x4, b1h);
        const pc_intx8 A0 = (pc_intx8){(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, 0, 0};
        const pc_intx8 B0 = (pc_intx8){(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, 0, 0};
        const pc_intx8 A = __builtin_shufflevector(A0, A0, 0, 1, 2, 3, 4, 5, -1, -1), B = __builtin_shufflevector(B0, B0, 0, 1, 2, 3, 4, 5, -1, -1);
#if PC_ABL & 1
        asm volatile("" :: "v"(A), "v"(B));
        return c;
#else
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 2, 2, 0, (int)a1.z, 0, (int)b1.z);
#endif
    };

    for (int j = 0; j < ntile; ++j) {
        int img, tile, ty_, tx_;
        pc_work_tile(a, woff + j * G, img, tile, ty_, tx_);
        const int ty0 = ty_ * C::TH + wm * C::RPW, tx0 = tx_ * 32;
        floatx16 accm[C::RPW], accl[C::RPW];
#pragma unroll
        for (int m = 0; m < C::RPW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accm[m][r] = F6 ? 0.f : bias;       // (FP6 form: the bias joins in the epilogue - a splat of it would hold 16 registers through the tile loop)
                accl[m][r] = 0.f;
            }
#pragma unroll
        for (int ch = 0; ch < C::NCH; ++ch) {
            const int u = j * C::NCH + ch;
            half8 fa[NBUF][C::RPW][2];
            constexpr int WD = 4, WR = WD + 1;                                   // streamed weights: requested WD steps ahead (ring of WR)
            half8 wst[WR][2];
            const char* A = lds + (u & 1) * C::BUF + abase;
            auto loadA1 = [&](int b, int m, int s, int hl) { fa[b][m][hl] = *reinterpret_cast<const half8*>(A + aoff(m, s) + 64 * hl); };
            auto loadW = [&](int b, int s) {
                wst[b][0] = ldw(ch * TAPS + (s >> 1), s & 1, 0);
                wst[b][1] = ldw(ch * TAPS + (s >> 1), s & 1, 1);
            };
            if (!C::WRES && !F6) {                                               // (requested in front of the barrier: L2 latency under the wait)
#pragma unroll
                for (int d = 0; d < WD; ++d) loadW(d, d);
            }
            PC_T(0);
            pc_barrier();                                                        // #u: the producers have filled buffer u & 1
            PC_T(1);
            if constexpr (F6 && ROLL) {
                // a tap's MFMAs in three groups (main term of k16-step 0, of step 1, the FP6 correction of both); the four fragment sets of a
                // row - hi of both steps, the block's two 16-byte pieces - are re-requested for the NEXT tap right behind their last use:
                // every request 2 RPW .. 3 RPW MFMAs ahead of its first use.  One accumulator per m-tile (the weights' scale byte carries 2^-11).
                constexpr int WD6 = 4, WR6 = 6;                                  // streamed weights: steps requested 2 taps ahead, ring of 6 steps
                half8 w6[C::WRES ? 1 : WR6][2];
                half8 fh[C::RPW][2];
                pc_u32x4 fq[C::RPW][2];
                auto loadA6 = [&](int m, int tap) {
                    fh[m][0] = *reinterpret_cast<const half8*>(A + aoff(m, 2 * tap));
                    fh[m][1] = *reinterpret_cast<const half8*>(A + aoff(m, 2 * tap + 1));
                };
                auto loadQ6 = [&](int m, int tap) {
                    fq[m][0] = *reinterpret_cast<const pc_u32x4*>(A + aoff(m, 2 * tap) + 64);
                    fq[m][1] = *reinterpret_cast<const pc_u32x4*>(A + aoff(m, 2 * tap + 1) + 64);
                };
                if (!C::WRES) {
#pragma unroll
                    for (int d = 0; d < WD6; ++d) {
                        w6[d][0] = ldw(ch * TAPS + (d >> 1), d & 1, 0);
                        w6[d][1] = ldw(ch * TAPS + (d >> 1), d & 1, 1);
                    }
                }
#pragma unroll
                for (int m = 0; m < C::RPW; ++m) {
                    loadA6(m, 0);
                    loadQ6(m, 0);
                }
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int s0 = 2 * t, s1 = 2 * t + 1;
                    if (!C::WRES) {
#pragma unroll
                        for (int d = 0; d < 2; ++d)
                            if (s0 + WD6 + d < C::NS) {
                                w6[(s0 + WD6 + d) % WR6][0] = ldw(ch * TAPS + ((s0 + WD6 + d) >> 1), (s0 + WD6 + d) & 1, 0);
                                w6[(s0 + WD6 + d) % WR6][1] = ldw(ch * TAPS + ((s0 + WD6 + d) >> 1), (s0 + WD6 + d) & 1, 1);
                            }
                    }
                    const half8 wh0 = C::WRES ? wres[ch * C::NS + s0][0] : w6[s0 % WR6][0], wh1 = C::WRES ? wres[ch * C::NS + s1][0] : w6[s1 % WR6][0];
                    const half8 wq0 = C::WRES ? wres[ch * C::NS + s0][1] : w6[s0 % WR6][1], wq1 = C::WRES ? wres[ch * C::NS + s1][1] : w6[s1 % WR6][1];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < C::RPW; ++m) accm[m] = PC_MFMA(fh[m][0], wh0, accm[m]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < C::RPW; ++m) {
                        accm[m] = PC_MFMA(fh[m][1], wh1, accm[m]);
                        if (t + 1 < TAPS) loadA6(m, t + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int m = 0; m < C::RPW; ++m) {
                        accm[m] = mfma6(fq[m][0], fq[m][1], wq0, wq1, accm[m]);
                        if (t + 1 < TAPS) loadQ6(m, t + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if constexpr (F6) {
                // one m-tile per wave (the stride-2 3x3 layer): fragments of four steps in a ring, requested two steps ahead; the correction
                // of a tap follows the main term of its second step into a SECOND accumulator (a single one would chain all three MFMAs)
                half8 fh[4];
                pc_u32x4 fq[4];
                auto load6 = [&](int b, int s) {
                    fh[b] = *reinterpret_cast<const half8*>(A + aoff(0, s));
                    fq[b] = *reinterpret_cast<const pc_u32x4*>(A + aoff(0, s) + 64);
                };
                load6(0, 0);
                load6(1, 1);
#pragma unroll
                for (int s = 0; s < C::NS; ++s) {
                    if (s + 2 < C::NS) load6((s + 2) % 4, s + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    accm[0] = PC_MFMA(fh[s % 4], wres[ch * C::NS + s][0], accm[0]);
                    if (s & 1) accl[0] = mfma6(fq[(s - 1) % 4], fq[s % 4], wres[ch * C::NS + s - 1][1], wres[ch * C::NS + s][1], accl[0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (ROLL) {
                // a step's MFMAs in three groups (main term, hi x lo, lo x hi); a row's hi fragment is re-requested for the next step
                // right behind its last use in group 2, its lo fragment behind group 3: one fragment set in registers, every request
                // RPW .. 3 RPW MFMAs (>= 128 cycles at RPW = 2) ahead of its first use
#pragma unroll
                for (int m = 0; m < C::RPW; ++m) {
                    loadA1(0, m, 0, 0);
                    loadA1(0, m, 0, 1);
                }
#pragma unroll
                for (int s = 0; s < C::NS; ++s) {
                    if (!C::WRES && s + WD < C::NS) loadW((s + WD) % WR, s + WD);
                    const half8 bh = C::WRES ? wres[ch * C::NS + s][0] : wst[s % WR][0];
                    const half8 bl = C::WRES ? wres[ch * C::NS + s][1] : wst[s % WR][1];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < C::RPW; ++m) accm[m] = PC_MFMA(fa[0][m][0], bh, accm[m]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < C::RPW; ++m) {
                        accl[m] = PC_MFMA_C2(fa[0][m][0], bl, accl[m], s);
                        if (s + 1 < C::NS) loadA1(0, m, s + 1, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int m = 0; m < C::RPW; ++m) {
                        accl[m] = PC_MFMA_C3(fa[0][m][1], bh, accl[m], s);
                        if (s + 1 < C::NS) loadA1(0, m, s + 1, 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    loadA1(d, 0, d, 0);
                    loadA1(d, 0, d, 1);
                }
#pragma unroll
                for (int s = 0; s < C::NS; ++s) {
                    if (s + 2 < C::NS) {
                        loadA1((s + 2) % 3, 0, s + 2, 0);
                        loadA1((s + 2) % 3, 0, s + 2, 1);
                    }
                    if (!C::WRES && s + WD < C::NS) loadW((s + WD) % WR, s + WD);
                    __builtin_amdgcn_sched_barrier(0);                           // the requests stay in front of this step's MFMAs
                    const int b = s % 3;
                    const half8 bh = C::WRES ? wres[ch * C::NS + s][0] : wst[s % WR][0];
                    const half8 bl = C::WRES ? wres[ch * C::NS + s][1] : wst[s % WR][1];
                    accm[0] = PC_MFMA(fa[b][0][0], bh, accm[0]);
                    accl[0] = PC_MFMA_C2(fa[b][0][0], bl, accl[0], s);
                    accl[0] = PC_MFMA_C3(fa[b][0][1], bh, accl[0], s);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        PC_T(2);
        // ---- epilogue: the lane holds channel co of pixels x = tx0 + (r & 3) + 8 (r >> 2) + 4 kg of each of its rows.  The values
        // leave through a wave-private LDS transpose as 16-byte stores: with one dword store per accumulator register (256 B per
        // wave instruction) the consumers issued 128 stores per tile and CU and every one of them waited ~100 cycles for a slot in
        // the CU's memory pipeline behind the producers' halo loads (3.2 k of a tile's 7.6 k cycles by the trace); the pipeline is
        // bound by instructions in flight, not by bytes.  Partial tiles take the guarded copy (uniform branch: with per-store bounds
        // tests in the common path hipcc turns every store into its own exec-mask branch).
        const int co = wn * 32 + li;
        float ssum = 0.f, ssq = 0.f;
        bool fsat = false;
        // (one lane-dependent 32-bit offset per access pattern; everything else is uniform or immediate: the kernel has no registers to spare)
        float* Et_w = patch + cw * (32 * 36) + 4 * kg * 36 + li;                 // + ((r & 3) + 8 (r >> 2)) * 36
        const float* Et_r = patch + cw * (32 * 36) + (lane >> 3) * 36 + 4 * (lane & 7);   // + 8 jj * 36
        auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int m = 0; m < C::RPW; ++m) {
                const int gy = ty0 + m;
                float* o;                                                        // pixel tx0, this wave's 32 channels (uniform)
                int pstride;
                if (EPI == PC_EPI_RAW) {
                    o = a.out + (((long)img * a.ho + gy) * a.wo + tx0) * COUT + wn * 32;
                    pstride = COUT;
                } else if (EPI == PC_EPI_FMAP) {
                    const int wob = a.wo + 2 * a.out_border;
                    o = a.out + (((long)img * (a.ho + 2 * a.out_border) + gy + a.out_border) * wob + tx0 + a.out_border) * COUT + wn * 32;
                    pstride = COUT;
                } else if (EPI == PC_EPI_FSPLIT) {
                    o = nullptr;                                                 // (8-byte plane stores below)
                    pstride = 0;
                } else {
                    constexpr int HALF = COUT / 2, HW_ = C::WN / 2;              // waves 0 .. WN/2-1: tanh -> out, the rest: relu -> out2
                    o = (wn < HW_ ? a.out : a.out2) + (((long)img * a.ho + gy) * a.wo + tx0) * HALF + (wn % HW_) * 32;
                    pstride = HALF;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pxc = (r & 3) + 8 * (r >> 2);
                    float v = F6 ? (ROLL ? accm[m][r] + bias : (accm[m][r] + accl[m][r]) + bias) : fmaf(accl[m][r], 1.0f / 2048.0f, accm[m][r]);
                    if (EPI == PC_EPI_RAW) {
                        if (FULL || (gy < a.ho && tx0 + pxc + 4 * kg < a.wo)) {
                            ssum += v;
                            ssq = fmaf(v, v, ssq);
                        }
                    } else if (EPI == PC_EPI_FMAP) {
                        v *= a.out_scale;
                    } else if (EPI == PC_EPI_FSPLIT) {
                        v *= a.out_scale * 64.0f;                                // x * 2^CL_LOG2S of cost_lines.hip (out_scale = 1/8: exact)
                    } else {
                        v = (wn < C::WN / 2) ? tanhf(v) : fmaxf(v, 0.f);
                    }
                    Et_w[pxc * 36] = v;
                }
                const int lane_off = (lane >> 3) * pstride + 4 * (lane & 7);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 v4 = *reinterpret_cast<const float4*>(Et_r + 8 * jj * 36);       // (LDS operations of a wave execute in order)
                    if (EPI == PC_EPI_FSPLIT) {
                        // planes p = hl * 4 + ks of block `img`: halves at ((img * 8 + p) * bt + t) * 16 + c; this lane holds channels
                        // wn * 32 + 4 g4 + 0..3 of pixel tx0 + px: ks = 2 wn + (g4 >> 2), c = 4 (g4 & 3) .. + 3 -> 8 bytes per plane
                        const int px = (lane >> 3) + 8 * jj, g4 = lane & 7;
                        if (FULL || (gy < a.ho && tx0 + px < a.wo)) {
                            cer_f2 x0 = (cer_f2){v4.x, v4.y}, x1 = (cer_f2){v4.z, v4.w};
                            if (!(fabsf(v4.x) <= 65504.0f) || !(fabsf(v4.y) <= 65504.0f) || !(fabsf(v4.z) <= 65504.0f) || !(fabsf(v4.w) <= 65504.0f)) fsat = true;
                            x0 = __builtin_elementwise_min(__builtin_elementwise_max(x0, (cer_f2){-65504.0f, -65504.0f}), (cer_f2){65504.0f, 65504.0f});
                            x1 = __builtin_elementwise_min(__builtin_elementwise_max(x1, (cer_f2){-65504.0f, -65504.0f}), (cer_f2){65504.0f, 65504.0f});
                            const cer_h2 h0 = __builtin_convertvector(x0, cer_h2), h1 = __builtin_convertvector(x1, cer_h2);
                            const cer_h2 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, cer_f2), cer_h2);
                            const cer_h2 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, cer_f2), cer_h2);
                            typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                            // uniform row base + one 32-bit lane offset (the kernel has no registers to spare)
                            const int wob = a.wo + 2 * a.out_border, bt16 = (a.ho + 2 * a.out_border) * wob * 16;
                            _Float16* rowb = reinterpret_cast<_Float16*>(a.out) + ((long)img * 8 + 2 * wn) * bt16 +
                                             ((long)(gy + a.out_border) * wob + tx0 + a.out_border) * 16;
                            const int loff = ((lane >> 3) + 8 * jj) * 16 + 4 * (g4 & 3) + (g4 >> 2) * bt16;
                            *reinterpret_cast<half4_t*>(rowb + loff) = (half4_t){h0.x, h0.y, h1.x, h1.y};
                            *reinterpret_cast<half4_t*>(rowb + loff + 4 * (long)bt16) = (half4_t){l0.x, l0.y, l1.x, l1.y};
                        }
                    } else if (PC_ABL & 2) asm volatile("" :: "v"(v4.x), "v"(v4.y), "v"(v4.z), "v"(v4.w));
                    else if ((!PC_EXP24 || EPI != PC_EPI_RAW || jj != 3) && (FULL || (gy < a.ho && tx0 + (lane >> 3) + 8 * jj < a.wo))) *reinterpret_cast<float4*>(o + lane_off + 8 * jj * pstride) = v4;
                }
            }
        };
        if (ty0 + C::RPW <= a.ho && tx0 + 32 <= a.wo) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (EPI == PC_EPI_FSPLIT && a.out2 && __ballot(fsat) != 0ull && lane == 0)          // sticky: a feature beyond +-1023 was clamped (bit 1)
            atomicOr(reinterpret_cast<int*>(a.out2), 1);
        if (EPI == PC_EPI_RAW && a.part) {
            const float s2 = ssum + __shfl_xor(ssum, 32), q2 = ssq + __shfl_xor(ssq, 32);
            if (kg == 0) {
                float* r = red + (j & 1) * (C::WM * COUT * 2) + (wm * COUT + co) * 2;
                r[0] = s2;
                r[1] = q2;
            }
        }
        PC_T(3);
    }
    pc_barrier();                                                                // #U
#if PC_TRACE
    if (lane == 0 && a.out2 && EPI == PC_EPI_RAW) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.out2) + ((long)blockIdx.x * 8 + 4 + cw) * 8;
        for (int k = 0; k < 4; ++k) o[k] = tsum[k];
        o[6] = __builtin_readcyclecounter() - tstart;
        o[7] = (unsigned long long)ntile;
    }
#endif
}

template <int CIN, int COUT, int STRIDE, int TAPS, int EPI, bool DUAL, bool F6 = false>
__global__ __launch_bounds__(512, 2) void enc_pc_kernel(const PcArgs a) {
    using C = PcCfg<CIN, COUT, STRIDE, TAPS>;
    extern __shared__ __attribute__((aligned(16))) char pc_smem[];
    float* red = reinterpret_cast<float*>(pc_smem + 2 * C::BUF);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);           // wave index in an SGPR: the role branch is scalar
    if (wave < 4) pc_producer<CIN, COUT, STRIDE, TAPS, DUAL, F6>(a, pc_smem, red, (int)threadIdx.x);
    else pc_consumer<CIN, COUT, STRIDE, TAPS, EPI, F6>(a, pc_smem, red, reinterpret_cast<float*>(pc_smem + 2 * C::BUF + C::RED), wave - 4, (int)(threadIdx.x & 63));
}

// ------------------------------------------------------------------------------------------------------------------ host side
template <int CIN, int COUT, int STRIDE, int TAPS, int EPI>
static int pc_launch(PcArgs a, int nimg, hipStream_t st) {
    using C = PcCfg<CIN, COUT, STRIDE, TAPS>;
    a.tiles_x = (a.wo + 31) / 32;
    a.nblk = a.tiles_x * ((a.ho + C::TH - 1) / C::TH);
    const long total = (long)a.nblk * nimg;
    if (total >= (1L << 31)) return CER_ESHAPE;
    a.total_tiles = (int)total;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const int cus = cer_num_cus();                         // (persistent: one block per CU the launch may count on)
    const unsigned grid = (unsigned)(total < cus ? total : cus);
    const bool dual = a.srcB != nullptr, f6 = (a.flags & 8) != 0;                // flags & 8: FP6-correction form (weights from cer_enc_conv_pack_f6)
    const void* fn = f6 ? (dual ? (const void*)enc_pc_kernel<CIN, COUT, STRIDE, TAPS, EPI, true, true> : (const void*)enc_pc_kernel<CIN, COUT, STRIDE, TAPS, EPI, false, true>)
                        : (dual ? (const void*)enc_pc_kernel<CIN, COUT, STRIDE, TAPS, EPI, true> : (const void*)enc_pc_kernel<CIN, COUT, STRIDE, TAPS, EPI, false>);
    static bool raised[4][64];                                                   // per instantiation, per device (not "first device only")
    const int inst = (f6 ? 2 : 0) + (dual ? 1 : 0);
    if (C::SMEM > 64 * 1024 && (dev < 0 || dev >= 64 || !raised[inst][dev])) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 64) raised[inst][dev] = true;
    }
    if (f6 && dual) hipLaunchKernelGGL((enc_pc_kernel<CIN, COUT, STRIDE, TAPS, EPI, true, true>), dim3(grid), dim3(512), C::SMEM, st, a);
    else if (f6) hipLaunchKernelGGL((enc_pc_kernel<CIN, COUT, STRIDE, TAPS, EPI, false, true>), dim3(grid), dim3(512), C::SMEM, st, a);
    else if (dual) hipLaunchKernelGGL((enc_pc_kernel<CIN, COUT, STRIDE, TAPS, EPI, true>), dim3(grid), dim3(512), C::SMEM, st, a);
    else hipLaunchKernelGGL((enc_pc_kernel<CIN, COUT, STRIDE, TAPS, EPI, false>), dim3(grid), dim3(512), C::SMEM, st, a);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

static int pc_tile_rows(int Cout, int taps, int stride) { return (taps == 9 && stride == 2) ? 2 : (Cout >= 128 ? 4 : 8); }

extern "C" int cer_enc_pc_supported(int Cin, int Cout, int taps, int stride, int epi) {
    if (epi == PC_EPI_RAW) {
        if (Cin == 32 && Cout == 32 && taps == 9 && stride == 1) return 1;
        if (Cin == 32 && Cout == 64 && stride == 2 && (taps == 9 || taps == 1)) return 1;
        if (Cin == 64 && Cout == 64 && taps == 9 && stride == 1) return 1;
        return 0;
    }
    if (epi == PC_EPI_FMAP || epi == PC_EPI_FSPLIT) return Cin == 64 && Cout == 64 && taps == 1 && stride == 1;
    if (epi == PC_EPI_CTX) return Cin == 64 && Cout == 128 && taps == 1 && stride == 1;
    return 0;
}

// e2m3 (FP6: 1 sign, 2 exponent, 3 mantissa bits: 0, 0.125 .. 0.875, 1 .. 1.875, 2 .. 3.75, 4 .. 7.5), round to nearest even, saturating
static unsigned pc_e2m3(double v) {
    const unsigned sgn = v < 0 ? 32u : 0u;
    const double m = fabs(v);
    if (!(m == m) || m >= 7.5) return sgn | 31u;
    const int E = m < 2.0 ? 0 : (m < 4.0 ? 1 : 2);          // steps of 0.125 (subnormals and [1, 2)), 0.25, 0.5
    const int q = (int)nearbyint(ldexp(m, 3 - E));          // in units of the step: 0..16 (E = 0: the codes are linear in the value), 8..16
    return sgn | (unsigned)(E == 0 ? q : 8 * E + q);        // (q = 16 carries into the next exponent's first code)
}

// Weights of the FP6-correction form (round 6): cer_enc_conv_pack's order and size, [chunk32][tap][ntile32][k16-step][hi | q][lane][16 B]: the hi
// plane as before (f16 hi halves of channels 16 ks + 8 kg + 0..7 for output channel lane & 31); the q planes of a tap's two steps hold the
// lane's K block of v_mfma_scale_f32_32x32x64_f8f6f4 in e2m3: 32 six-bit fields, field i at bit 6 i = [wl' (8) | wh (8)] of step 0, then of step
// 1 (wl' = (w - wh) 2^11), divided by t = 2^(e - 2), e the exponent of the block's largest magnitude (one up where that would land above 7.75):
// dwords 0-3 in step 0's q plane, dwords 4-5 | E8M0 byte of t 2^-11 | 0 in step 1's.
extern "C" int cer_enc_conv_pack_f6(const float* w, void* packed_v, int Cout, int Cin, int taps) {
    if (!w || !packed_v) return CER_EINVAL;
    if (Cout % 32 || Cin % 32 || (taps != 1 && taps != 9)) return CER_ESHAPE;
    _Float16* packed = (_Float16*)packed_v;
    const int NT = Cout / 32;
    for (int kc = 0; kc < Cin / 32; ++kc)
        for (int tap = 0; tap < taps; ++tap)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = nt * 32 + (lane & 31);
                    const long base = (((long)kc * taps + tap) * NT + nt) * 4;              // 512-half planes: (ks 0: hi, q), (ks 1: hi, q)
                    double f[32], mx = 0.0;
                    for (int ks = 0; ks < 2; ++ks)
                        for (int e = 0; e < 8; ++e) {
                            const int ci = kc * 32 + ks * 16 + (lane >> 5) * 8 + e;
                            float v = w[((long)co * Cin + ci) * taps + tap];
                            v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
                            packed[(base + 2 * ks) * 512 + lane * 8 + e] = hi;
                            f[ks * 16 + e] = (double)(float)lo;
                            f[ks * 16 + 8 + e] = (double)(float)hi;
                            mx = fmax(mx, fmax(fabs(f[ks * 16 + e]), fabs(f[ks * 16 + 8 + e])));
                        }
                    int te = -100;                                                            // t = 2^te
                    if (mx > 0.0) {
                        int e2;
                        frexp(mx, &e2);                                                       // mx in [2^(e2-1), 2^e2)
                        te = e2 - 1 - 2;
                        if (ldexp(mx, -te) > 7.75) ++te;
                        if (te < -100) te = -100;
                    }
                    unsigned q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int i = 0; i < 32; ++i) {
                        const unsigned fld = pc_e2m3(ldexp(f[i], -te));
                        const int bit = 6 * i;
                        q[bit >> 5] |= fld << (bit & 31);
                        if ((bit & 31) > 26) q[(bit >> 5) + 1] |= fld >> (32 - (bit & 31));
                    }
                    q[6] = (unsigned)(te - 11 + 127);
                    q[7] = 0;
                    memcpy(reinterpret_cast<char*>(packed + (base + 1) * 512) + lane * 16, q, 16);
                    memcpy(reinterpret_cast<char*>(packed + (base + 3) * 512) + lane * 16, q + 4, 16);
                }
    return CER_OK;
}

extern "C" int cer_enc_pc_tiles(int ho, int wo, int Cout, int taps, int stride) {
    const int th = pc_tile_rows(Cout, taps, stride);
    return ((wo + 31) / 32) * ((ho + th - 1) / th);
}

extern "C" int cer_enc_pc_conv(const float* srcA, const float* statsA, const float* srcB, const float* statsB, int flags, float* merged_out,
                               const void* packed_w, const float* bias, float* out, float* out2, float* stats_partial, int N, int h, int w,
                               int Cin, int Cout, int taps, int stride, int epi, int out_border, float out_scale, void* stream) {
    if (!srcA || !packed_w || !out || N <= 0 || h <= 0 || w <= 0) return CER_EINVAL;
    if (!cer_enc_pc_supported(Cin, Cout, taps, stride, epi)) return CER_ESHAPE;
    if (epi == PC_EPI_CTX && !out2) return CER_EINVAL;
    if (merged_out && (!srcB || stride != 1)) return CER_EINVAL;
    if (!cer_aligned16(srcA) || !cer_aligned16(packed_w) || (srcB && !cer_aligned16(srcB)) || (merged_out && !cer_aligned16(merged_out)))
        return CER_EALIGN;
    PcArgs a;
    memset(&a, 0, sizeof(a));
    a.srcA = srcA;
    a.tfA = statsA;
    a.srcB = srcB;
    a.tfB = srcB ? statsB : nullptr;
    a.flags = flags;
    a.mout = merged_out;
    a.wpk = (const _Float16*)packed_w;
    a.bias = bias;
    a.out = out;
    a.out2 = out2;
    a.part = epi == PC_EPI_RAW ? stats_partial : nullptr;
    a.h = h;
    a.w = w;
    const int pad = taps == 9 ? 1 : 0, ks = taps == 9 ? 3 : 1;
    a.ho = (h + 2 * pad - ks) / stride + 1;
    a.wo = (w + 2 * pad - ks) / stride + 1;
    a.out_border = out_border;
    a.out_scale = out_scale;
    hipStream_t st = (hipStream_t)stream;
    if (epi == PC_EPI_FMAP) return pc_launch<64, 64, 1, 1, PC_EPI_FMAP>(a, N, st);
    if (epi == PC_EPI_FSPLIT) return pc_launch<64, 64, 1, 1, PC_EPI_FSPLIT>(a, N, st);
    if (epi == PC_EPI_CTX) return pc_launch<64, 128, 1, 1, PC_EPI_CTX>(a, N, st);
    if (Cin == 32 && Cout == 32) return pc_launch<32, 32, 1, 9, PC_EPI_RAW>(a, N, st);
    if (Cin == 32 && taps == 9) return pc_launch<32, 64, 2, 9, PC_EPI_RAW>(a, N, st);
    if (Cin == 32) return pc_launch<32, 64, 2, 1, PC_EPI_RAW>(a, N, st);
    return pc_launch<64, 64, 1, 9, PC_EPI_RAW>(a, N, st);
}
