// K3, round 2: the update block's 3x3 convolutions (reference: core/update.py:13-25,61-71,80-85,87-120) as barrier-light
// implicit GEMMs on v_mfma_f32_32x32x16_f16 with fp32-class accuracy, ONE accumulator per output tile.
//
// Arithmetic ("s16": split-f16, single accumulator).  Every fp32 operand x is carried as two halves of  xs = x * 2^k  (k: a
// per-tensor power of two, exact):  hi = f16(xs),  lo = f16(xs - hi)  (UNSCALED residual; |xs - hi - lo| <= max(2^-22 |xs|,
// 2^-25)), and  x*w ~= (xh*wh + xh*wl + xl*wh) / (2^kx 2^kw):  three f16 MFMAs into the SAME fp32 accumulator (the dropped
// xl*wl term is 2^-22 relative).  Measured on the MI355X (tools/ubench/mfma_merge_acc.hip, K = 1248): error / sum|x||w| rms
// 1.6e-8 against 2.7e-8 for an fp32 fmaf chain and 1.0e-8 for round 1's two-accumulator form - the MFMA adder shows no
// truncation bias, so the merged form is fp32-class.  It halves the accumulator registers, which is what makes the structure
// below possible.  All sources of one conv share the product scale S = 2^kx(src) * 2^kw(src) (cer_conv3x3_s16_scale).
//
// Activation layout "split16": a [P, C] tensor keeps, per pixel and 16-channel group, 16 hi halves | 16 lo halves (64 bytes) in
// the bytes of the fp32 slots - written by the producers' epilogues, staged by the consumers with plain 16-byte copies.
//
// Structure (block = 4 waves, 2 blocks per CU = 2 waves per SIMD, 256 VGPRs per wave):
//   * tile = TH x 16 pixels; wave (wm, wn) owns MT m-tiles of 32 pixels (2 rows x 16) x 32 output channels: MT accumulators;
//   * MFMA orientation: A = weights (rows = channels), B = activations (columns = pixels), so a lane ends up with channels
//     8j + 4kg + 0..3 of ONE pixel: after one v_permlane32_swap per register pair a lane owns 8 consecutive channels - the
//     epilogue loads / stores 16-byte vectors straight from the accumulators (no LDS transpose), and the delta head's
//     projection takes them as B fragments directly;
//   * weights never touch LDS: each wave loads ITS 32-channel slice of a (half-chunk, tap) step - 2 x 1 KiB, lane-linear in
//     fragment order - straight into registers two steps ahead; with the 1 x 4 wave layout no two waves of a block load the
//     same bytes, so there is no weight ring, no DMA and NO per-step barrier;
//   * activations: per 16-channel half-chunk the halo tile ((TH+2) x 18 pixels x 64 B, XOR-swizzled 16-byte slots: conflict
//     -free ds_read_b128 for every tap) is loaded to registers at the first tap of the previous half-chunk and written to the
//     other LDS buffer five taps later; ONE barrier per half-chunk (9 taps), placed before the last tap so that the operand
//     prefetch of the next half-chunk's first tap is already legal; fragments are double-buffered in registers one step ahead;
//   * the disparity encoder's 49 channels are generated from an LDS disparity tile: interior tiles use the collapsed 81-tap
//     form (6 single-tap steps), border tiles the literal one (4 half-chunks x 9 taps) - same algebra as round 1;
//   * the hoisted `init` term / bias is the accumulators' initial value (16-byte loads in the prologue).
#include "conv_s16_shared.hpp"
#ifndef SX_MT8
#define SX_MT8 0
#endif

struct SxStage {                           // what to put into the NEXT activation buffer
    int kind;                              // 0 nothing, 2 tensor half-chunk, 1 literal disparity group, 3 collapsed disparity group
    const char* base;                      // tensor: source + 2048 * group
    long mtb;                              // tensor: bytes per m-tile (channels / 16 * 2048)
    int g;                                 // disparity group index
};

template <int WM_, int WN_, int MT, int EPI, int F8>
__global__ __launch_bounds__(256, (WM_ == 2 && F8 && SX_OCC3) ? 3 : (MT >= 8 ? 1 : 2)) void conv3x3_s16_kernel(const S16Args a) {
    constexpr int TH = WM_ * 2 * MT, HR = TH + 2;
    constexpr int ABUF = HR * (F8 ? SX_ROWB8 : SX_ROWB);
    constexpr int NPIX = HR * SX_HW, NITEM = NPIX * 4, ITEMS = (NITEM + 255) / 256;
    // F8 = 2 (round 6): the correction terms in FP6 (e2m3) with one E8M0 scale per K block - half the matrix-pipe passes of the fp8 form.
    constexpr bool F6 = F8 == 2;
    constexpr int NITEM8 = NPIX * 4, ITEMS8 = F8 ? (NITEM8 + 255) / 256 : 1;
    constexpr int NB = WN_ * 32;
    constexpr int DROWS = HR + 6;
    extern __shared__ __attribute__((aligned(16))) char sx_smem[];
    float* ldsD = reinterpret_cast<float*>(sx_smem + 2 * ABUF);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN_, wn = wave % WN_;
    const int li = lane & 31, kg = lane >> 5;
    CER_FUZZ_INIT();
#if SX_TRACE
    unsigned long long trace_t[24];
    int trace_n = 0;
    trace_t[trace_n++] = __builtin_readcyclecounter();
#define SX_STAMP() do { if (trace_n < 24) trace_t[trace_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define SX_STAMP() do { } while (0)
#endif

    // ---- block -> (tile, channel block): consecutive virtual ids stay on one XCD (blocks are dealt round-robin over the 8
    // XCDs), so neighbouring tiles - which share halo lines - and the channel blocks of one tile meet in the same L2
    auto xcd_order = [](int n, int j) {                    // j-th block of n (dealt round-robin) -> virtual id, contiguous per XCD
        const int q = n >> 3, r = n & 7, xcd = j & 7, k = j >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    };
    int tile, by;
    int tile_y, tile_x;
    if (a.border_first) {
        // convs with a disparity source: tiles on the image rim run the rim correction (left / right column + 24 % block life, top / bottom row
        // + 5 %), and in row-major order the slowest tiles of the last rows were the launch's tail.  Rim tiles go to the FIRST blocks launched
        // (columns, then rows; spread over the XCDs by the launch order itself), the interior follows in the XCD-contiguous order.  (ny = 1)
        const int bid = blockIdx.x, tys = a.ntiles / a.tiles_x, nbord = 2 * tys + 2 * (a.tiles_x - 2);
        if (bid < 2 * tys) {
            tile_y = bid >> 1; tile_x = (bid & 1) ? a.tiles_x - 1 : 0;
        } else if (bid < nbord) {
            const int r2 = bid - 2 * tys;
            tile_y = (r2 & 1) ? tys - 1 : 0; tile_x = 1 + (r2 >> 1);
        } else {
            const int i = xcd_order(a.ntiles - nbord, bid - nbord);
            tile_y = 1 + i / (a.tiles_x - 2); tile_x = 1 + i - (tile_y - 1) * (a.tiles_x - 2);
        }
        tile = tile_y * a.tiles_x + tile_x;
        by = 0;
    } else {
        const int vid = xcd_order((int)gridDim.x, (int)blockIdx.x);
        tile = vid / a.ny; by = vid - tile * a.ny;
        tile_y = tile / a.tiles_x; tile_x = tile - tile_y * a.tiles_x;
    }
    const int ty0 = tile_y * TH, tx0 = tile_x * SX_TW;
    const int nb0 = by * NB;
    const int NT = a.cout >> 5;

    // ---- sources: tensors, then (optionally) the disparity
    int ntens = a.nsrc;
    const float* dsrc = nullptr;
    if (a.kind[a.nsrc - 1] == 1) { ntens = a.nsrc - 1; dsrc = reinterpret_cast<const float*>(a.src[a.nsrc - 1]); }
    // collapsed disparity form: exact for pixels whose 3x3 neighbourhood lies inside the image - i.e. on interior tiles; with the
    // rim correction of the epilogue (a.edge) on every tile
    const bool interior = ty0 >= 1 && ty0 + TH <= a.h - 1 && tx0 >= 1 && tx0 + SX_TW <= a.w - 1;
    const bool coll = dsrc && a.wpk_c && (interior || a.edge);
    int nsteps = 0;
    for (int s = 0; s < ntens; ++s) nsteps += (a.ch[s] >> 4) * 9;
    const int nsteps_t = nsteps;           // steps of the tensor sources (F8: two of them form one 32-channel chunk step of 4 KiB)
    if (dsrc) nsteps += coll ? 6 : 36;
#if SX_STAGGER
    // Stagger (round 6).  The two workgroups a CU holds start together and run their phases in lock-step - prologue, chunk loop, epilogue of both at the
    // same time, the matrix pipe idle through two of the three (34 % busy over a z|r launch by PMC).  With several depth maps in flight the blocks of a
    // launch start whenever another kernel's block retires and the phases are mixed anyway; ONE depth map at a time they are not.  So in launches of more
    // than three blocks per CU pair the SECOND workgroup of a CU (HW_ID.TG_ID parity) of the first round of blocks sleeps for about a quarter of a chunk
    // loop before it starts: one block's memory phases then fall under the other's matrix phase.  Timing only - results cannot change.  Same-box A/B
    // (profiles/r06_stagger_ab.txt): + 1.7 % depth maps per second one at a time (four pairs, every pair), +- 0 with three in flight; longer sleeps lose.
    if ((!SX_STAGGER_WM1 || WM_ == 1) && (int)gridDim.x > 3 * SX_STAGGER_CUS && (int)blockIdx.x < 2 * SX_STAGGER_CUS) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        if ((hw >> SX_STAGGER_BIT) & 1)
            for (int i = 0; i < nsteps * SX_STAGGER; ++i) __builtin_amdgcn_s_sleep(1);      // 64 cycles each
    }
#endif
    const char* wlane = reinterpret_cast<const char*>(coll ? a.wpk_c : a.wpk) + ((long)(nb0 >> 5) + wn) * 2048 + lane * 16;
    const long wstep = (long)NT * 2048;    // bytes per step
    // F8: [chunk step][n-tile][f16 hi of half-chunk 0 | of half-chunk 1 | fp8 bytes 0-15 | fp8 bytes 16-31][lane][16 B]
    const char* wlane8 = reinterpret_cast<const char*>(coll ? a.wpk_c : a.wpk) + ((long)(nb0 >> 5) + wn) * 4096 + lane * 16;

    // ---- per-lane LDS addresses of the activation fragments: pixel (row 2*(wm*MT+m) + (li>>4) + dy, col (li&15) + dx) of the
    // halo tile; 16-byte slot kg (hi) / kg ^ 2 (lo), XOR-swizzled with ((col >> 2) & 3)
    int xh[3], xl[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int col = (li & 15) + dx;
        const int base = (2 * wm * MT + (li >> 4)) * SX_ROWB + col * 64;
        const int sl = kg ^ ((col >> 2) & 3);
        xh[dx] = base + sl * 16;
        xl[dx] = base + (sl ^ 2) * 16;
    }

    // ---- staging items of this thread: (halo pixel n, physical slot ps); the source piece is plane (hl, kg) = logical slot of
    // the pixel's m-tile: frag16 byte offset  mt * mtb + group * 2048 + logical * 512 + li * 16
    int st_pk[ITEMS];                      // m-tile | li << 20 | logical << 25 | valid << 27 | in-range << 28
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int it = tid + 256 * i;
        const int n = min(it >> 2, NPIX - 1), ps = it & 3;
        const int r = n / SX_HW, c = n - r * SX_HW;
        const int gy = ty0 + r - 1, gx = tx0 + c - 1;
        const bool valid = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
        const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
        const int logical = ps ^ ((c >> 2) & 3);
        st_pk[i] = ((cy >> 1) * a.mtx + (cx >> 4)) | ((((cy & 1) << 4) | (cx & 15)) << 20) | (logical << 25) | ((valid ? 1 : 0) << 27) |
                   ((it < NITEM ? 1 : 0) << 28);
    }
    // the items are staged in two batches (taps 0 -> 3 and 3 -> 6 of the previous group) through the same registers
    constexpr int IB = (ITEMS + 1) / 2;
    uint4 raw[IB];
    // `vary` = 0 at run time, but derived from the loop counter: everything computed from (tid ^ vary) is re-derived where it is
    // used instead of being hoisted out of the MFMA loop as a loop invariant (hipcc otherwise keeps ~100 such registers alive
    // across the loop and spills them)
    int vary = 0;
    auto stage_load = [&](const SxStage& st, int batch) {
#pragma unroll
        for (int k = 0; k < IB; ++k) {
            const int i = batch * IB + k;
            if (i >= ITEMS) break;
            const int pk = st_pk[i] ^ vary;
            const char* p = st.base + (long)(pk & 0xFFFFF) * st.mtb + (((pk >> 25) & 3) * 512 + ((pk >> 20) & 31) * 16);
            raw[k] = *reinterpret_cast<const uint4*>(p);
        }
    };
    auto stage_store = [&](int bufoff, int batch) {
        CER_FUZZ_POINT();
        const int tid_ = tid ^ vary;
#pragma unroll
        for (int k = 0; k < IB; ++k) {
            const int i = batch * IB + k;
            if (i >= ITEMS) break;
            const int pk = st_pk[i] ^ vary;
            const unsigned m = (pk & (1 << 27)) ? 0xFFFFFFFFu : 0u;            // zero padding of the feature map
            uint4 v = raw[k];
            v.x &= m; v.y &= m; v.z &= m; v.w &= m;
            const int it = tid_ + 256 * i;
            const int n = it >> 2, r = n / SX_HW, c = n - r * SX_HW;
            if (pk & (1 << 28)) *reinterpret_cast<uint4*>(sx_smem + bufoff + (r * SX_PITCH + c) * 64 + (it & 3) * 16) = v;
        }
    };
    // ---- fp8-correction form (F8): a 32-channel chunk per buffer, 128 B per halo pixel in eight 16-byte slots:
    //   2 kg + hc (0-3): f16 hi halves of the 8 channels (half-chunk hc, kg) - the B fragments of the main term;
    //   4 + 2 kg + hc:   8 fp8 bytes of those channels' hi halves * 2^-8 | 8 fp8 bytes of their lo halves * 2^3:
    //   slots 4 + 2 kg and 5 + 2 kg are the lane's 32-byte B operand of v_mfma_scale_f32_32x32x64_f8f6f4 (K = [xh | xl] of half-chunk 0,
    //   then of half-chunk 1; the weights carry [wl | wh] in the same positions).  Physical slot = logical ^ ((col >> 1) & 7): the 16
    //   pixels of a quarter wave land in 16 different 16-byte columns of the 256-byte bank line for every tap.  A lane's four
    //   fragments of a pixel are at xa ^ {0, 16, 64, 80}.
    int xa[3];
    if constexpr (F8) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int col = (li & 15) + dx;
            xa[dx] = (2 * wm * MT + (li >> 4)) * SX_ROWB8 + col * 128 + (((2 * kg) ^ ((col >> 1) & 7)) << 4);
        }
    }
    // staging items: (halo pixel n, hc, kg) = 8 channels: their hi and lo pieces (two 16-byte loads 1 KiB apart) become one f16 slot
    // (the hi piece as it is) and one fp8 slot (v_cvt_scalef32_pk_fp8_f16 divides by its scale operand: hi / 2^8 | lo / 2^-3)
    int st8_pk[ITEMS8];                    // m-tile | li << 20 | valid << 28 | in-range << 29
    if constexpr (F8) {
#pragma unroll
        for (int i = 0; i < ITEMS8; ++i) {
            const int it = tid + 256 * i;
            const int n = min(it >> 2, NPIX - 1);
            const int r = n / SX_HW, c = n - r * SX_HW;
            const int gy = ty0 + r - 1, gx = tx0 + c - 1;
            const bool valid = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
            const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
            st8_pk[i] = ((cy >> 1) * a.mtx + (cx >> 4)) | ((((cy & 1) << 4) | (cx & 15)) << 20) | ((valid ? 1 : 0) << 28) | ((it < NITEM8 ? 1 : 0) << 29);
        }
    }
    constexpr int IB8 = (ITEMS8 + 2) / 3;  // items per batch; three batches per chunk (taps 0 -> 2, 2 -> 4, 4 -> 6)
    constexpr int NLD8 = 2;                // 16-byte pieces per item: hi | lo (1 KiB apart)
    uint4 raw8[IB8][NLD8];
    auto stage8_addr = [&](const SxStage& st, int pk, int it) {
        return st.base + (long)(pk & 0xFFFFF) * st.mtb + (((it >> 1) & 1) * 2048 + (it & 1) * 512 + ((pk >> 20) & 31) * 16);
    };
    // FP6 form (same items as the fp8 form: (pixel, half-chunk hc, kg) = 8 channels, hi and lo piece).  The lane's B operand of the FP6
    // instruction lives in the pixel's slots 4 + 2 kg | 5 + 2 kg: 32 six-bit e2m3 fields, field i at bit 6 i = [xh (8) | xl * 2^11 (8)] of
    // half-chunk 0, then of half-chunk 1 (24 bytes), all divided by ONE power of two s = 2^(e - 2) - e: exponent of the largest |xh| of the 16
    // channels (|xl * 2^11| <= |xh| element by element, so the block maximum lands in [4, 7.75), or one exponent up in [3.875, 4]: nothing
    // saturates) - then the E8M0 byte of s in byte 24 (dword 6: the register the instruction takes this lane's scale from).  The two items of
    // a block sit in lanes tid and tid ^ 2: they exchange their maxima by DPP, convert their own 16 values (v_cvt_scalef32_pk32_fp6_f16
    // divides by its scale operand, rounds to nearest even and packs field i at bit 6 i - tools/ubench/mfma_fp6.hip; the upper 16 inputs are
    // don't-cares) and write their 96 bits: hc = 0 dwords 0-2 of the region, hc = 1 dwords 3-5 and the scale.
    auto stage6_put = [&](int bufoff, int pk, int it, uint4 vh, uint4 vl) {
        const unsigned m = (pk & (1 << 28)) ? 0xFFFFFFFFu : 0u;                // zero padding of the feature map
        vh.x &= m; vh.y &= m; vh.z &= m; vh.w &= m;
        vl.x &= m; vl.y &= m; vl.z &= m; vl.w &= m;
        const int n = it >> 2, r = (n * 3641) >> 16, c = n - r * SX_HW;         // n / 18 for n < 3641
        const int hc = (it >> 1) & 1, kgs = it & 1, key = (c >> 1) & 7;
        char* px = sx_smem + bufoff + (r * SX_PITCH + c) * 128;
        // exponent of the block's largest |xh|: f16 bit patterns without their sign order like the magnitudes (packed unsigned max)
        typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
        union U2 { unsigned u; ushort2_t s; };
        U2 a0, a1, a2, a3;
        a0.u = vh.x & 0x7FFF7FFFu; a1.u = vh.y & 0x7FFF7FFFu; a2.u = vh.z & 0x7FFF7FFFu; a3.u = vh.w & 0x7FFF7FFFu;
        a0.s = __builtin_elementwise_max(a0.s, a1.s);
        a2.s = __builtin_elementwise_max(a2.s, a3.s);
        a0.s = __builtin_elementwise_max(a0.s, a2.s);
        unsigned mm = max(a0.u & 0xFFFFu, a0.u >> 16);
        mm = max(mm, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mm, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]: the item of the other half-chunk
        const unsigned sb = ((mm + 0x40u) >> 10) + 110u;                        // E8M0 of s = 2^(bexp - 15 - 2); + 0x40: mantissas >= 1.9375 go one up
        const float sc = __uint_as_float(sb << 23);
        typedef _Float16 half32_t __attribute__((ext_vector_type(32)));
        typedef _Float16 half16_t __attribute__((ext_vector_type(16)));
        typedef int intx6_t __attribute__((ext_vector_type(6)));
        union { half16_t v; uint4 q[2]; cer_h2 h[8]; } in;
        in.q[0] = vh; in.q[1] = vl;
        const cer_h2 k2048 = (cer_h2){(_Float16)2048.0f, (_Float16)2048.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) in.h[4 + j] = in.h[4 + j] * k2048;
        const half32_t in32 = __builtin_shufflevector(in.v, in.v, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                                      -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
        const intx6_t f6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(in32, sc);
        if (pk & (1 << 29)) {
            *reinterpret_cast<uint4*>(px + (((2 * kgs + hc) ^ key) << 4)) = vh;
            char* s0 = px + (((4 + 2 * kgs) ^ key) << 4);
            char* s1 = px + (((5 + 2 * kgs) ^ key) << 4);
            *reinterpret_cast<unsigned*>(hc ? s0 + 12 : s0) = (unsigned)f6[0];
            *reinterpret_cast<unsigned*>(hc ? s1 : s0 + 4) = (unsigned)f6[1];
            *reinterpret_cast<unsigned*>(hc ? s1 + 4 : s0 + 8) = (unsigned)f6[2];
            *reinterpret_cast<unsigned*>(hc ? s1 + 8 : s1 + 12) = hc ? sb : 0u;
        }
    };
    auto stage8_put = [&](int bufoff, int pk, int it, uint4 vh, uint4 vl) {
        const unsigned m = (pk & (1 << 28)) ? 0xFFFFFFFFu : 0u;                // zero padding of the feature map
        vh.x &= m; vh.y &= m; vh.z &= m; vh.w &= m;
        vl.x &= m; vl.y &= m; vl.z &= m; vl.w &= m;
        const int n = it >> 2, r = (n * 3641) >> 16, c = n - r * SX_HW;         // n / 18 for n < 3641
        const int hc = (it >> 1) & 1, kgs = it & 1, key = (c >> 1) & 7;
        char* px = sx_smem + bufoff + (r * SX_PITCH + c) * 128;
        union { uint4 u; cer_h2 h[4]; } ch, cl;
        ch.u = vh; cl.u = vl;
        typedef short short2_t __attribute__((ext_vector_type(2)));
        union { uint4 u; short2_t s[4]; } q;
        q.s[0] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16((short2_t){0, 0}, ch.h[0], 256.0f, false);
        q.s[0] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q.s[0], ch.h[1], 256.0f, true);
        q.s[1] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16((short2_t){0, 0}, ch.h[2], 256.0f, false);
        q.s[1] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q.s[1], ch.h[3], 256.0f, true);
        q.s[2] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16((short2_t){0, 0}, cl.h[0], 0.125f, false);
        q.s[2] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q.s[2], cl.h[1], 0.125f, true);
        q.s[3] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16((short2_t){0, 0}, cl.h[2], 0.125f, false);
        q.s[3] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q.s[3], cl.h[3], 0.125f, true);
        if (pk & (1 << 29)) {
            *reinterpret_cast<uint4*>(px + (((2 * kgs + hc) ^ key) << 4)) = vh;
            *reinterpret_cast<uint4*>(px + (((4 + 2 * kgs + hc) ^ key) << 4)) = q.u;
        }
    };
    auto stage8_load = [&](const SxStage& st, int batch) {
#pragma unroll
        for (int k = 0; k < IB8; ++k) {
            const int i = batch * IB8 + k;
            if (i >= ITEMS8) break;
            const char* p = stage8_addr(st, st8_pk[i] ^ vary, (tid ^ vary) + 256 * i);
#pragma unroll
            for (int q = 0; q < NLD8; ++q) raw8[k][q] = *reinterpret_cast<const uint4*>(p + 1024 * q);
        }
    };
    auto stage8_store = [&](int bufoff, int batch) {
        CER_FUZZ_POINT();
#pragma unroll
        for (int k = 0; k < IB8; ++k) {
            const int i = batch * IB8 + k;
            if (i >= ITEMS8) break;
            if constexpr (F6) stage6_put(bufoff, st8_pk[i] ^ vary, (tid ^ vary) + 256 * i, raw8[k][0], raw8[k][1]);
            else stage8_put(bufoff, st8_pk[i] ^ vary, (tid ^ vary) + 256 * i, raw8[k][0], raw8[k][1]);
        }
    };
    // literal disparity features, group g (channels 16g .. 16g+15 of 100 * (unfold7x7(d) - d), core/update.py:80-85,97)
    auto gen_literal = [&](int g, int bufoff) {
        CER_FUZZ_POINT();
        for (int n = tid ^ vary; n < NPIX; n += 256) {
            const int r = n / SX_HW, c = n - r * SX_HW;
            const int gy = ty0 + r - 1, gx = tx0 + c - 1;
            const bool inside = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;    // the 3x3 conv zero-pads the FEATURE map
            const float ctr = ldsD[(r + 3) * SX_DTW + c + 3];
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int chn = 16 * g + k;
                const int uy = (chn * 37) >> 8, ux = chn - 7 * uy;              // chn / 7, chn % 7 for chn < 64
                v[k] = (inside && chn < 49) ? 100.0f * (ldsD[(r + uy) * SX_DTW + c + ux] - ctr) : 0.f;
            }
            char* px = sx_smem + bufoff + (r * SX_PITCH + c) * 64;
            const int key = (c >> 2) & 3;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const float u[8] = {v[8 * kh], v[8 * kh + 1], v[8 * kh + 2], v[8 * kh + 3], v[8 * kh + 4], v[8 * kh + 5], v[8 * kh + 6], v[8 * kh + 7]};
                half8 hi, lo;
                sx_split8(u, a.disp_scale, hi, lo);
                *reinterpret_cast<half8*>(px + ((kh ^ key) * 16)) = hi;
                SX_LDS_STORE_WAIT();
                *reinterpret_cast<half8*>(px + (((2 + kh) ^ key) * 16)) = lo;
                SX_LDS_STORE_WAIT();
            }
        }
    };
    // collapsed disparity features, group g: channel s = (sy, sx) of the 9x9 window holds 100 * (d[p + s - 4] - d[p]); only the
    // tile's own pixels (the centre tap) are read
    // (Waves 0 and 1 generate, as in round 2.  Spreading the two channel halves over the wave pairs - waves 0, 1: channels 0-7,
    // waves 2, 3: channels 8-15, a wave-uniform branch - produced intermittently wrong LAST tile rows on the MI355X in 74-99 % of the
    // launches, 0 of 400 with `s_nop 3` behind the generator's stores; cause not identified, DESIGN.md 3g.  This form: 0 of 1 600.)
    auto gen_collapsed = [&](int g, int bufoff) {
        // one half unit = (pixel, 8 of the group's 16 channels): the window offsets are compile-time constants and fold into the
        // ds_read offsets (the section was bound by the generators' address arithmetic: 11.0 k -> 7.3 k cycles for its six steps)
        auto half_unit = [&](auto kh_tag, int n) {
            constexpr int kh = decltype(kh_tag)::value;
            const int pr = n >> 4, pc = n & 15;
            const float* dp = ldsD + pr * SX_DTW + pc;
            const float ctr = dp[4 * SX_DTW + 4];
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int s = 16 * g + 8 * kh + k;
                const int sy = (s * 57) >> 9, sx = s - 9 * sy;                  // s / 9, s % 9 for s < 96
                v[k] = (s < 81) ? 100.0f * (dp[sy * SX_DTW + sx] - ctr) : 0.f;
            }
            const int c = pc + 1;
            char* px = sx_smem + bufoff + ((pr + 1) * SX_PITCH + c) * 64;
            const int key = (c >> 2) & 3;
            half8 hi, lo;
            sx_split8(v, a.disp_scale, hi, lo);
            // (wait states behind the 16-byte LDS stores: see the note above - the failing variant became clean with them)
            *reinterpret_cast<half8*>(px + ((kh ^ key) * 16)) = hi;
            SX_LDS_STORE_WAIT();
            *reinterpret_cast<half8*>(px + (((2 + kh) ^ key) * 16)) = lo;
            SX_LDS_STORE_WAIT();
        };
        CER_FUZZ_POINT();
        for (int u = tid ^ vary; u < TH * SX_TW; u += 256) {
            half_unit(std::integral_constant<int, 0>{}, u);
            half_unit(std::integral_constant<int, 1>{}, u);
        }
    };
    // staging schedule inside a 9-tap group: tap 0 load batch 0; tap 3 write batch 0, load batch 1; tap 6 write batch 1 (or
    // generate the disparity group); the barrier follows before tap 8
    auto stage_tap = [&](const SxStage& st, int bufoff, int t) {
        if (st.kind == 2) {
            if constexpr (F8) {
                if (t == 0) stage8_load(st, 0);
                if (t == 2) { stage8_store(bufoff, 0); stage8_load(st, 1); }
                if (t == 4) { stage8_store(bufoff, 1); stage8_load(st, 2); }
                if (t == 6) stage8_store(bufoff, 2);
            } else {
                if (t == 0) stage_load(st, 0);
                if (t == 3) { stage_store(bufoff, 0); stage_load(st, 1); }
                if (t == 6) stage_store(bufoff, 1);
            }
        } else if (t == 6) {
            if (st.kind == 1) gen_literal(st.g, bufoff);
            else if (st.kind == 3) gen_collapsed(st.g, bufoff);
        }
    };
    // LDS-only release / acquire around the barrier: the bare s_barrier builtin orders nothing for the compiler (a generator's last
    // ds_write was once sunk below it), a full __syncthreads() would also wait for the weight and staging loads in flight (vmcnt)
    //
    // Happens-before argument for every LDS region of this kernel (B(g) = the barrier inside group g's tap
    // loop, in front of its last tap; every barrier() is preceded by s_waitcnt lgkmcnt(0) in each wave, so a wave's LDS accesses
    // issued before a barrier are COMPLETE when any wave leaves it):
    //   * activation buffer g & 1, group g's operands.  WRITTEN by every wave during group g-1 (tensor batches behind taps 0-6, the
    //     disparity generators at tap 6), i.e. after B(g-2) and before B(g-1).  Its previous contents, group g-2, are READ by the
    //     rolling fragment loads of group g-2's taps 0-7 - the last ones are issued inside tap 7, in front of B(g-2); tap 8 of group
    //     g-2 multiplies registers and rolls in group g-1's first fragments from the OTHER buffer.  So the last read of group g-2 by
    //     any wave precedes B(g-2), and the first write of group g follows it.  Group g is read after B(g-1).
    //   * collapsed disparity section: step t generates group t+1 into the other half, barrier, multiplies step t while rolling in
    //     step t+1's fragments; the half written at step t+1 was last read by the rolling loads of step t-1, in front of step t's barrier.
    //   * disparity tile ldsD: written once in the prologue (in front of the first barrier), read-only afterwards.
    //   * DELTA epilogue: `red` overlays the activation buffers - barrier() after the main loop (all fragment reads complete), partial
    //     tap planes, barrier(), reduction.
    // The schedule-fuzz build (common.hpp CER_FUZZ, tools/fuzz_schedule.sh: random per-wave sleeps of up to 7.7 k cycles behind every
    // barrier and in front of every LDS write phase) reproduces the first launch bit for bit over 10 000 launches of the five
    // epilogues in both arithmetic forms: no missing edge was found in the form that ships.
    auto barrier = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");        // s_waitcnt lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        CER_FUZZ_POINT();
    };

    // ---- the group sequence: tensors chunk by chunk (two half-chunk groups of 9 taps), then the disparity source
    // stage descriptor of group number `gi` (or kind 0 past the end)
    int ngroups_t = 0;                     // (F8: 32-channel chunks)
    for (int s = 0; s < ntens; ++s) ngroups_t += a.ch[s] >> (F8 ? 5 : 4);
    const int ngroups = ngroups_t + (dsrc ? (coll ? 6 : 4) : 0);
    auto describe = [&](int gi) {
        SxStage st;
        st.kind = 0; st.base = nullptr; st.mtb = 0; st.g = 0;
        if (gi >= ngroups) return st;
        if (gi >= ngroups_t) { st.kind = coll ? 3 : 1; st.g = gi - ngroups_t; return st; }
        int s = 0, g = gi;
        while (g >= (a.ch[s] >> (F8 ? 5 : 4))) { g -= a.ch[s] >> (F8 ? 5 : 4); ++s; }
        st.kind = 2; st.base = a.src[s] + g * (F8 ? 4096 : 2048); st.mtb = (long)(a.ch[s] >> 4) * 2048;
        return st;
    };

    // ---- prologue: every global load is issued before anything waits (one memory latency instead of four): activation tile of
    // group 0, disparity tile, the first two weight slices, the accumulators' initial value
    struct XFrag { half8 h[MT], l[MT]; };
    struct WFrag { half8 h, l; };
    XFrag fx;                                              // activation fragments of the CURRENT step; each m-tile's pair is re-loaded
    WFrag fw[3];                                           // for the next step as soon as its last MFMA of this step has issued
    auto load_w = [&](WFrag& f, int step) {
        const char* p = wlane + (long)min(step, nsteps - 1) * wstep;
        f.h = *reinterpret_cast<const half8*>(p);
        f.l = *reinterpret_cast<const half8*>(p + 1024);
    };
    const SxStage st0 = describe(0);
    uint4 raw0[F8 ? NLD8 * ITEMS8 : ITEMS];
    if (st0.kind == 2) {
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < ITEMS8; ++i) {
                const char* p = stage8_addr(st0, st8_pk[i], tid + 256 * i);
#pragma unroll
                for (int q = 0; q < NLD8; ++q) raw0[NLD8 * i + q] = *reinterpret_cast<const uint4*>(p + 1024 * q);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int pk = st_pk[i];
                raw0[i] = *reinterpret_cast<const uint4*>(st0.base + (long)(pk & 0xFFFFF) * st0.mtb + (((pk >> 25) & 3) * 512 + ((pk >> 20) & 31) * 16));
            }
        }
    }
    constexpr int DITEMS = (DROWS * SX_DTW + 255) / 256;
    float dval[DITEMS];
    if (dsrc) {
#pragma unroll
        for (int i = 0; i < DITEMS; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / SX_DTW, c = idx - r * SX_DTW;
            const int gy = ty0 + r - 4, gx = tx0 + c - 4;
            dval[i] = (idx < DROWS * SX_DTW && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) ? dsrc[(long)gy * a.w + gx] : 0.f;
        }
    }
    // the 8- / 6-bit operand of the block-scaled instruction: 32 bytes (fp8), or 24 bytes of fields + the dword that holds the lane's E8M0 scale
    // (FP6: 16 + 12 bytes - seven registers instead of eight per fragment and weight slot)
    typedef unsigned uintx3 __attribute__((ext_vector_type(3)));
    struct Op8 {
        uint4 a;
        typename std::conditional<F8 == 2, uintx3, uint4>::type b;
        __device__ __forceinline__ void load(const char* p0, const char* p1) {
            a = *reinterpret_cast<const uint4*>(p0);
            b = *reinterpret_cast<const decltype(b)*>(p1);
        }
        __device__ __forceinline__ intx8 v() const {
            if constexpr (F8 == 2) {
                const intx8 lo = (intx8){(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, 0, 0};
                return __builtin_shufflevector(lo, lo, 0, 1, 2, 3, 4, 5, -1, -1);
            } else {
                return (intx8){(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
            }
        }
        __device__ __forceinline__ int scale() const { return (int)b.z; }
    };
    struct W8 { half8 h0, h1; Op8 q; };
    // F8: weights of the current chunk step and of the next WR - 1.  The 64-output-channel kernels (2 x 2 waves, two m-tiles per wave) spend only
    // 256 matrix-pipe cycles per wave on a chunk step - with two co-resident blocks ~500 cycles between a slice's request and its use, less
    // than an L2 round trip: their chunk loop ran at 1.75 x its pipe floor (tools/trace_s16.py --conv q: 8.0 k cycles per chunk against
    // 4.6 k; alone on a CU 388 per 16-channel step against 128).  They request two steps ahead (ring of three; 9 taps = 3 x 3: no slot swap):
    // 6.7 k per chunk.
    constexpr int WR = (F8 && ((WM_ == 2 && MT == 2 && SX_WRING3) || MT >= 8)) ? 3 : 2;         // (three m-tiles per wave: 229 VGPRs already, and 384 pipe cycles per step)
    constexpr bool WSPLIT = F8 && WR == 2 && SX_WSPLIT;                           // two slots, refilled part by part (see mma8_roll)
    W8 w8[WR];
    auto load_w8 = [&](W8& f, int cstep) {                 // (clamped: the steps past the tensors are never multiplied)
        const char* p = wlane8 + (long)min(cstep, (nsteps_t >> 1) - 1) * (2 * wstep);
        f.h0 = *reinterpret_cast<const half8*>(p);
        f.h1 = *reinterpret_cast<const half8*>(p + 1024);
        f.q.load(p + 2048, p + 3072);
    };
    // the steps of the disparity source follow the tensors' (same bytes per 16-channel step in both forms)
    if (F8 && ngroups_t > 0) {
        load_w8(w8[0], 0);
        if constexpr (WR == 3 || WSPLIT) load_w8(w8[1], 1);
    } else {
        load_w(fw[0], 0);
        load_w(fw[1], 1);
    }
    // m-tile m of this wave: m-tile row (ty0 >> 1) + wm*MT + m, column tile_x (tiles are whole m-tiles; rows past the image's
    // last m-tile row do not exist in the tensors)
    const int mrow0 = (ty0 >> 1) + wm * MT;
    const long mt0 = (long)mrow0 * a.mtx + tile_x;
    floatx16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const bool ok = mrow0 + m < a.mty;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.init) {                                  // acc32 layout: one 1-KiB line per (m-tile, n-tile, j)
                if (ok) v = cer_ld4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.init) +
                                                                     (((mt0 + (long)m * a.mtx) * NT + (nb0 >> 5) + wn) * 4 + j) * 1024 + lane * 16));
            } else if (a.bias) {
                v = cer_ld4(a.bias + nb0 + wn * 32 + 8 * j + 4 * kg);
            }
            acc[m][4 * j + 0] = v.x; acc[m][4 * j + 1] = v.y; acc[m][4 * j + 2] = v.z; acc[m][4 * j + 3] = v.w;
        }
    }
    if (dsrc) {
#pragma unroll
        for (int i = 0; i < DITEMS; ++i)
            if (tid + 256 * i < DROWS * SX_DTW) ldsD[tid + 256 * i] = dval[i];
    }
    if (st0.kind == 2) {
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < ITEMS8; ++i) {
                if constexpr (F6) stage6_put(0, st8_pk[i], tid + 256 * i, raw0[2 * i], raw0[2 * i + 1]);
                else stage8_put(0, st8_pk[i], tid + 256 * i, raw0[2 * i], raw0[2 * i + 1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int pk = st_pk[i];
                const unsigned m = (pk & (1 << 27)) ? 0xFFFFFFFFu : 0u;
                uint4 v = raw0[i];
                v.x &= m; v.y &= m; v.z &= m; v.w &= m;
                const int it = tid + 256 * i;
                const int n = it >> 2, r = n / SX_HW, c = n - r * SX_HW;
                if (pk & (1 << 28)) *reinterpret_cast<uint4*>(sx_smem + (r * SX_PITCH + c) * 64 + (it & 3) * 16) = v;
            }
        }
    } else {
        barrier();                                         // the disparity tile feeds the generators
        stage_tap(st0, 0, 6);
    }
    barrier();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] *= a.S;

    auto load_x = [&](XFrag& f, int ph, int pl, int rowoff) {      // rowoff: compile-time (dy rows)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f.h[m] = *reinterpret_cast<const half8*>(sx_smem + ph + (2 * m) * SX_ROWB + rowoff);
            f.l[m] = *reinterpret_cast<const half8*>(sx_smem + pl + (2 * m) * SX_ROWB + rowoff);
        }
    };
    // one step: 3 * MT MFMAs (consecutive MFMAs hit different accumulators), and the next step's fragments rolled in behind them
    auto mma_roll = [&](const WFrag& w, int ph, int pl, int rowoff) {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h, fx.h[m], acc[m], 0, 0, 0);
        // the hi fragments are needed first in the next step: roll them in during the SECOND term (2 * MT - 1 MFMAs of slack for the
        // LDS latency), the lo fragments during the third (a whole step of slack)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.l, fx.h[m], acc[m], 0, 0, 0);
            fx.h[m] = *reinterpret_cast<const half8*>(sx_smem + ph + (2 * m) * SX_ROWB + rowoff);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h, fx.l[m], acc[m], 0, 0, 0);
            fx.l[m] = *reinterpret_cast<const half8*>(sx_smem + pl + (2 * m) * SX_ROWB + rowoff);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // first tap of group gi inside its buffer: tap 0 (dy = dx = 0), or the centre tap for a collapsed disparity group
    const bool first_is_centre0 = st0.kind == 3;
    typedef Op8 F8Frag;
    F8Frag f8[F8 ? MT : 1];                                // F8: fx.h / fx.l hold the f16 hi halves of half-chunk 0 / 1, f8 the fp8 operand
    if (F8 && st0.kind == 2) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            fx.h[m] = *reinterpret_cast<const half8*>(sx_smem + xa[0] + (2 * m) * SX_ROWB8);
            fx.l[m] = *reinterpret_cast<const half8*>(sx_smem + (xa[0] ^ 16) + (2 * m) * SX_ROWB8);
            f8[m].load(sx_smem + (xa[0] ^ 64) + (2 * m) * SX_ROWB8, sx_smem + (xa[0] ^ 80) + (2 * m) * SX_ROWB8);
        }
    } else if (first_is_centre0) load_x(fx, xh[1], xl[1], SX_ROWB);
    else load_x(fx, xh[0], xl[0], 0);

    SX_STAMP();                                            // [1] prologue done
    int step = 0, gi = 0;                                  // gi: group of the current step; its buffer is (gi & 1) * ABUF
    // ---- F8: the tensor sources in 32-channel chunks of 9 taps; per tap and m-tile two f16 MFMAs (xh * wh of both half-chunks) and one
    // fp8 MFMA with K = 64 = both correction terms of both half-chunks:  [xh | xl] * 2^(-8 | 3)  against  [wl | wh] * 2^(5 | -6),
    // the instruction's E8M0 block scale 2^3 puts the products on the accumulator's scale S.  64 + 64 matrix-pipe cycles per m-tile
    // and 32 channels instead of 6 x 32.
    if constexpr (F8) {
        // both correction terms of a 32-channel tap: fp8 form - uniform block scales (2^0, 2^3); FP6 form - the per-lane E8M0 bytes that travel
        // in dword 6 of either operand (the instruction reads six registers of an FP6 operand)
        auto mma_corr = [&](const W8& w, const F8Frag& x, const floatx16& c) {
            if constexpr (F6) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w.q.v(), x.v(), c, 2, 2, 0, w.q.scale(), 0, x.scale());
            else return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w.q.v(), x.v(), c, 0, 0, 0, 127, 0, 130);
        };
        // `refill` (WSPLIT kernels, round 6): chunk step whose weights replace this slot's, PART BY PART, as soon as the part's last MFMA of this
        // tap has been issued - h0 behind the first group, h1 behind the second, the 8- / 6-bit operand behind the third.  Every part then has
        // 1 2/3 taps between its request and its use with the same two register slots (loading a whole slot at the top of the tap before its
        // use gave the f16 hi part of half-chunk 0 ONE tap: with the FP6 form's 768-cycle taps that is less than an L2 round trip, and the
        // chunk loop ran at 1 100 cycles per tap - 7.9 k per chunk with the weights not streamed at all, tools/trace_s16.py).  -1: no refill.
        auto mma8_roll = [&](W8& w, int pa, int rowoff, int refill) {
            const char* wp = wlane8 + (long)min(max(refill, 0), (nsteps_t >> 1) - 1) * (2 * wstep);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h0, fx.h[m], acc[m], 0, 0, 0);
            if (WSPLIT && refill >= 0) w.h0 = *reinterpret_cast<const half8*>(wp);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h1, fx.l[m], acc[m], 0, 0, 0);
                fx.h[m] = *reinterpret_cast<const half8*>(sx_smem + pa + (2 * m) * SX_ROWB8 + rowoff);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (WSPLIT && refill >= 0) w.h1 = *reinterpret_cast<const half8*>(wp + 1024);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = mma_corr(w, f8[m], acc[m]);
                fx.l[m] = *reinterpret_cast<const half8*>(sx_smem + (pa ^ 16) + (2 * m) * SX_ROWB8 + rowoff);
                f8[m].load(sx_smem + (pa ^ 64) + (2 * m) * SX_ROWB8 + rowoff, sx_smem + (pa ^ 80) + (2 * m) * SX_ROWB8 + rowoff);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (WSPLIT && refill >= 0) {
                w.q.load(wp + 2048, wp + 3072);
            }
        };
        // the last tap of the last chunk rolls in the first operands of the disparity section (16-channel layout, f16 hi | lo)
        auto mma8_last = [&](W8& w, int ph, int pl, int rowoff) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h0, fx.h[m], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h1, fx.l[m], acc[m], 0, 0, 0);
                fx.h[m] = *reinterpret_cast<const half8*>(sx_smem + ph + (2 * m) * SX_ROWB + rowoff);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = mma_corr(w, f8[m], acc[m]);
                fx.l[m] = *reinterpret_cast<const half8*>(sx_smem + pl + (2 * m) * SX_ROWB + rowoff);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // one chunk; `last` (compile time): the chunk after which the disparity section (or the epilogue) follows
        auto chunk = [&](auto last_tag) {
            constexpr bool last = decltype(last_tag)::value;
            vary = (gi >> 28) * 0x11111111;
            const SxStage nx = describe(gi + 1);
            const int bufC = (gi & 1) * ABUF, bufN = ABUF - bufC;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (nx.kind == 2 ? (t == 0 || t == 2 || t == 4 || t == 6) : t == 6) stage_tap(nx, bufN, t);
                if (t == 8) barrier();
                // weights one chunk step ahead (a chunk step lasts as long as two 16-channel steps); before the last chunk's last taps
                // the first two slices of the disparity section instead
                W8& wc = w8[WR == 3 ? t % 3 : (t & 1)];
                W8& wn = w8[WR == 3 ? (t + 2) % 3 : ((t + 1) & 1)];
                if (!WSPLIT && (WR == 3 ? (t < 7 || !last) : (t < 8 || !last))) load_w8(wn, gi * 9 + t + WR - 1);
                if (t == 8 && last) { load_w(fw[0], nsteps_t); load_w(fw[1], nsteps_t + 1); }     // (the free weight slot's registers)
                __builtin_amdgcn_sched_barrier(0);
                const int refill = (WSPLIT && (t < 7 || !last)) ? gi * 9 + t + 2 : -1;
                if (t < 8) mma8_roll(wc, (xa[(t + 1) % 3] ^ vary) + bufC, ((t + 1) / 3) * SX_ROWB8, refill);
                else if (!last) mma8_roll(wc, (xa[0] ^ vary) + bufN, 0, refill);
                else mma8_last(wc, xh[1] + bufN, xl[1] + bufN, SX_ROWB);       // (collapsed disparity group or nothing: see the launcher)
            }
            if constexpr (WR == 2) {
                if (!last) {   // nine taps per chunk: the slot that holds the next step's weights becomes slot 0 again
                    const W8 tmp = w8[0]; w8[0] = w8[1]; w8[1] = tmp;
                }
            }
            SX_STAMP();
            ++gi;
        };
        while (gi + 1 < ngroups_t) chunk(std::false_type{});
        if (gi < ngroups_t) chunk(std::true_type{});
        step = nsteps_t;
    }
    // ---- tensor half-chunks and literal disparity groups: 9 taps each
    const int ngroups9 = F8 ? 0 : ngroups_t + ((dsrc && !coll) ? 4 : 0);      // (F8: tensors in chunks above, disparity collapsed only)
    for (; gi < ngroups9; ++gi, step += 9) {
        vary = (gi >> 28) * 0x11111111;
        const SxStage nx = describe(gi + 1);
        const int bufC = (gi & 1) * ABUF, bufN = ABUF - bufC;
        const bool nxt_centre = nx.kind == 3;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // (1) staging of the next group into the other buffer
            if (t == 0 || t == 3 || t == 6) stage_tap(nx, bufN, t);
            if (t == 8) barrier();                         // next group's data complete and visible; nobody reads its buffer's old contents any more
            // (2) weights of step t + 2
            load_w(fw[(t + 2) % 3], step + t + 2);
            __builtin_amdgcn_sched_barrier(0);
            // (3) multiply step t, rolling in the operands of step t + 1
            if (t < 8) mma_roll(fw[t % 3], xh[(t + 1) % 3] + bufC, xl[(t + 1) % 3] + bufC, ((t + 1) / 3) * SX_ROWB);
            else if (nxt_centre) mma_roll(fw[t % 3], xh[1] + bufN, xl[1] + bufN, SX_ROWB);
            else mma_roll(fw[t % 3], xh[0] + bufN, xl[0] + bufN, 0);
        }
        SX_STAMP();                                        // after every group
    }
    // ---- collapsed disparity: 6 groups of one (centre) tap
    if (dsrc && coll) {
        vary = (gi >> 28) * 0x11111111;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int bufC = ((gi + t) & 1) * ABUF, bufN = ABUF - bufC;
            if (t < 5) {
                gen_collapsed(t + 1, bufN);
                barrier();
            }
            load_w(fw[(t + 2) % 3], step + t + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma_roll(fw[t % 3], xh[1] + bufN, xl[1] + bufN, SX_ROWB);
        }
        step += 6;
        gi += 6;
    }
    SX_STAMP();                                            // main loop done

    // ---- epilogue.  acc[m][4j + e] = S * conv for channel nb0 + wn*32 + 8j + 4kg + e of the lane's pixel.
    // v_permlane32_swap on register pairs (j = 2jp, 2jp + 1): lanes 0-31 end up with channels 16jp + 0..7, lanes 32-63 with
    // 16jp + 8..15 of their pixel - 8 consecutive channels starting at cb = nb0 + wn*32 + 16jp + 8kg: exactly this lane's
    // 16-byte pieces of the frag16 planes (hi | lo) of group cb >> 4, so every tensor access below is one contiguous KiB per wave.
    // ---- rim correction of the collapsed disparity form.  The 3x3 conv zero-pads the FEATURE map (core/update.py:80-85: features
    // of a position q outside the image are 0), while the collapsed filter evaluated 100 * (d~(q + u - 3) - 0) there (d~ = zero
    // -padded disparity).  For a pixel on the image rim subtract  E = sum_{taps t with q_t outside} sum_u w[t][u] * 100 * d~(q_t+u-3):
    // per touched edge a 27-tap filter on the disparity, minus the corner tap two edges share.  It is a small matmul
    // ([32 channels x 27] x [27 x pixels]), so it runs as extra K-steps: the filters arrive packed like weights (negated, same
    // scale as the disparity source), the "activations" are generated in registers as B fragments (zero for pixels off the edge).
    if (coll && a.edge && !interior) {
        const _Float16* ew = reinterpret_cast<const _Float16*>(a.edge) + ((long)(nb0 >> 5) + wn) * 1024 + lane * 8;
#pragma unroll                                             // (acc[m] must stay statically indexed: registers, not scratch)
        for (int m = 0; m < MT; ++m) {
            const int py = 2 * (wm * MT + m) + (li >> 4), px = li & 15;       // pixel inside the tile
            const int y = ty0 + py, x = tx0 + px;
            const bool in = y < a.h && x < a.w;
            const bool top = in && y == 0, bot = in && y == a.h - 1, lef = in && x == 0, rig = in && x == a.w - 1;
            // disparity at image (r, c) from the tile in LDS: ldsD(i, j) = image (ty0 - 4 + i, tx0 - 4 + j), zero outside the image
            auto D = [&](int r, int c) { return ldsD[(r - ty0 + 4) * SX_DTW + (c - tx0 + 4)]; };
            auto apply = [&](int e, int nks, bool on, auto&& pos) {           // pos(k, r, c): image position of filter tap k
                if (!__any(on)) return;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks >= nks) break;
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int k = 16 * ks + 8 * kg + q;
                        int r = 0, c = 0;
                        const bool ok = pos(k, r, c);
                        v[q] = (on && ok) ? 100.0f * D(r, c) : 0.f;
                    }
                    half8 fh, fl;
                    sx_split8(v, a.disp_scale, fh, fl);
                    const _Float16* wp = ew + (long)((e * 2 + ks) * NT) * 1024;
                    const half8 wh = *reinterpret_cast<const half8*>(wp), wl = *reinterpret_cast<const half8*>(wp + 512);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fh, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fl, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fh, acc[m], 0, 0, 0);
                }
            };
            apply(0, 2, top, [&](int k, int& r, int& c) { r = k / 9; c = x + k % 9 - 4; return k < 27; });
            apply(1, 2, bot, [&](int k, int& r, int& c) { r = a.h - 3 + k / 9; c = x + k % 9 - 4; return k < 27; });
            apply(2, 2, lef, [&](int k, int& r, int& c) { r = y + k / 3 - 4; c = k % 3; return k < 27; });
            apply(3, 2, rig, [&](int k, int& r, int& c) { r = y + k / 3 - 4; c = a.w - 3 + k % 3; return k < 27; });
            apply(4, 1, top && lef, [&](int k, int& r, int& c) { r = k / 3; c = k % 3; return k < 9; });
            apply(5, 1, top && rig, [&](int k, int& r, int& c) { r = k / 3; c = a.w - 3 + k % 3; return k < 9; });
            apply(6, 1, bot && lef, [&](int k, int& r, int& c) { r = a.h - 3 + k / 3; c = k % 3; return k < 9; });
            apply(7, 1, bot && rig, [&](int k, int& r, int& c) { r = a.h - 3 + k / 3; c = a.w - 3 + k % 3; return k < 9; });
        }
    }
    const int half = a.cout >> 1;
    if constexpr (EPI == SX_EPI_DELTA) {
        // delta head, fused (core/update.py:68-71): hid = relu(conv) stays in registers; its hi|lo halves ARE B fragments of the
        // projection  T[tap][p] = sum_c w2[tap][c] * hid[p][c]  over this wave's 32 channels (2 k16-steps x 3 MFMAs per m-tile);
        // the four waves' partial tap planes are summed through LDS and leave as [9][P] planes for cer_delta_sum_f32.
        static_assert(EPI != SX_EPI_DELTA || WM_ == 1, "the DELTA epilogue is built for the 1 x 4 wave layout");
        const _Float16* w2 = reinterpret_cast<const _Float16*>(a.aux) + ((long)(by * WN_ + wn) * 2) * 1024;     // [jp][hi|lo][lane][8]
        float* red = reinterpret_cast<float*>(sx_smem);    // [wn][m][tap 9][32 px]
        static_assert(WN_ * MT * 9 * 32 * 4 <= 2 * ABUF, "reduction scratch does not fit");
        barrier();                                         // every wave has finished reading the activation buffers
        const float hs = a.invS * (float)(1 << SX_HID_LOG2);
        float hmax = 0.f;                                  // largest hidden activation of this lane (the map never reaches HBM: checked here)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            floatx16 tm;
#pragma unroll
            for (int r = 0; r < 16; ++r) tm[r] = 0.f;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][8 * jp + e]), __float_as_uint(acc[m][8 * jp + 4 + e]), false, false);
                    v[e] = fmaxf(__uint_as_float(sw[0]), 0.f);
                    v[4 + e] = fmaxf(__uint_as_float(sw[1]), 0.f);
                }
                hmax = fmaxf(hmax, fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]))));
                half8 hh, hl;
                sx_split8(v, hs, hh, hl);
                const half8 wh = *reinterpret_cast<const half8*>(w2 + (jp * 2 + 0) * 512 + lane * 8);
                const half8 wl = *reinterpret_cast<const half8*>(w2 + (jp * 2 + 1) * 512 + lane * 8);
                tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hh, tm, 0, 0, 0);
                tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hl, tm, 0, 0, 0);
                tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hh, tm, 0, 0, 0);
            }
            // tm[r]: tap (r&3) + 8(r>>2) + 4kg of pixel li: kg = 0 holds taps 0-3 and 8, kg = 1 taps 4-7
            float* rp = red + ((wn * MT + m) * 9) * 32 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) rp[(r + 4 * kg) * 32] = tm[r];
            if (kg == 0) rp[8 * 32] = tm[4];
        }
        if (a.flag && __ballot(!(hmax * hs <= 65504.0f)) != 0ull && lane == 0) atomicOr(a.flag, 2);     // saturated (or not finite)
        barrier();
        const long P = (long)a.h * a.w;
        for (int idx = tid; idx < MT * 9 * 32; idx += 256) {
            const int m = idx / (9 * 32), rem = idx - m * 9 * 32, tap = rem >> 5, px = rem & 31;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < WN_; ++q) s += red[((q * MT + m) * 9 + tap) * 32 + px];
            const int gy = ty0 + 2 * m + (px >> 4), gx = tx0 + (px & 15);
            if (gy < a.h && gx < a.w) a.out[((long)by * 9 + tap) * P + (long)gy * a.w + gx] = s * a.proj_inv;
        }
        return;
    } else {
        const int nt = (nb0 >> 5) + wn;                    // this wave's 32-channel tile of the output
        // frag16 / f32x8 byte offset of the lane's piece of (m-tile, 16-channel group g) in a tensor of G groups: both planes
        // (hi | lo, or the two float4) are 1 KiB apart
        auto piece = [&](int m, int G, int g) { return (((mt0 + (long)m * a.mtx) * G + g) * 2) * 1024 + lane * 16; };
        if (EPI == CER_EPI_LINEAR && !a.out_split) {       // acc32 layout: the accumulators as they are (a later conv's `init`)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (mrow0 + m >= a.mty) continue;          // wave-uniform: the m-tile row is past the image
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.out) + (((mt0 + (long)m * a.mtx) * NT + nt) * 4 + j) * 1024 + lane * 16) =
                        make_float4(acc[m][4 * j] * a.invS, acc[m][4 * j + 1] * a.invS, acc[m][4 * j + 2] * a.invS, acc[m][4 * j + 3] * a.invS);
            }
        } else {
            const bool is_r = EPI == CER_EPI_GATES && nt * 32 >= half;          // wave-uniform: this wave holds reset-gate channels
            const int G = (EPI == CER_EPI_GATES ? half : a.cout) >> 4;         // 16-channel groups of the destination / aux tensors
            const int g0 = (EPI == CER_EPI_GATES ? ((nt * 32) % half) >> 4 : nt * 2);
            // (1) every operand of the gate math is requested before anything is computed or stored: the stores below may alias
            // the loads as far as the compiler knows, so it would otherwise run load -> math -> store once per (m, jp)
            half8 ph[MT][2], pl[MT][2];
            float4 z0[MT][2], z1[MT][2];
            if ((EPI == CER_EPI_GATES && is_r) || EPI == CER_EPI_GRU) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        if (mrow0 + m >= a.mty) continue;
                        const char* p = reinterpret_cast<const char*>(a.aux) + piece(m, G, g0 + jp);
                        ph[m][jp] = *reinterpret_cast<const half8*>(p);
                        pl[m][jp] = *reinterpret_cast<const half8*>(p + 1024);
                        if (EPI == CER_EPI_GRU) {
                            const char* zp = reinterpret_cast<const char*>(a.aux2) + piece(m, G, g0 + jp);
                            z0[m][jp] = *reinterpret_cast<const float4*>(zp);
                            z1[m][jp] = *reinterpret_cast<const float4*>(zp + 1024);
                        }
                    }
            }
            // (2) math + stores
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][8 * jp + e]), __float_as_uint(acc[m][8 * jp + 4 + e]), false, false);
                        v[e] = __uint_as_float(sw[0]) * a.invS;
                        v[4 + e] = __uint_as_float(sw[1]) * a.invS;
                    }
                    if (mrow0 + m >= a.mty) continue;      // wave-uniform
                    const long off = piece(m, G, g0 + jp);
                    auto st_split = [&](float* base, const float (&o)[8]) {
                        half8 hi, lo;
                        sx_split8(o, a.out_scale, hi, lo);
                        *reinterpret_cast<half8*>(reinterpret_cast<char*>(base) + off) = hi;
                        *reinterpret_cast<half8*>(reinterpret_cast<char*>(base) + off + 1024) = lo;
                    };
                    if (EPI == CER_EPI_LINEAR || EPI == CER_EPI_RELU) {
                        if (EPI == CER_EPI_RELU)
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                        st_split(a.out, v);
                    } else if (EPI == CER_EPI_GATES) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = sx_sigmoid(v[e]);
                        if (!is_r) {                       // z stays fp32 (f32x8 layout): blend operand of the q conv's epilogue
                            *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.out) + off) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.out) + off + 1024) = make_float4(v[4], v[5], v[6], v[7]);
                        } else {
                            float hp[8];
                            sx_join8(ph[m][jp], pl[m][jp], a.aux_inv, hp);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] *= hp[e];
                            st_split(a.out2, v);
                        }
                    } else if (EPI == CER_EPI_GRU) {
                        float hp[8];
                        sx_join8(ph[m][jp], pl[m][jp], a.aux_inv, hp);
                        const float z[8] = {z0[m][jp].x, z0[m][jp].y, z0[m][jp].z, z0[m][jp].w, z1[m][jp].x, z1[m][jp].y, z1[m][jp].z, z1[m][jp].w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (1.0f - z[e]) * hp[e] + z[e] * sx_tanh(v[e]);
                        st_split(a.out, v);
                    }
                }
            }
        }
    }
#if SX_TRACE
    if ((EPI == CER_EPI_GATES || EPI == CER_EPI_GRU) && lane == 0) {
        unsigned long long* tb = (unsigned long long*)(EPI == CER_EPI_GATES ? (const void*)a.aux2 : (const void*)a.out2) + ((long)blockIdx.x * 4 + wave) * 32;
        tb[0] = (unsigned long long)trace_n;
        tb[1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        tb[2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        tb[3] = __builtin_readcyclecounter();
        tb[4] = (unsigned long long)nsteps;
#pragma unroll
        for (int k = 0; k < 24; ++k) tb[8 + k] = trace_t[k];
    }
#endif
}

// ---------------------------------------------------------------------------------------- host side

static int sx_order(const int* kind, int nsrc, int* order) {               // tensors first, the disparity source last
    int n = 0, nd = 0;
    for (int s = 0; s < nsrc; ++s) if (kind[s] != 1) order[n++] = s;
    for (int s = 0; s < nsrc; ++s) if (kind[s] == 1) { order[n++] = s; ++nd; }
    return nd;
}

static long sx_steps(const int* ch, const int* kind, int nsrc, int collapsed) {
    long steps = 0;
    for (int s = 0; s < nsrc; ++s) steps += kind[s] == 1 ? (collapsed ? 6 : 36) : (ch[s] / 16) * 9;
    return steps;
}

extern "C" long cer_conv3x3_s16_packed_size(int Cout, const int* ch, const int* kind, int nsrc, int collapsed) {
    if (!ch || !kind || nsrc <= 0 || nsrc > CER_CONV_MAX_SRC || Cout <= 0 || Cout % 32) return CER_ESHAPE;
    int nd = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (kind[s] == 1) { ++nd; if (ch[s] != 49) return CER_ESHAPE; }
        else if (ch[s] % 32) return CER_ESHAPE;
    }
    collapsed &= 1;                                        // (bit 1 = fp8-, bit 2 = FP6-correction form: same size)
    if (nd > 1 || (collapsed && nd == 0)) return CER_ESHAPE;
    return sx_steps(ch, kind, nsrc, collapsed) * (Cout / 32) * 1024;        // in halves
}

// collapsed 81-tap filter of the disparity source (see gru_f16x3.hip: cer_conv3x3_f16x3_pack_collapsed), fp64 sums
static double sx_w9(const float* w, int Cin, int co, int c, int sidx) {
    if (sidx >= 81) return 0.0;
    const int sy = sidx / 9, sx = sidx % 9;
    double acc = 0.0;
    for (int ty = 0; ty < 3; ++ty)
        for (int tx = 0; tx < 3; ++tx) {
            const int uy = sy - ty, ux = sx - tx;
            if (uy >= 0 && uy < 7 && ux >= 0 && ux < 7) acc += (double)w[((long)co * Cin + c + uy * 7 + ux) * 9 + ty * 3 + tx];
        }
    if (sy >= 3 && sy <= 5 && sx >= 3 && sx <= 5) {
        const int t = (sy - 3) * 3 + (sx - 3);
        for (int u = 0; u < 49; ++u) acc -= (double)w[((long)co * Cin + c + u) * 9 + t];
    }
    return acc;
}

// log2 of the product scale S shared by all sources: the largest power of two that keeps every scaled weight
// |w| * S / 2^log2sx(src) below 2^14 (f16 max 65504), for the literal and the collapsed packing alike.  Returns CER_ESHAPE
// (as a value < -1000) if some source's largest scaled weight would then fall below 2^-3 (its lo halves go subnormal: fewer than 22 bits).
extern "C" int cer_conv3x3_s16_scale(const float* w, int Cout, int Cin, const int* ch, const int* kind, const int* log2sx, int nsrc) {
    if (!w || !ch || !kind || !log2sx || nsrc <= 0 || nsrc > CER_CONV_MAX_SRC) return -100000;
    double wmax[CER_CONV_MAX_SRC];
    int c = 0;
    for (int s = 0; s < nsrc; ++s) {
        double m = 0.0;
        for (int co = 0; co < Cout; ++co) {
            for (int i = 0; i < ch[s]; ++i)
                for (int t = 0; t < 9; ++t) m = fmax(m, fabs((double)w[((long)co * Cin + c + i) * 9 + t]));
            if (kind[s] == 1) {
                for (int sidx = 0; sidx < 81; ++sidx) m = fmax(m, fabs(sx_w9(w, Cin, co, c, sidx)));
                // the rim-correction filters sum at most 3 taps per entry
                double m3 = 0.0;
                for (int u = 0; u < 49; ++u) {
                    for (int ty = 0; ty < 3; ++ty) {
                        double sr = 0.0;
                        for (int tx = 0; tx < 3; ++tx) sr += fabs((double)w[((long)co * Cin + c + u) * 9 + ty * 3 + tx]);
                        m3 = fmax(m3, sr);
                    }
                    for (int tx = 0; tx < 3; ++tx) {
                        double sc = 0.0;
                        for (int ty = 0; ty < 3; ++ty) sc += fabs((double)w[((long)co * Cin + c + u) * 9 + ty * 3 + tx]);
                        m3 = fmax(m3, sc);
                    }
                }
                m = fmax(m, m3);
            }
        }
        wmax[s] = m;
        c += ch[s];
    }
    if (c != Cin) return -100000;
    int best = 1000;
    for (int s = 0; s < nsrc; ++s) {
        if (wmax[s] <= 0.0) continue;
        const int k = (int)floor(log2(16384.0 / wmax[s])) + log2sx[s];
        best = k < best ? k : best;
    }
    if (best == 1000) best = 14;
    // the shared scale is set by the source with the largest weights.  A source's lo halves are f16 residuals of magnitude <= 2^-11 of
    // its scaled weights; once they fall into the f16 subnormal range (spacing 2^-24) the source keeps fewer than 22 bits relative to
    // its own largest weight when that weight, scaled, is below 2^-3: refuse, the caller falls back to the f16x3 kernels (per-tensor splits)
    for (int s = 0; s < nsrc; ++s)
        if (wmax[s] > 0.0 && ldexp(wmax[s], best - log2sx[s]) < 0.125) return -100000 + CER_ESHAPE;
    return best;
}

// Rim-correction filters of the collapsed disparity form (see the kernel's epilogue), packed like weight slices:
// [edge 8][k16-step 2][ntile][hi|lo][lane][8] halves of  -sign * Wedge[k][co] * 2^(log2S - log2sx(disparity source)):
// edges 0-3 = top, bottom, left, right (27 taps: top/bottom k = a * 9 + sx, left/right k = sy * 3 + b), 4-7 = the corner taps two
// edges share (9 taps, k = i * 3 + j; opposite sign): top-left, top-right, bottom-left, bottom-right.
extern "C" long cer_conv3x3_s16_edge_size(int Cout) { return (Cout > 0 && Cout % 32 == 0) ? 16L * (Cout / 32) * 1024 : CER_ESHAPE; }

static void sx_pack_slice(_Float16* packed, long step, int NT, int nt, const double* col, double scale);

extern "C" int cer_conv3x3_s16_edge_pack(const float* w, void* out_v, int Cout, int Cin, const int* ch, const int* kind, const int* log2sx, int nsrc,
                                         int log2S) {
    if (!w || !out_v || !ch || !kind || !log2sx || nsrc <= 0 || nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    if (Cout % 32) return CER_ESHAPE;
    int c0 = -1, c = 0, sd = -1;
    for (int s = 0; s < nsrc; ++s) {
        if (kind[s] == 1) { if (ch[s] != 49 || c0 >= 0) return CER_ESHAPE; c0 = c; sd = s; }
        c += ch[s];
    }
    if (c != Cin || c0 < 0) return CER_ESHAPE;
    auto W = [&](int co, int uy, int ux, int ty, int tx) -> double {
        if (uy < 0 || uy > 6 || ux < 0 || ux > 6) return 0.0;
        return (double)w[((long)co * Cin + c0 + uy * 7 + ux) * 9 + ty * 3 + tx];
    };
    // Wedge[edge][k][co]
    auto edge_w = [&](int e, int k, int co) -> double {
        if (e < 4) {
            if (k >= 27) return 0.0;
            double v = 0;
            if (e == 0) { const int a = k / 9, sx = k % 9; for (int tx = 0; tx < 3; ++tx) v += W(co, a + 4, sx - tx, 0, tx); }
            if (e == 1) { const int a = k / 9, sx = k % 9; for (int tx = 0; tx < 3; ++tx) v += W(co, a, sx - tx, 2, tx); }
            if (e == 2) { const int sy = k / 3, b = k % 3; for (int ty = 0; ty < 3; ++ty) v += W(co, sy - ty, b + 4, ty, 0); }
            if (e == 3) { const int sy = k / 3, b = k % 3; for (int ty = 0; ty < 3; ++ty) v += W(co, sy - ty, b, ty, 2); }
            return -v;                                     // the rim terms are SUBTRACTED from the collapsed result
        }
        if (k >= 9) return 0.0;
        const int i = k / 3, j = k % 3;
        if (e == 4) return W(co, i + 4, j + 4, 0, 0);      // ... and the shared corner tap is added back once
        if (e == 5) return W(co, i + 4, j, 0, 2);
        if (e == 6) return W(co, i, j + 4, 2, 0);
        return W(co, i, j, 2, 2);
    };
    _Float16* packed = (_Float16*)out_v;
    const int NT = Cout / 32;
    const double scale = ldexp(1.0, log2S - log2sx[sd]);
    double col[16 * 32];
    for (int e = 0; e < 8; ++e)
        for (int ks = 0; ks < 2; ++ks)
            for (int nt = 0; nt < NT; ++nt) {
                for (int k = 0; k < 16; ++k)
                    for (int j = 0; j < 32; ++j) col[k * 32 + j] = edge_w(e, ks * 16 + k, nt * 32 + j);
                sx_pack_slice(packed, e * 2 + ks, NT, nt, col, scale);
            }
    return CER_OK;
}

static void sx_pack_slice(_Float16* packed, long step, int NT, int nt, const double* col /* [16 k][32 co] */, double scale) {
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
            float v = (float)(col[((lane >> 5) * 8 + e) * 32 + (lane & 31)] * scale);
            v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            const long base = (step * NT + nt) * 2;
            packed[(base + 0) * 512 + lane * 8 + e] = hi;
            packed[(base + 1) * 512 + lane * 8 + e] = lo;
        }
}

// OIHW fp32 -> [step][ntile32][hi|lo][lane][8] halves of w * 2^(log2S - log2sx(src)); steps: tensors in source order
// (32-channel chunk, 16-channel half, tap), then the disparity source (collapsed: 6 single-tap groups of the 81-tap filter;
// literal: 4 groups x 9 taps of the 49 unfold channels)
// e4m3 (OCP: bias 7, subnormals, no infinities, largest finite 448), round to nearest even, saturating
static unsigned char sx_e4m3(double v) {
    const unsigned sgn = v < 0 ? 0x80u : 0u;
    double a = fabs(v);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0) return (unsigned char)(sgn | 0x7e);
    if (a < ldexp(1.0, -10)) return (unsigned char)sgn;    // below half the smallest subnormal (ties to even: 0)
    int e;
    frexp(a, &e);                                          // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;                                         // a in [2^E, 2^(E+1))
    if (E < -6) E = -6;                                    // subnormal range: spacing 2^-9
    const double q = nearbyint(ldexp(a, 3 - E));           // in units of 2^(E-3) (nearbyint: ties to even in the default mode)
    int M = (int)q;                                        // 8..16 for normals, 0..8 for subnormals
    int Eb = E + 7;
    if (a < ldexp(1.0, -6)) { Eb = 0; if (M == 8) { Eb = 1; M = 0; } }
    else { if (M == 16) { M = 0; ++Eb; } else M -= 8; }
    if (Eb > 15 || (Eb == 15 && M > 6)) return (unsigned char)(sgn | 0x7e);
    return (unsigned char)(sgn | (Eb << 3) | M);
}

// fp8-correction form of a tensor chunk step (32 channels, one tap): 4 KiB per n-tile = f16 hi halves of half-chunk 0 | of half-chunk 1
// (each [lane][8], as in sx_pack_slice) | the A operand of v_mfma_scale_f32_32x32x64_f8f6f4, bytes 0-15 | bytes 16-31 of every lane:
// lane (co = lane & 31, kg = lane >> 5): [wl * 2^5 (8) | wh * 2^-6 (8)] of half-chunk 0's channels 8kg..8kg+7, then the same of half-chunk 1
static void sx_pack_chunk8(_Float16* packed, long cstep, int NT, int nt, const double* col /* [32 k][32 co] */, double scale) {
    char* base = reinterpret_cast<char*>(packed) + (cstep * NT + nt) * 4096;
    for (int lane = 0; lane < 64; ++lane)
        for (int hc = 0; hc < 2; ++hc)
            for (int e = 0; e < 8; ++e) {
                float v = (float)(col[(hc * 16 + (lane >> 5) * 8 + e) * 32 + (lane & 31)] * scale);
                v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                reinterpret_cast<_Float16*>(base + hc * 1024)[lane * 8 + e] = hi;
                unsigned char* q = reinterpret_cast<unsigned char*>(base + 2048 + hc * 1024 + lane * 16);
                q[e] = sx_e4m3(ldexp((double)(float)lo, 5));
                q[8 + e] = sx_e4m3(ldexp((double)(float)hi, -6));
            }
}

// e2m3 (FP6: 1 sign, 2 exponent, 3 mantissa bits: 0, 0.125 .. 0.875, 1 .. 1.875, 2 .. 3.75, 4 .. 7.5), round to nearest even, saturating
static unsigned sx_e2m3(double v) {
    const unsigned sgn = v < 0 ? 32u : 0u;
    const double a = fabs(v);
    if (!(a == a)) return sgn | 31u;
    if (a >= 7.5) return sgn | 31u;
    const int E = a < 2.0 ? 0 : (a < 4.0 ? 1 : 2);          // steps of 0.125 (subnormals and [1, 2)), 0.25, 0.5
    const int q = (int)nearbyint(ldexp(a, 3 - E));          // a in units of the step: 0..16 (E = 0), 8..16
    if (E == 0) return sgn | (unsigned)q;                   // codes 0..15 are linear in the value (q = 16: code 16 = 2.0)
    return sgn | (unsigned)(8 * E + q);                     // e = E + 1, m = q - 8 (q = 16 carries into the next exponent: 7.5 is the cap above)
}

// FP6-correction form of a tensor chunk step (round 6; 32 channels, one tap): 4 KiB per n-tile = f16 hi halves of half-chunk 0 | of
// half-chunk 1 (as in the fp8 form) | bytes 0-15 | bytes 16-31 of every lane's A operand of v_mfma_scale_f32_32x32x64_f8f6f4 in its FP6 form:
// lane (co = lane & 31, kg = lane >> 5): 32 six-bit e2m3 fields, field i at bit 6 i = [wl * 2^11 (8) | wh (8)] of half-chunk 0's channels
// 8kg..8kg+7, then the same of half-chunk 1, all divided by ONE power of two t = 2^(e - 2) (e: exponent of the block's largest magnitude; one up
// where that would land above 7.75) = 24 bytes; byte 24 = E8M0 of t * 2^-11 (the lane's scale operand: it undoes the 2^11 on BOTH correction
// terms - the activations carry [xh | xl * 2^11] in the same positions), bytes 25-31 zero.
static void sx_pack_chunk6(_Float16* packed, long cstep, int NT, int nt, const double* col /* [32 k][32 co] */, double scale) {
    char* base = reinterpret_cast<char*>(packed) + (cstep * NT + nt) * 4096;
    for (int lane = 0; lane < 64; ++lane) {
        double f[32];
        double mx = 0.0;
        for (int hc = 0; hc < 2; ++hc)
            for (int e = 0; e < 8; ++e) {
                float v = (float)(col[(hc * 16 + (lane >> 5) * 8 + e) * 32 + (lane & 31)] * scale);
                v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                reinterpret_cast<_Float16*>(base + hc * 1024)[lane * 8 + e] = hi;
                f[hc * 16 + e] = ldexp((double)(float)lo, 11);
                f[hc * 16 + 8 + e] = (double)(float)hi;
                mx = fmax(mx, fmax(fabs(f[hc * 16 + e]), fabs(f[hc * 16 + 8 + e])));
            }
        int te = -100;                                       // t = 2^te
        if (mx > 0.0) {
            int e2;
            frexp(mx, &e2);                                  // mx in [2^(e2-1), 2^e2)
            te = e2 - 1 - 2;
            if (ldexp(mx, -te) > 7.75) ++te;
            if (te < -100) te = -100;
        }
        unsigned fields[32];
        for (int i = 0; i < 32; ++i) fields[i] = sx_e2m3(ldexp(f[i], -te));
        unsigned q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 32; ++i) {
            const int bit = 6 * i;
            q[bit >> 5] |= fields[i] << (bit & 31);
            if ((bit & 31) > 26) q[(bit >> 5) + 1] |= fields[i] >> (32 - (bit & 31));
        }
        q[6] = (unsigned)(te - 11 + 127);
        memcpy(base + 2048 + lane * 16, q, 16);
        memcpy(base + 3072 + lane * 16, q + 4, 16);
    }
}

extern "C" int cer_conv3x3_s16_pack(const float* w, void* packed_v, int Cout, int Cin, const int* ch, const int* kind, const int* log2sx,
                                    int nsrc, int collapsed, int log2S) {
    if (!w || !packed_v || !ch || !kind || !log2sx) return CER_EINVAL;
    if ((collapsed & 6) == 6) return CER_EINVAL;
    const int fp6 = (collapsed & 4) != 0;                   // tensor sources in the FP6-correction form (CER_EPI_CORR_FP6 launches)
    const int fp8 = (collapsed & 2) != 0 || fp6;            // ... in the fp8-correction form (CER_EPI_CORR_FP8 launches): same step structure
    collapsed &= 1;
    if (cer_conv3x3_s16_packed_size(Cout, ch, kind, nsrc, collapsed) < 0) return CER_ESHAPE;
    int order[CER_CONV_MAX_SRC], c0[CER_CONV_MAX_SRC];
    sx_order(kind, nsrc, order);
    int c = 0;
    for (int s = 0; s < nsrc; ++s) { c0[s] = c; c += ch[s]; }
    if (c != Cin) return CER_ESHAPE;
    _Float16* packed = (_Float16*)packed_v;
    const int NT = Cout / 32;
    double col[16 * 32];
    long step = 0;
    for (int oi = 0; oi < nsrc; ++oi) {
        const int s = order[oi];
        const double scale = ldexp(1.0, log2S - log2sx[s]);
        if (kind[s] != 1 && fp8) {
            double col32[32 * 32];
            for (int c32 = 0; c32 < ch[s] / 32; ++c32)
                for (int tap = 0; tap < 9; ++tap, step += 2)
                    for (int nt = 0; nt < NT; ++nt) {
                        for (int k = 0; k < 32; ++k)
                            for (int j = 0; j < 32; ++j) col32[k * 32 + j] = w[((long)(nt * 32 + j) * Cin + c0[s] + c32 * 32 + k) * 9 + tap];
                        if (fp6) sx_pack_chunk6(packed, step / 2, NT, nt, col32, scale);
                        else sx_pack_chunk8(packed, step / 2, NT, nt, col32, scale);
                    }
        } else if (kind[s] != 1) {
            for (int g = 0; g < ch[s] / 16; ++g)
                for (int tap = 0; tap < 9; ++tap, ++step)
                    for (int nt = 0; nt < NT; ++nt) {
                        for (int k = 0; k < 16; ++k)
                            for (int j = 0; j < 32; ++j) col[k * 32 + j] = w[((long)(nt * 32 + j) * Cin + c0[s] + g * 16 + k) * 9 + tap];
                        sx_pack_slice(packed, step, NT, nt, col, scale);
                    }
        } else if (collapsed) {
            for (int g = 0; g < 6; ++g, ++step)
                for (int nt = 0; nt < NT; ++nt) {
                    for (int k = 0; k < 16; ++k)
                        for (int j = 0; j < 32; ++j) col[k * 32 + j] = sx_w9(w, Cin, nt * 32 + j, c0[s], g * 16 + k);
                    sx_pack_slice(packed, step, NT, nt, col, scale);
                }
        } else {
            for (int g = 0; g < 4; ++g)
                for (int tap = 0; tap < 9; ++tap, ++step)
                    for (int nt = 0; nt < NT; ++nt) {
                        for (int k = 0; k < 16; ++k)
                            for (int j = 0; j < 32; ++j) {
                                const int ci = g * 16 + k;
                                col[k * 32 + j] = ci < 49 ? w[((long)(nt * 32 + j) * Cin + c0[s] + ci) * 9 + tap] : 0.0;
                            }
                        sx_pack_slice(packed, step, NT, nt, col, scale);
                    }
        }
    }
    return CER_OK;
}

// delta head projection: w2 OIHW [1, C, 3, 3] -> A fragments [C/32][k16-step 2][hi|lo][lane][8]: lane (tap = lane & 31, kg = lane >> 5)
// holds channels 32*blk + 16*jp + 8*kg + e, scaled by 2^log2s (tap >= 9: zero)
extern "C" long cer_delta_proj_s16_packed_size(int C) { return C % 128 ? CER_ESHAPE : (long)(C / 32) * 2 * 2 * 512; }

extern "C" int cer_delta_proj_s16_pack(const float* w2, void* packed_v, int C, int* log2s_out) {
    if (!w2 || !packed_v || !log2s_out) return CER_EINVAL;
    if (C % 128) return CER_ESHAPE;
    double m = 0.0;
    for (long i = 0; i < (long)C * 9; ++i) m = fmax(m, fabs((double)w2[i]));
    const int log2s = m > 0.0 ? (int)floor(log2(16384.0 / m)) : 14;
    *log2s_out = log2s;
    _Float16* packed = (_Float16*)packed_v;
    for (int blk = 0; blk < C / 32; ++blk)
        for (int jp = 0; jp < 2; ++jp)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int tap = lane & 31, c = blk * 32 + jp * 16 + (lane >> 5) * 8 + e;
                    float v = tap < 9 ? (float)ldexp((double)w2[(long)c * 9 + tap], log2s) : 0.f;
                    v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    packed[(((long)blk * 2 + jp) * 2 + 0) * 512 + lane * 8 + e] = hi;
                    packed[(((long)blk * 2 + jp) * 2 + 1) * 512 + lane * 8 + e] = lo;
                }
    return CER_OK;
}

template <int WM_, int WN_, int MT, int F8>
static int sx_launch(S16Args& a, int epi, hipStream_t st) {
    constexpr int TH = WM_ * 2 * MT, HR = TH + 2;
    const size_t smem = 2 * HR * (F8 ? SX_ROWB8 : SX_ROWB) + (HR + 6) * SX_DTW * 4;
    const int tiles_y = (a.h + TH - 1) / TH;
    a.tiles_x = (a.w + SX_TW - 1) / SX_TW;
    a.mtx = a.tiles_x;
    a.mty = (a.h + 1) / 2;
    if ((long)a.mtx * a.mty >= (1L << 20)) return CER_ESHAPE;
    a.ntiles = a.tiles_x * tiles_y;
    a.ny = a.cout / (WN_ * 32);
    a.border_first = a.edge && a.wpk_c && a.nsrc > 0 && a.kind[a.nsrc - 1] == 1 && a.ny == 1 && a.tiles_x >= 3 && tiles_y >= 3;
    dim3 grid((unsigned)(a.ntiles * a.ny)), block(256);
    if (smem > 64 * 1024) {                                // (two blocks per CU still fit: 2 x 74 KiB at most)
        static bool raised[64][5];                         // per device (the attribute is a per-device property of the function)
        int dev_ = 0;
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 0;
        const void* fn = nullptr;
        switch (epi) {
            case CER_EPI_LINEAR: fn = (const void*)conv3x3_s16_kernel<WM_, WN_, MT, CER_EPI_LINEAR, F8>; break;
            case CER_EPI_RELU: fn = (const void*)conv3x3_s16_kernel<WM_, WN_, MT, CER_EPI_RELU, F8>; break;
            case CER_EPI_GRU: fn = (const void*)conv3x3_s16_kernel<WM_, WN_, MT, CER_EPI_GRU, F8>; break;
            default: break;
        }
        if constexpr (WM_ == 1) {
            if (epi == CER_EPI_GATES) fn = (const void*)conv3x3_s16_kernel<WM_, WN_, MT, CER_EPI_GATES, F8>;
            if (epi == SX_EPI_DELTA) fn = (const void*)conv3x3_s16_kernel<WM_, WN_, MT, SX_EPI_DELTA, F8>;
        }
        if (!fn || epi < 0 || epi > 4) return CER_EINVAL;
        if (!raised[dev_][epi]) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return CER_EINVAL;
            raised[dev_][epi] = true;
        }
    }
    switch (epi) {
        case CER_EPI_LINEAR: hipLaunchKernelGGL((conv3x3_s16_kernel<WM_, WN_, MT, CER_EPI_LINEAR, F8>), grid, block, smem, st, a); break;
        case CER_EPI_RELU: hipLaunchKernelGGL((conv3x3_s16_kernel<WM_, WN_, MT, CER_EPI_RELU, F8>), grid, block, smem, st, a); break;
        case CER_EPI_GRU: hipLaunchKernelGGL((conv3x3_s16_kernel<WM_, WN_, MT, CER_EPI_GRU, F8>), grid, block, smem, st, a); break;
        case CER_EPI_GATES:
            if constexpr (WM_ == 1) {
                hipLaunchKernelGGL((conv3x3_s16_kernel<WM_, WN_, MT, CER_EPI_GATES, F8>), grid, block, smem, st, a);
                break;
            } else {
                return CER_ESHAPE;
            }
        case SX_EPI_DELTA:
            if constexpr (WM_ == 1) {
                hipLaunchKernelGGL((conv3x3_s16_kernel<WM_, WN_, MT, SX_EPI_DELTA, F8>), grid, block, smem, st, a);
                break;
            } else {
                return CER_ESHAPE;
            }
        default: return CER_EINVAL;
    }
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

static int sx_num_cus() { return cer_num_cus(); }

extern "C" int cer_conv3x3_s16(const cer_conv_inputs* in, const int* log2sx, const void* packed_w, const void* packed_collapsed,
                               const void* edge_w, int log2S, const float* bias, const float* init, float* out, float* out2, const float* aux, const float* aux2, int h,
                               int w, int Cout, int epi, int log2s_out, int log2s_aux, int tile_mt, void* stream) {
    if (!in || !log2sx || !packed_w || !out || h <= 0 || w <= 0 || Cout <= 0) return CER_EINVAL;
    if (in->nsrc <= 0 || in->nsrc > CER_CONV_MAX_SRC) return CER_EINVAL;
    const int out_split = (epi & CER_EPI_OUT_SPLIT) != 0;
    const int corr_fp6 = (epi & CER_EPI_CORR_FP6) != 0;
    const int corr_fp8 = (epi & CER_EPI_CORR_FP8) != 0 || corr_fp6;       // (the FP6 form shares the fp8 form's structure and restrictions)
    if ((epi & CER_EPI_CORR_FP8) && corr_fp6) return CER_EINVAL;
    epi &= ~(CER_EPI_OUT_SPLIT | CER_EPI_AUX_SPLIT | CER_EPI_CORR_FP8 | CER_EPI_CORR_FP6);
    if (epi == CER_EPI_GATES && (!out2 || !aux || Cout != 128)) return CER_EINVAL;
    if (epi == CER_EPI_GRU && (!aux || !aux2)) return CER_EINVAL;
    if (epi == SX_EPI_DELTA && (!aux || Cout % 128 != 0)) return CER_EINVAL;
    if (Cout % 64 != 0) return CER_ESHAPE;
    if ((long)h * w >= (1L << 27)) return CER_ESHAPE;
    if (!cer_aligned16(packed_w) || !cer_aligned16(packed_collapsed) || !cer_aligned16(edge_w) || !cer_aligned16(init) || !cer_aligned16(bias) || !cer_aligned16(out) ||
        !cer_aligned16(out2) || !cer_aligned16(aux) || !cer_aligned16(aux2))
        return CER_EALIGN;
    S16Args a;
    memset(&a, 0, sizeof(a));
    int order[CER_CONV_MAX_SRC];
    const int nd = sx_order(in->kind, in->nsrc, order);
    if (nd > 1) return CER_ESHAPE;
    a.nsrc = in->nsrc;
    for (int oi = 0; oi < in->nsrc; ++oi) {
        const int s = order[oi];
        if (!in->src[s]) return CER_EINVAL;
        if (in->kind[s] != 1 && in->kind[s] != 2) return CER_EINVAL;
        if (in->kind[s] == 2 && (in->ch[s] % 32 != 0 || !cer_aligned16(in->src[s]))) return CER_ESHAPE;
        if (in->kind[s] == 1 && in->ch[s] != 49) return CER_ESHAPE;
        a.src[oi] = reinterpret_cast<const char*>(in->src[s]);
        a.ch[oi] = in->ch[s];
        a.kind[oi] = in->kind[s];
        if (in->kind[s] == 1) a.disp_scale = ldexpf(1.0f, log2sx[s]);
    }
    a.wpk = (const _Float16*)packed_w;
    a.wpk_c = (const _Float16*)packed_collapsed;
    a.edge = packed_collapsed ? edge_w : nullptr;
    a.bias = bias;
    a.init = init;
    a.out = out;
    a.out2 = out2;
    a.aux = aux;
    a.aux2 = aux2;
    a.h = h;
    a.w = w;
    a.cout = Cout;
    a.S = ldexpf(1.0f, log2S);
    a.invS = ldexpf(1.0f, -log2S);
    a.out_scale = ldexpf(1.0f, log2s_out);
    a.aux_inv = ldexpf(1.0f, -log2s_aux);
    a.proj_inv = ldexpf(1.0f, -(SX_HID_LOG2 + log2s_aux));         // DELTA: log2s_aux carries the projection weights' scale
    a.out_split = out_split;
    a.flag = cer_overflow_flag_get();
    hipStream_t st = (hipStream_t)stream;
    const int ncu = sx_num_cus();
    // Tile height: blocks are 4 waves, two of them fit a CU.  A launch costs rounds x (tile rows + a fixed prologue / epilogue
    // share worth ~1.1 m-tile rows per wave); images too small to fill the CUs with full-height tiles (the row slabs of the
    // multi-GPU forward: 51 x 400 at 8 ranks) take the half-height tiles.  tile_mt forces a height (experiments, tests).
    // (10-row tiles - fewer idle CU-rounds at 296 x 400 - spill up to 240 registers and are not built: VERDICT r2)
    const long slots = 2L * ncu;
    const int tx = (w + SX_TW - 1) / SX_TW;
    auto pick = [&](int rows_per_mt, int lo, int hi, int ny) {
        int best_mt = hi;
        double best = -1.0;
        for (int c = hi; c >= lo; --c) {
            const long nblk = (long)((h + rows_per_mt * c - 1) / (rows_per_mt * c)) * tx * ny;
            const double cost = (double)((nblk + slots - 1) / slots) * (c + 1.1);
            if (best < 0 || cost < best - 1e-9) { best = cost; best_mt = c; }
        }
        return best_mt;
    };
#ifdef SX_PROBE          // probe builds (tools/archive/r05/mkprobe.sh): ONE kernel configuration - the 64-output-channel fp8-correction form - compiled in seconds
    return sx_launch<2, 2, 2, 1>(a, epi, st);
#else
    // the fp8-correction kernels evaluate a disparity source in the collapsed form with the rim correction only
    if (corr_fp8 && nd == 1 && !(packed_collapsed && edge_w)) return CER_ESHAPE;
#if CER_WITH_SXPC
    // variants/libcermvs_optin.so only (round 4's producer / consumer form, experimental/conv_s16pc.hip: measured slower): taken
    // when its switch cer_conv3x3_s16_pc is on and no tile height is forced.  It gets a COPY of the arguments (it edits the tiling fields
    // before it can refuse a shape).
    if (corr_fp8 && tile_mt == 0) {
        S16Args b = a;
        const int rc = sxpc_dispatch(b, epi, tile_mt, st);
        if (rc != CER_ESHAPE) return rc;
    }
#endif
    if (Cout % 128 == 0) {
        int mt = tile_mt;
#if SX_MT8
        // experiment (round 6): ONE block per CU, eight m-tiles (16 rows) per wave - a weight slice is fetched once per CU and tap instead of twice
        if (mt == 8 && corr_fp6) return sx_launch<1, 4, 8, 2>(a, epi, st);
        if (mt == 8 && corr_fp8) return sx_launch<1, 4, 8, 1>(a, epi, st);
#endif
        if (mt != 2 && mt != 4) mt = pick(2, 2, 4, Cout / 128) == 2 ? 2 : 4;
        if (corr_fp6) return mt == 2 ? sx_launch<1, 4, 2, 2>(a, epi, st) : sx_launch<1, 4, 4, 2>(a, epi, st);
        if (corr_fp8) return mt == 2 ? sx_launch<1, 4, 2, 1>(a, epi, st) : sx_launch<1, 4, 4, 1>(a, epi, st);
        return mt == 2 ? sx_launch<1, 4, 2, 0>(a, epi, st) : sx_launch<1, 4, 4, 0>(a, epi, st);
    }
    if (epi == SX_EPI_DELTA || epi == CER_EPI_GATES) return CER_ESHAPE;
    int mt = tile_mt;
    if (corr_fp8) {
        // (the 32-channel chunk buffers of a 16-row tile - 2 x 46 KiB - would leave room for one block per CU: 12 rows at most)
        if (mt < 2 || mt > 3) mt = pick(4, 2, 3, Cout / 64);
        if (corr_fp6) return mt == 2 ? sx_launch<2, 2, 2, 2>(a, epi, st) : sx_launch<2, 2, 3, 2>(a, epi, st);
        return mt == 2 ? sx_launch<2, 2, 2, 1>(a, epi, st) : sx_launch<2, 2, 3, 1>(a, epi, st);
    }
    if (mt < 2 || mt > 4) mt = pick(4, 2, 4, Cout / 64);
    return mt == 2 ? sx_launch<2, 2, 2, 0>(a, epi, st) : mt == 3 ? sx_launch<2, 2, 3, 0>(a, epi, st) : sx_launch<2, 2, 4, 0>(a, epi, st);
#endif
}

// ---- layout conversion (model load / API boundaries / tests): plain fp32 [h*w, C] <-> the m-tile-major layouts of the s16 convs.
// An m-tile = 2 rows x 16 columns of pixels, lane order li = (y & 1) * 16 + (x & 15); tensors hold ceil(h/2) * ceil(w/16) m-tiles.
//   layout 0 "frag16": per (m-tile, 16-channel group): 1 KiB of hi halves | 1 KiB of lo halves of x * 2^log2s; inside a plane
//            piece (kg = (c >> 3) & 1, li) holds channels 8kg .. 8kg+7: the B-fragment order of v_mfma_f32_32x32x16_f16;
//   layout 1 "acc32":  fp32, per (m-tile, 32-channel tile, j < 4): 1 KiB, lane (kg, li) holds channels 8j + 4kg + 0..3 (the MFMA
//            accumulator order; `init` of cer_conv3x3_s16 and its LINEAR output);
//   layout 2 "f32x8":  fp32, per (m-tile, 16-channel group, q < 2): 1 KiB, lane (kg, li) holds channels 8kg + 4q + 0..3 (the GATES
//            epilogue's z).
//   layout 3: frag16 pieces copied verbatim (no arithmetic): the plain side holds, per pixel and 8 channels, hi 16 B | lo 16 B -
//            used to move image rows of a frag16 tensor (the row-slab halo exchange, slab.py).
// The plain side covers image rows [y0, y0 + nrows) only.
__global__ __launch_bounds__(256) void s16_layout_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int C, int layout,
                                                         float scale, float inv, int inverse, int y0, int nrows) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per (pixel, 8 consecutive channels)
    const long n8 = (long)nrows * w * (C / 8);
    if (i >= n8) return;
    const long pix = i / (C / 8);
    const int cb = (int)(i - pix * (C / 8)) * 8;
    const int y = y0 + (int)(pix / w), x = (int)(pix % w);
    const long mt = (long)(y >> 1) * ((w + 15) >> 4) + (x >> 4);
    const int li = ((y & 1) << 4) | (x & 15);
    char* t = reinterpret_cast<char*>(inverse ? const_cast<float*>(src) : dst);
    float* plain = (inverse ? dst : const_cast<float*>(src)) + pix * C + cb;
    char *p0, *p1;                                         // the two 16-byte pieces of this thread
    if (layout == 0 || layout == 3) {
        p0 = t + ((mt * (C >> 4) + (cb >> 4)) * 2) * 1024 + ((((cb >> 3) & 1) << 5) | li) * 16;
        p1 = p0 + 1024;
    } else if (layout == 1) {
        p0 = t + ((mt * (C >> 5) + (cb >> 5)) * 4 + ((cb & 31) >> 3)) * 1024 + li * 16;
        p1 = p0 + 512;                                     // kg = 1
    } else {
        p0 = t + ((mt * (C >> 4) + (cb >> 4)) * 2) * 1024 + ((((cb >> 3) & 1) << 5) | li) * 16;
        p1 = p0 + 1024;
    }
    float v[8];
    if (!inverse) {
        const float4 t0 = cer_ld4(plain), t1 = cer_ld4(plain + 4);
        v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
        if (layout == 0) {
            half8 hi, lo;
            sx_split8(v, scale, hi, lo);
            *reinterpret_cast<half8*>(p0) = hi;
            *reinterpret_cast<half8*>(p1) = lo;
        } else {
            *reinterpret_cast<float4*>(p0) = t0;
            *reinterpret_cast<float4*>(p1) = t1;
        }
    } else {
        if (layout == 0) {
            sx_join8(*reinterpret_cast<const half8*>(p0), *reinterpret_cast<const half8*>(p1), inv, v);
            *reinterpret_cast<float4*>(plain) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(plain + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            *reinterpret_cast<float4*>(plain) = *reinterpret_cast<const float4*>(p0);
            *reinterpret_cast<float4*>(plain + 4) = *reinterpret_cast<const float4*>(p1);
        }
    }
}

extern "C" long cer_s16_padded_pixels(int h, int w) { return (h <= 0 || w <= 0) ? CER_EINVAL : (long)((h + 1) / 2) * ((w + 15) / 16) * 32; }

// ---- saturation scan of a split-f16 tensor (frag16 / split rows): any half at the f16 maximum (the clamp of the split) or not finite
__global__ __launch_bounds__(256) void f16_scan_kernel(const uint4* __restrict__ p, long n16, int* __restrict__ flag, int bit) {
    bool hit = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const uint4 q = p[i];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) hit |= ((w[j] & 0x7FFFu) >= 0x7BFFu) || ((w[j] & 0x7FFF0000u) >= 0x7BFF0000u);
    }
    if (__ballot(hit) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}
extern "C" int cer_f16_scan_overflow(const void* data, long bytes, int* flag, int bit, void* stream) {
    if (!data || !flag || bytes <= 0 || bytes % 16 != 0) return CER_EINVAL;
    if (!cer_aligned16(data)) return CER_EALIGN;
    const long n16 = bytes / 16;
    long blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(f16_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)data, n16, flag, bit);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

extern "C" int cer_s16_layout_f32(const float* src, float* dst, int h, int w, int C, int layout, int log2s, int inverse, void* stream) {
    if (!src || !dst || h <= 0 || w <= 0 || C <= 0 || layout < 0 || layout > 2) return CER_EINVAL;
    if (C % (layout == 1 ? 32 : 16) != 0) return CER_ESHAPE;
    if (!cer_aligned16(src) || !cer_aligned16(dst)) return CER_EALIGN;
    const long n8 = (long)h * w * (C / 8);
    hipLaunchKernelGGL(s16_layout_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, w, C, layout,
                       ldexpf(1.0f, log2s), ldexpf(1.0f, -log2s), inverse, 0, h);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}

// image rows [y0, y0 + nrows) of a frag16 tensor (h x w image, C channels) -> rows [nrows * w, C] (to_tensor = 0) or back (= 1);
// bit-exact copies of the hi | lo pieces (the row-slab halo exchange)
extern "C" int cer_s16_rows_f32(float* tensor, float* rows, int h, int w, int C, int y0, int nrows, int to_tensor, void* stream) {
    if (!tensor || !rows || h <= 0 || w <= 0 || C <= 0 || y0 < 0 || nrows <= 0 || y0 + nrows > h) return CER_EINVAL;
    if (C % 16 != 0) return CER_ESHAPE;
    if (!cer_aligned16(tensor) || !cer_aligned16(rows)) return CER_EALIGN;
    const long n8 = (long)nrows * w * (C / 8);
    if (to_tensor)
        hipLaunchKernelGGL(s16_layout_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, tensor, w, C, 3, 1.0f, 1.0f, 0, y0, nrows);
    else
        hipLaunchKernelGGL(s16_layout_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tensor, rows, w, C, 3, 1.0f, 1.0f, 1, y0, nrows);
    CER_RETURN_IF_LAUNCH_FAILED();
    return CER_OK;
}
