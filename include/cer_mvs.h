/*
 * cer_mvs.h - C ABI of libcermvs.so: the MI355X (gfx950) kernels behind the CER-MVS
 * depth-inference hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor's data_ptr());
 *     tensors are dense, row-major, float32 unless stated; nothing is allocated inside
 *     (workspace is passed in) and nothing is retained after return;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); the call only
 *     enqueues work on it and returns (asynchronous), so it is hipGraph-capturable;
 *   - return 0 on success, a negative CER_E* for an argument error (nothing was launched),
 *     or a positive hipError_t from the launch; never throws.  Compute entry points keep no state between calls and are
 *     re-entrant; the library holds exactly two pieces of process-wide state, both set through their own entry points and read by later
 *     launches: the per-device overflow flag (cer_overflow_flag) and the cost-volume algorithm selector (cer_cost_build_algo).
 *     (The opt-in kernel forms of round 4 and their switches live in variants/libcermvs_optin.so only: include/cer_mvs_variants.h.)
 *
 * Each function cites the reference interface (file:line under the reference tree) it
 * replaces.  INTEGRATION.md shows the binding a reference maintainer would add.
 */
#ifndef CER_MVS_H
#define CER_MVS_H

#ifdef __cplusplus
extern "C" {
#endif

#define CER_OK 0
#define CER_EINVAL (-1)   /* null pointer / non-positive size                       */
#define CER_ESHAPE (-2)   /* shape not supported by this build (e.g. C % 64 != 0)   */
#define CER_EALIGN (-3)   /* pointer not 16-byte aligned                            */

/* ABI version: major*1000 + minor.  Bumped on any signature change. */
int cer_abi_version(void);
/* Static description of the last-resort error codes above / hipError_t names. */
const char* cer_error_string(int code);
/* Number of visible HIP devices (>=1) or a negative CER_E*; used by loaders to fail loudly. */
int cer_device_count(void);

/* Saturation is never silent.  The split-f16 kernels clamp an operand whose scaled value leaves the f16 range (update block,
 * "s16" path: ReLU-class activations beyond 4094 = 65504 / 2^4; cost-volume feature rows beyond 1023).  cer_overflow_flag
 * registers a caller-owned DEVICE int for the CURRENT device (hipGetDevice; one per device, NULL = off, the default; launches use
 * the flag of the device they run on) that such kernels or into when they clamp in
 * registers (bit 2: the hidden map of the fused delta head, which never reaches memory).  Tensors that do reach memory are
 * checked after the fact: cer_f16_scan_overflow ors `bit` into *flag if any half of a split-f16 buffer (frag16 tensors, split
 * feature rows; `bytes` % 16 == 0) sits at the f16 maximum or is not finite.  The host reads the flag when convenient. */
int cer_overflow_flag(int* flag);
int cer_f16_scan_overflow(const void* data, long bytes, int* flag, int bit, void* stream);

/* ------------------------------------------------------------------------------------
 * alt_cuda_corr.forward  (reference: alt_cuda_corr/correlation.cpp:23-33,
 * correlation_kernel.cu:18-119,260-286).  Literal semantics, any radius:
 *   corr[b,n,ky+rd*kx,h,w] = sum_c fmap1[b,h,w,c] * bilerp(fmap2[b], x-r+kx, y-r+ky),
 *   (x,y) = coords[b,n,h,w,:], rd = 2r+1, texels outside fmap2 read as 0.
 * fmap1 [B,H1,W1,C]  fmap2 [B,H2,W2,C]  coords [B,N,H1,W1,2]  corr [B,N,rd*rd,H1,W1].
 * corr is fully overwritten (the reference zero-fills then accumulates, :273).  C % 64 == 0, C <= 256.
 */
int cer_alt_corr_forward_f32(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                             int B, int N, int H1, int W1, int H2, int W2, int C, int radius, void* stream);

/* alt_cuda_corr.backward (correlation.cpp:36-48, correlation_kernel.cu:122-256,288-324), radius 0:
 * fmap1_grad [B,H1,W1,C] and fmap2_grad [B,H2,W2,C] are overwritten (fmap2_grad is zero-filled
 * inside, then accumulated with atomics); coords_grad is zero-filled, as in the reference (:307).
 */
int cer_alt_corr_backward_f32(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                              float* fmap1_grad, float* fmap2_grad, float* coords_grad,
                              int B, int N, int H1, int W1, int H2, int W2, int C, int radius, void* stream);
/* Deterministic fmap2 gradient (no float atomics).  cer_alt_corr_backward_f32 with fmap2_grad = NULL
 * gives fmap1_grad only; then
 *   cer_alt_corr_bwd_tuples_f32: per (sample, footprint texel) a key (b*H2*W2 + texel, or the sentinel B*H2*W2 for texels
 *     outside the map), a coefficient and the source pixel (b*H1*W1 + p): arrays of B*N*H1*W1*(2r+2)^2 entries;
 *   the caller sorts the keys stably (`order` = the permutation) and computes seg[t] = first sorted position with key >= t,
 *     t = 0 .. B*H2*W2 (T + 1 entries);
 *   cer_alt_corr_bwd_reduce_f32: fmap2_grad[t, :] = sum of coef * fmap1[src, :] over segment t, in sorted order. */
int cer_alt_corr_bwd_tuples_f32(const float* coords, const float* corr_grad, long* keys, float* coef, int* src,
                                int B, int N, int H1, int W1, int H2, int W2, int radius, void* stream);
int cer_alt_corr_bwd_reduce_f32(const float* fmap1, const long* order, const float* coef, const int* src, const long* seg,
                                float* fmap2_grad, long T, int C, void* stream);

/* ------------------------------------------------------------------------------------
 * Epipolar cost-volume build, one cascade stage (reference: CorrBlock.__init__
 * core/corr.py:56-91, projective_transform utils/projective_ops.py:16-28, direct_corr
 * core/corr.py:28-43 and the kernel above), fused: no coordinate tensors are materialised.
 *
 *   origin[p]   = shift ? (disp_in[p] < (D/2)*incre ? (D/2)*incre : disp_in[p]) : disp_in[p]
 *   hyp[p,k]    = (float)(k - D/2) * incre + origin[p],                       k = 0..D-1
 *   (X,Y,Z,_)   = Pij[v] * (x, y, 1, hyp[p,k]);  (u,w) = clamp((X/Z, Y/Z), +-1e4)
 *   c[v,p,k]    = sum_c fmap1[p,c] * bilerp(fmap2[v], u, w)      (texels outside -> 0;
 *                                                                 non-finite (u,w) -> 0)
 * fmap1 [h1*w1, C] and fmap2 [V, h2+4, w2+4, C] are NHWC and already carry the reference's 1/8
 * scaling (core/corr.py:30-31); fmap2 has a 2-texel ZERO BORDER on every side (texel (y,x) of view v at
 * ((v*(h2+4) + y+2)*(w2+4) + x+2)*C) - the kernel clamps cells into the border instead of testing bounds.
 * Pij [V,4,4] row-major = K_j P_j P_i^-1 K_i^-1.
 *
 * Output rows have `row_stride` floats (>= D, multiple of 4); only columns [0,D) are written
 * here (cer_pyramid_f32 fills the pooled levels behind them).
 *   mode 0: vol [V, P, row_stride], vol[v,p,k] = c[v,p,k]               (per-view, literal)
 *   mode 1: vol [P, row_stride],    vol[p,k]   = sum_v c[v,p,k]         (view-sum fold)
 *   mode 2: as mode 1 but adds into the existing vol (accumulate across calls)
 * (h1, w1) is the reference-pixel grid this call covers and y0 the image row of its first row: a multi-GPU rank that owns
 * a row slab passes its slab height and first row (fmap1 / disp_in / vol then hold only those rows); y0 = 0 otherwise.
 * fuse_levels >= 1 (mode 1, D <= 64 only; else CER_EINVAL): the epilogue also scales level 0 by fuse_scale (1/V: the view mean)
 * and writes the avg-pooled levels 1..fuse_levels-1 behind it - what cer_pyramid_f32(vol, ..., fuse_levels, fuse_scale) would do in
 * a second pass (core/corr.py:94-97).  fuse_levels = 1: level 0 only, SCALED (the level-0-only rows the lookup pools on the fly);
 * fuse_levels = 0: level 0 only, unscaled, fuse_scale ignored, mode 2 allowed.  The same meaning in all three builders
 * (cer_cost_build_f32, cer_cost_lines_f32, cer_cost_lines_reduce_f32) since ABI 1060: before, this entry point treated 1 like 0.
 * origin_out [P] may be NULL.  C % 64 == 0.  `incre` is the reference's Python float (core/raft.py:81): the kernel
 * uses (float)incre for the hypothesis spacing and (float)((D/2)*incre) for the shift limit, as torch does.
 */
int cer_cost_build_f32(const float* fmap1, const float* fmap2, const float* Pij, const float* disp_in,
                       float* vol, float* origin_out,
                       int V, int h1, int w1, int h2, int w2, int C, int D, int row_stride,
                       double incre, int shift, int mode, int y0, int fuse_levels, float fuse_scale, void* stream);
/* Implementation choice of the fold modes (1, 2) of cer_cost_build_f32 when called through the host layer: 0 = automatic
 * (cer_cost_lines_f32 below wherever it applies, else the walk), 1 = always the wave-per-pixel walk.  Returns the previous
 * setting.  (Process-wide switch for tests and A/B timing; the library's entry points themselves are stateless.) */
int cer_cost_build_algo(int algo);

/* ------------------------------------------------------------------------------------
 * The same cost volume (fold modes 1 / 2 of cer_cost_build_f32; C == 64, D <= 64) on epipolar-line tiles
 * (csrc/cost_lines.hip): per source view the reference grid is cut into 32-pixel digital lines along the view's
 * epipolar direction; per (view, tile) the dot products of the tile's reference rows with every texel of the tile's
 * epipolar band are MFMA products (split-f16, fp32-class) and every sample gathers its 4 dots from LDS.  Cells, weights
 * and the treatment of out-of-map / non-finite samples are those of cer_cost_build_f32 (same fp32 expressions); results
 * agree to fp32 rounding of the 64-channel dot.
 *
 * cer_feat_split_f16: fp32 rows [blocks, block_texels, 64] -> split-f16 operand planes, same bytes: per block 8 planes
 *   p = hl * 4 + ks (hl 0 / 1: hi = f16(xs) / lo = f16(xs - hi) of xs = x * 2^6; ks: 16-channel group), each [block_texels][16];
 *   a block = one view (block_texels = (h2+4)*(w2+4), zero border included and kept zero) or the reference map (h1*w1).
 *   |x| > 1023 saturates: *overflow_flag (device int, may be NULL) is or-ed with 1.
 * cer_cost_lines_workspace: bytes of `workspace` (per-view partial volumes [V,P,D] + tile parameters + the hand-over list below).
 * cer_cost_lines_f32: arguments as cer_cost_build_f32 with the split rows in place of fmap1 / fmap2; mode 1 or 2 only.
 *   view_slot (device int [V], may be NULL = identity): view v's rows are block view_slot[v] of fmap2_split - the multi-GPU
 *   forward all-gathers every rank's split rows into one [G, ceil(V/G), ...] buffer and builds from it without a reordering copy.
 */
int cer_feat_split_f16(const float* src, void* dst, long blocks, long block_texels, int C, int* overflow_flag, void* stream);
long cer_cost_lines_workspace(int V, int h1, int w1, int D);
int cer_cost_lines_f32(const void* fmap1_split, const void* fmap2_split, const int* view_slot, const float* Pij, const float* disp_in,
                       float* vol, float* origin_out, void* workspace,
                       int V, int h1, int w1, int h2, int w2, int C, int D, int row_stride,
                       double incre, int shift, int mode, int y0, int fuse_levels, float fuse_scale, int two_term, void* stream);
/* two_term (round 6, ABI 1070; 0 or 1, else CER_EINVAL): 0 = three-term split-f16 dots, fp32-class (4e-8 relative L1 from the walk); 1 = the lo planes
 * of fmap2_split are NOT READ - the source features enter the dots as f16 (the reference rows keep both halves): half the bytes of the band-fragment
 * stream that bounds the tile kernel (stage 0 at the bench workload 1.58 -> 1.02 ms), 1.2e-5 relative L1 on the volume, ~5e-6 on the final disparity.
 * RAFT's "auto" calibration decides per set of weights ("+c2" forms); the planes keep their layout either way.
 * The two halves of cer_cost_lines_f32, for callers that build views as their features become available (RAFT.forward encodes the
 * source views in batches and builds each batch's partial volumes on a second stream while the next batch is being encoded):
 * _views_ writes the partial volumes of views v0 .. v0 + nv - 1 into the V-view workspace (Pij, view_slot, fmap2_split indexed by
 * the absolute view number); _reduce_ sums all V partials in view order into vol (origin, scale, pooled levels as above). */
int cer_cost_lines_views_f32(const void* fmap1_split, const void* fmap2_split, const int* view_slot, const float* Pij, const float* disp_in,
                             void* workspace, int V, int v0, int nv, int h1, int w1, int h2, int w2, int C, int D,
                             double incre, int shift, int y0, int two_term, void* stream);
int cer_cost_lines_reduce_f32(const void* workspace, const float* disp_in, float* vol, float* origin_out, int V, int h1, int w1, int D,
                              int row_stride, double incre, int shift, int mode, int fuse_levels, float fuse_scale, void* stream);

/* Correlation pyramid (reference: core/corr.py:94-97, F.avg_pool2d([1,2]) x (L-1)), in place on
 * rows laid out [level0 (D) | level1 (D/2) | level2 (D/4) | ... | pad]: first level0 *= scale
 * (1/V for the view-mean fold, 1 otherwise), then each level = pairwise mean of the previous.
 * vol [rows, row_stride].
 */
int cer_pyramid_f32(float* vol, long rows, int D, int row_stride, int num_levels, float scale, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-level correlation lookup (reference: CorrBlock.__call__ core/corr.py:102-143 with
 * bilinear_sampler1 utils/bilinear_sampler.py:6-25):
 *   c        = max((disp[p] - origin[p]) / incre + D/2, 0)
 *   out[v, i*(2r+1) + (dx+r), p] = lerp(level_i[v,p,:], c/2^i + dx)   (zero outside the row)
 * vol [nv, P, row_stride] (nv = V per-view, or 1 for the folded volume); out [nv, L*(2r+1), P].
 */
int cer_corr_lookup_f32(const float* vol, const float* origin, const float* disp, long disp_view_stride,
                        float* out, int nv, long P, int D, int row_stride, double incre, int num_levels, int radius,
                        int level0_only /* see cer_lookup_encode_f32 */, void* stream);
/* disp is [P] shared by all views (disp_view_stride = 0; RAFT.forward passes V identical copies,
 * core/raft.py:99) or [nv, P] (disp_view_stride = P). */

/* View aggregation "mean" + first corr_encoder layer on already looked-up features (reference:
 * core/update.py:103,61-62): out[p, co] = relu(b[co] + sum_k w[k, co] * mean_v feats[v, k, p]).
 * feats [nv, K, P] planar (the CorrBlock.__call__ layout), w [K, Cout], out [P, Cout]; Cout == 64. */
int cer_corr_encode_f32(const float* feats, const float* w, const float* b, float* out,
                        int nv, int K, long P, int Cout, void* stream);

/* Same lookup on the folded (view-mean) volume fused with the first corr_encoder layer
 * (reference: core/update.py:103 mean, :61-62 Conv2d(33,64,1)+ReLU): out [P, Cout] NHWC.
 * w [Cin=L*(2r+1), Cout] (transposed 1x1 weight), b [Cout].  Cout == 64.
 * Rows (both lookup entry points): level0_only = 0 - the row holds the whole pyramid [level0 | level1 | ...] (row_stride >= its length,
 *   else CER_ESHAPE) and the levels are read from it; level0_only = 1 (round 5; an explicit argument since ABI 1060 - it used to be
 *   inferred from row_stride, which is ambiguous for D <= 5) - the row holds LEVEL 0 ONLY (row_stride >= D) and level l is formed on the
 *   fly as the pairwise means ((a + b) * 0.5 level by level, core/corr.py:94-97) in the association cer_pyramid_f32 uses: bit-identical
 *   values, 43 % fewer bytes read (RAFT.forward builds its folded volume that way: cer_cost_lines_reduce_f32 with fuse_levels = 1);
 *   at most 4 levels in that form (CER_ESHAPE beyond).
 * delta_taps (may be NULL): the PREVIOUS GRU iteration's disparity update rides on this launch - core/update.py:114, core/raft.py:101:
 *   disp[p] += 0.01 * (delta_bias + sum over delta_nhalf (1 or 2) x 9 tap planes T[half][tap][p + (ky-1, kx-1)], zero outside the
 *   image), i.e. cer_delta_sum_f32 (same summation order: bit-identical), before the lookup of pixel p reads it; `disp` is then
 *   updated IN PLACE (img_w must be given: rows of P / img_w pixels).  With NULL `disp` is only read.
 */
int cer_lookup_encode_f32(const float* vol, const float* origin, float* disp,
                          const float* w, const float* b, float* out,
                          long P, int D, int row_stride, double incre, int num_levels, int radius, int Cout,
                          int out_split /* 0: fp32 [P, Cout]; 1: split32 layout (f16x3 convs); 2: frag16 layout of an image
                                           img_w pixels wide with scale 2^log2s_out (s16 convs) - both below */,
                          int log2s_out, int img_w, const float* delta_taps, int delta_nhalf, float delta_bias, int level0_only,
                          void* stream);

/* ------------------------------------------------------------------------------------
 * 3x3, stride 1, zero-padded convolution as an implicit GEMM on exact-fp32 MFMA
 * (v_mfma_f32_16x16x4_f32), channels-last.  This one kernel family carries every 3x3 of the
 * update block (reference: core/update.py:13-15,63-64,69-71).
 *
 * The K dimension is the concatenation of up to CER_CONV_MAX_SRC sources; source s is
 *   kind 0: a tensor [h*w, ch[s]]                              (ch multiple of 16)
 *   kind 1: the disparity encoder of disp [h*w] (update.py:80-85,97): 49 channels
 *           100*(disp0[y+uy-3, x+ux-3] - disp[y,x]) (disp0 = zero-padded disp), padded to 64
 *   kind 3: (cer_conv3x3_f16x3 only) a tensor [h*w, ch[s]] in the split32 layout described below (ch multiple of 32);
 *           the tensor sources of one call are all kind 0 or all kind 3; weights are packed as for kind 0
 * Weights are pre-packed by cer_conv3x3_pack_f32.  acc is initialised from `init` [h*w, Cout]
 * when non-NULL, else from bias [Cout] (or 0).  Epilogues (epi):
 *   0 LINEAR  out[p,co]  = acc
 *   1 RELU    out[p,co]  = max(acc, 0)
 *   2 GATES   co <  Cout/2: out [p,co] = sigmoid(acc)                (z)
 *             co >= Cout/2: out2[p,co-Cout/2] = sigmoid(acc) * aux[p,co-Cout/2]   (r * net)
 *   3 GRU     q = tanh(acc); out[p,co] = (1 - aux2[p,co]) * aux[p,co] + aux2[p,co] * q
 *             (aux = net, aux2 = z; update.py:23-24)
 */
#define CER_CONV_MAX_SRC 4
#define CER_EPI_LINEAR 0
#define CER_EPI_RELU 1
#define CER_EPI_GATES 2
#define CER_EPI_GRU 3
/* "split32" activation layout (f16x3 kernels only): a [P, C] tensor (C % 32 == 0) whose fp32 slots hold, per pixel and
 * 32-channel chunk, 32 hi halves followed by 32 lo halves (x = hi + 2^-11 lo; same bytes as fp32).  A producer epilogue
 * writes it with CER_EPI_OUT_SPLIT or'ed into `epi` (RELU / LINEAR: out; GATES: out2 = r*h; GRU: out), a consumer reads it
 * as source kind 3 (staging = plain 16-byte copies, no conversion arithmetic) or, with CER_EPI_AUX_SPLIT, as the previous
 * hidden state `aux` of the GATES / GRU epilogues (reconstructed as hi + 2^-11 lo).  cer_split32_f32 converts. */
#define CER_EPI_OUT_SPLIT 0x100
#define CER_EPI_AUX_SPLIT 0x200
/* cer_conv3x3_s16 only: the weights were packed with `collapsed | 2` and the two correction terms of the split-f16 product of the
 * TENSOR sources (xh*wl + xl*wh, 2^-11 of the main term) run on the block-scaled fp8 matrix instruction (twice the f16 rate). */
#define CER_EPI_CORR_FP8 0x400
/* cer_conv3x3_s16 only (round 6): the weights were packed with `collapsed | 4` and the same two correction terms run on the FP6 (e2m3) form of
 * that instruction - half its matrix-pipe passes again - with one E8M0 scale per K block of 16 channels x [hi | lo] on either operand
 * (activations: per pixel, chosen while the halo tile is staged; weights: per output channel and tap, chosen by the packer). */
#define CER_EPI_CORR_FP6 0x800
#define CER_EPI_DELTA 4   /* cer_conv3x3_f16x3 only - see cer_delta_proj_pack */

typedef struct {
    const float* src[CER_CONV_MAX_SRC];
    int ch[CER_CONV_MAX_SRC];     /* logical channels of each source (49 for kind 1) */
    int kind[CER_CONV_MAX_SRC];
    int nsrc;
} cer_conv_inputs;

/* Number of floats cer_conv3x3_pack_f32 writes for (Cout, padded K channels). */
long cer_conv3x3_packed_size(int Cout, int Kpad);
/* Pack OIHW weights [Cout, Cin, 3, 3] whose Cin is the concatenation described by (ch, kind)
 * into the MFMA B-fragment order.  HOST pointers (done once at model load). */
int cer_conv3x3_pack_f32(const float* w_oihw, float* packed, int Cout, int Cin,
                         const int* ch, const int* kind, int nsrc);

int cer_conv3x3_f32(const cer_conv_inputs* in, const float* packed_w, const float* bias, const float* init,
                    float* out, float* out2, const float* aux, const float* aux2,
                    int h, int w, int Cout, int epi, void* stream);

/* Same convolution on the f16 matrix cores with fp32-equivalent accuracy ("f16x3": every fp32 operand is
 * split x = f16(x) + 2^-11 * f16((x - f16(x)) * 2^11); three v_mfma_f32_32x32x16_f16 per tile into two fp32
 * accumulators; the dropped lo*lo term is 2^-22 relative).  Same arguments and epilogues as cer_conv3x3_f32;
 * weights packed by cer_conv3x3_f16x3_pack (size in 2-byte halves from cer_conv3x3_f16x3_packed_size;
 * Kpad = sum over sources of channels rounded up to 32, 64 for kind 1).  Cout % 64 == 0. */
long cer_conv3x3_f16x3_packed_size(int Cout, int Kpad);
int cer_conv3x3_f16x3_pack(const float* w_oihw, void* packed, int Cout, int Cin,
                           const int* ch, const int* kind, int nsrc);
/* `packed_collapsed` (may be NULL): a second packing of the same weights in which a kind-1 source is ONE 81-tap
 * filter on the raw disparity (the 3x3 conv over the 49 unfold features of core/update.py:80-85,97 is linear in
 * the disparity: W9[s] = sum_{t+u=s} w[u][t] minus the centre terms).  Tiles whose pixels all have their 3x3
 * neighbourhood inside the image use it (3 MFMA steps instead of 18 for that source); border tiles keep the
 * literal form.  Built by cer_conv3x3_f16x3_pack_collapsed, size (halves) from cer_conv3x3_f16x3_collapsed_size. */
long cer_conv3x3_f16x3_collapsed_size(int Cout, const int* ch, const int* kind, int nsrc);
int cer_conv3x3_f16x3_pack_collapsed(const float* w_oihw, void* packed, int Cout, int Cin,
                                     const int* ch, const int* kind, int nsrc);
int cer_conv3x3_f16x3(const cer_conv_inputs* in, const void* packed_w, const void* packed_collapsed,
                      const float* bias, const float* init,
                      float* out, float* out2, const float* aux, const float* aux2,
                      int h, int w, int Cout, int epi, void* stream);

/* ---- round 2: the same convolutions, "s16" form (csrc/conv_s16.hip) - the update block's fast path.
 * Arithmetic: every fp32 operand x is carried as the two f16 halves of xs = x * 2^k (k a per-tensor power of two):
 * hi = f16(xs), lo = f16(xs - hi) (unscaled residual), and x*w ~= (xh*wh + xh*wl + xl*wh) / (2^kx 2^kw): three
 * v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator per tile (fp32-class: measured rms error 1.6e-8 of sum|x||w| at
 * K = 1248, an fp32 fmaf chain has 2.7e-8).  All sources of a conv share the product scale 2^log2S = 2^kx(src) 2^kw(src).
 *
 * Tensor layouts (m-tile-major, so that every wave-level access of the kernels is one contiguous KiB).  An m-tile is 2 rows x
 * 16 columns of pixels, lane order li = (y & 1) * 16 + (x & 15); a tensor of an h x w image holds ceil(h/2) * ceil(w/16)
 * m-tiles = cer_s16_padded_pixels(h, w) pixels (allocate [padded pixels, C] floats; padding pixels are never read as data).
 *   "frag16" (split activations, C % 16 == 0): per (m-tile, 16-channel group) 1 KiB of hi halves | 1 KiB of lo halves of
 *       x * 2^log2s; inside a plane the 16-byte piece (kg = (c >> 3) & 1, li) holds channels 8kg .. 8kg+7 (MFMA fragment order);
 *   "acc32"  (fp32, C % 32 == 0): per (m-tile, 32-channel tile, j < 4) 1 KiB, lane (kg, li) holds channels 8j + 4kg + 0..3 -
 *       the accumulator order: `init` of cer_conv3x3_s16 and the output of its LINEAR epilogue;
 *   "f32x8"  (fp32, C % 16 == 0): per (m-tile, 16-channel group, q < 2) 1 KiB, lane (kg, li) holds channels 8kg + 4q + 0..3 -
 *       the z gate between the GATES and GRU epilogues.
 * cer_s16_layout_f32 converts plain fp32 [h*w, C] to layout 0 | 1 | 2 (inverse = 0; log2s used by layout 0) or back.
 *
 * Sources: kind 2 = frag16 tensor (C % 32 == 0) with scale 2^log2sx[s]; kind 1 = disparity [P] fp32, plain (49 encoder
 * channels, generated in the kernel with scale 2^log2sx[s]; at most one).  `ch`/`kind` list the sources in the weights'
 * input-channel order; the packing reorders steps itself (tensors first, disparity last).
 * cer_conv3x3_s16_scale returns log2S for a weight tensor (< -1000 on error); cer_conv3x3_s16_pack builds the literal
 * (collapsed = 0) or collapsed (= 1, needs a kind-1 source; used by interior tiles) packing, size in 2-byte halves from
 * cer_conv3x3_s16_packed_size.  HOST pointers.  `collapsed | 2`: the tensor sources' steps in the fp8-correction form (for launches
 * with CER_EPI_CORR_FP8: per 32-channel tap and 32 output channels 4 KiB = f16 hi halves of the two 16-channel halves | the e4m3
 * A operand of v_mfma_scale_f32_32x32x64_f8f6f4: [lo * 2^5 | hi * 2^-6] of each half); same size.  `collapsed | 4` (round 6): the FP6 form for
 * launches with CER_EPI_CORR_FP6 (the instruction's e2m3 A operand: 32 six-bit fields [lo * 2^11 | hi] of each half over one power of two per
 * lane, whose E8M0 byte follows the 24 bytes of fields; csrc/conv_s16.hip sx_pack_chunk6); same size.  A launch with CER_EPI_CORR_FP8 / _FP6
 * and a kind-1 source needs packed_collapsed AND edge_w (the fp8 kernels evaluate the disparity source in the collapsed form with
 * the rim correction only; CER_ESHAPE otherwise); tile_mt = 4 is not available for Cout = 64 in that form (3 is used).
 * cer_conv3x3_s16: bias (fp32 [Cout], plain) or init (acc32 [., Cout]) - at most one - seed the accumulators; epilogues:
 *   LINEAR: out acc32 [., Cout];  with CER_EPI_OUT_SPLIT or'ed in, and RELU always: out frag16 with scale 2^log2s_out;
 *   GATES (Cout = 128): out = z f32x8 [., 64], out2 = r * h frag16 (2^log2s_out), aux = h frag16 (2^log2s_aux);
 *   GRU: out = new h frag16 (2^log2s_out), aux = h frag16 (2^log2s_aux; may alias out), aux2 = z f32x8;
 *   DELTA (Cout % 128 == 0): out = tap planes T [Cout/128, 9, h*w] (plain) as for cer_conv3x3_f16x3, aux = projection weights
 *   packed by cer_delta_proj_s16_pack, log2s_aux = their scale.
 * tile_mt: 0 = choose the tile height from (h, w) and the CU count; else force it (tests: 2 | 4 for Cout % 128 == 0,
 * 2 | 3 | 4 for Cout = 64; other values choose automatically). */
long cer_conv3x3_s16_packed_size(int Cout, const int* ch, const int* kind, int nsrc, int collapsed);
int cer_conv3x3_s16_scale(const float* w_oihw, int Cout, int Cin, const int* ch, const int* kind, const int* log2sx, int nsrc);
int cer_conv3x3_s16_pack(const float* w_oihw, void* packed, int Cout, int Cin, const int* ch, const int* kind, const int* log2sx,
                         int nsrc, int collapsed, int log2S);
/* edge_w (may be NULL): rim-correction filters packed by cer_conv3x3_s16_edge_pack (size in 2-byte halves from
 * cer_conv3x3_s16_edge_size; same log2sx / log2S as the weights).  With them EVERY tile evaluates the disparity source in the
 * collapsed form and pixels on the image's 1-pixel rim subtract, as a few extra MFMA steps after the main loop, what the taps
 * whose feature position lies outside the image contributed (the 3x3 conv zero-pads the feature map, core/update.py:80-85);
 * without them border tiles run the literal form (36 steps instead of 6 for that source). */
long cer_conv3x3_s16_edge_size(int Cout);
int cer_conv3x3_s16_edge_pack(const float* w_oihw, void* out, int Cout, int Cin, const int* ch, const int* kind, const int* log2sx,
                              int nsrc, int log2S);
int cer_conv3x3_s16(const cer_conv_inputs* in, const int* log2sx, const void* packed_w, const void* packed_collapsed,
                    const void* edge_w, int log2S, const float* bias, const float* init, float* out, float* out2, const float* aux, const float* aux2,
                    int h, int w, int Cout, int epi, int log2s_out, int log2s_aux, int tile_mt, void* stream);
long cer_delta_proj_s16_packed_size(int C);
int cer_delta_proj_s16_pack(const float* w2_oihw, void* packed, int C, int* log2s_out);
long cer_s16_padded_pixels(int h, int w);
int cer_s16_layout_f32(const float* src, float* dst, int h, int w, int C, int layout, int log2s, int inverse, void* stream);
/* Image rows [y0, y0 + nrows) of a frag16 tensor (h x w image, C channels) -> rows [nrows * w, C] (to_tensor = 0) or back
 * (to_tensor = 1), as bit-exact copies of the 16-byte hi | lo pieces (the row-slab halo exchange). */
int cer_s16_rows_f32(float* tensor, float* rows, int h, int w, int C, int y0, int nrows, int to_tensor, void* stream);

/* fp32 [P, C] -> split32 (inverse = 0) or back (inverse = 1); C % 32 == 0. */
int cer_split32_f32(const float* src, float* dst, long P, int C, int inverse, void* stream);

/* Fused delta head (reference: core/update.py:68-71,114; core/raft.py:101).  cer_conv3x3_f16x3 with epi =
 * CER_EPI_DELTA computes hid = relu(conv3x3(net)) (Cout = 256) but never writes it: each 128-channel block
 * projects its hidden tile onto the nine taps of the following 256->1 conv,
 *   T[half][tap][p] = sum_{c in half} w2[0, c, tap] * hid[p, c],           out = T [Cout/128, 9, h*w],
 * with `aux` = the projection weights packed by cer_delta_proj_pack (2-byte halves, size from
 * cer_delta_proj_packed_size).  cer_delta_sum_f32 finishes:
 *   delta[p] = 0.01 * (bias + sum_half sum_tap T[half][tap][p + tap offset]) (zero padding),
 *   disp_out[p] = disp_in[p] + delta[p]   (delta may be NULL; disp_out may alias disp_in). */
long cer_delta_proj_packed_size(int C);
int cer_delta_proj_pack(const float* w2_oihw, void* packed, int C);
int cer_delta_sum_f32(const float* T, int nhalf, float bias, const float* disp_in, float* disp_out, float* delta,
                      int h, int w, void* stream);

/* delta head tail (reference: core/update.py:70-71,114 and core/raft.py:101):
 *   delta[p] = 0.01 * (b + sum_{tap,c} w[tap,c] * hid[p+tap, c]);  disp_out[p] = disp_in[p] + delta[p]
 * hid [h*w, C] (already ReLU'd), w [9, C] (tap-major), C % 256 == 0.  delta may be NULL.
 */
int cer_delta_tail_f32(const float* hid, const float* w, float bias, const float* disp_in,
                       float* disp_out, float* delta, int h, int w_, int C, void* stream);

/* Fused encoder passes (reference: core/extractor.py:49-57,143-150; InstanceNorm2d defaults :28-31).
 * cer_plane_stats_f32: x [planes, plane_size] (one plane per (image, channel), NCHW) -> stats [planes, 2] =
 *   (mean, 1/sqrt(biased_var + eps)), accumulated in fp64.
 * cer_norm_act_f32: out = relu_out( relu_a(norm(x)) + relu_b(norm_r(res)) ) elementwise; x_stats / res / res_stats
 *   may be NULL (no normalisation / no residual); flags: 1 relu_a, 2 relu_b, 4 relu_out; out may alias x.
 * Planes whose size is not a multiple of 4 take a scalar path. */
int cer_plane_stats_f32(const float* x, float* stats, long planes, long plane_size, float eps, void* stream);
int cer_norm_act_f32(const float* x, const float* x_stats, const float* res, const float* res_stats, float* out,
                     long planes, long plane_size, int flags, void* stream);

/* ------------------------------------------------------------------------------------
 * Encoder engine (reference: core/extractor.py:60-155, BasicEncoder "HR"), channels-last,
 * batched, on the split-f16 MFMA path (fp32-equivalent accuracy):
 *
 * cer_enc_stem_f32      7x7 stride-2 pad-3 conv 3->32 of NCHW images [N,3,H,W] (normalize != 0: x*(2/255)-1 first,
 *                       core/raft.py:40-41) -> raw channels-last [N, ho*wo, 32]; weights [147][32] (ci,ky,kx major).
 * cer_enc_conv_f16x3    k x k (taps = 9 or 1), stride 1 or 2, zero pad k/2, Cin,Cout multiples of 32 (stride 2: Cout
 *                       multiple of 64).  The producer's instance norm + ReLU is applied while staging:
 *                         x' = tf_relu ? relu(n(x)) : n(x),  n(x) = tf_stats ? (x - mean[n,c]) * rstd[n,c] : x.
 *                       epi 0 RAW : out [N, ho*wo, Cout] = conv + bias
 *                       epi 1 FMAP: out [N, (ho+2b)*(wo+2b), Cout] = (conv + bias) * out_scale inside a b-texel border
 *                                   (border texels are NOT written: zero the buffer once)
 *                       epi 2 CTX : channels < Cout/2 -> tanh -> out [N, ho*wo, Cout/2]; the rest -> relu -> out2
 * stats_partial (RAW / stem, may be NULL): per-block (sum, sum of squares) per output channel,
 *                       [N][tiles][Cout][2] with tiles = cer_enc_conv_tiles(...) / cer_enc_stem_tiles(...);
 * cer_enc_stats_reduce_f32  partials -> stats [N*C][2] = (mean, 1/sqrt(biased var + eps)) in fp64 (deterministic).
 * cer_enc_merge_f32     out = relu?( fa(a) + fb(b) ) channels-last, f = optional norm with stats, flags 1 relu a,
 *                       2 relu b, 4 relu sum  (core/extractor.py:49-57).
 * Weights: cer_enc_conv_pack (OIHW -> fragment order; size in 2-byte halves from cer_enc_conv_packed_size). */
int cer_enc_stem_tiles(int ho, int wo);
/* The stem on the matrix cores (csrc/enc_stem.hip): same operands and outputs as cer_enc_stem_f32, single-accumulator split-f16
 * MFMA arithmetic (fp32-class).  Weights are packed once on the host from OIHW [32][3][7][7] (cer_enc_stem_s16_pack returns the
 * power-of-two weight scale it applied in *log2s_w; packed buffer: cer_enc_stem_s16_packed_size() halves = two fragment layouts,
 * the second for the round-4 producer / consumer kernel that serves W % 4 == 0 images with 16-byte loads); stats_partial is
 * [N][cer_enc_stem_s16_tiles(ho, wo)][32][2] (8 x 32-pixel output tiles). */
long cer_enc_stem_s16_packed_size(void);
int cer_enc_stem_s16_tiles(int ho, int wo);
int cer_enc_stem_s16_pack(const float* w_oihw, void* packed, int* log2s_w);
int cer_enc_stem_s16(const float* images, const void* packed_w, const float* bias, float* out, float* stats_partial, int N, int H, int W,
                     int normalize, int log2s_w, void* stream);
int cer_enc_stem_f32(const float* images, const float* wgt_k_co, const float* bias, float* out, float* stats_partial,
                     int N, int H, int W, int normalize, void* stream);
long cer_enc_conv_packed_size(int Cout, int Cin, int taps);
int cer_enc_conv_pack(const float* w_oihw, void* packed, int Cout, int Cin, int taps);
int cer_enc_conv_tiles(int ho, int wo, int stride, int taps, int Cout);
int cer_enc_conv_f16x3(const float* src, const float* tf_stats, int tf_relu, const void* packed_w, const float* bias,
                       float* out, float* out2, float* stats_partial, int N, int h, int w, int Cin, int Cout,
                       int taps, int stride, int epi, int out_border, float out_scale, void* stream);
int cer_enc_stats_reduce_f32(const float* partial, float* stats, int N, int nblk, int C, long pixels, float eps, void* stream);
int cer_enc_merge_f32(const float* a, const float* a_stats, const float* b, const float* b_stats, float* out,
                      int N, long pixels, int C, int flags, void* stream);
/* Round 4: the same convolutions with producer / consumer wave roles (csrc/enc_pc.hip; reference: core/extractor.py:49-57,
 * 143-155).  Operand preparation (the producer layer's instance norm + ReLU, split to hi|lo f16) runs in four producer waves
 * UNDER the matrix work of four consumer waves of the same 512-thread persistent block, and the block's input may be the residual
 * merge of TWO tensors formed on the fly,
 *     x = relu4( relu1(n_A(A)) + relu2(n_B(B)) ),   n(t) = stats ? (t - mean[n,c]) * rstd[n,c] : t,   flags = 1 | 2 | 4 as in
 * cer_enc_merge_f32 (srcB NULL: x = relu1(n_A(A))), so that no merged activation has to be produced by a pass of its own;
 * merged_out (optional, stride 1, needs srcB) receives x once per pixel [N, h*w, Cin] for a later residual branch.
 * Same packed weights (cer_enc_conv_pack), same arithmetic (three f16 MFMA terms, two fp32 accumulators) and the same epilogues
 * (epi 0 RAW / 1 FMAP / 2 CTX) as cer_enc_conv_f16x3; stats_partial (RAW) is [N][cer_enc_pc_tiles(...)][Cout][2].
 * epi 3 FSPLIT (64 -> 64 1x1 only): as FMAP, but the scaled map is written straight into the split-f16 operand planes of
 * cer_cost_lines_f32 - `out` = [N][8 planes][(ho+2b)*(wo+2b)][16] halves, exactly what cer_feat_split_f16 makes of the FMAP output
 * (border texels are not written: zero the buffer once); `out2`, if not NULL, is the device overflow flag (int*; bit 1 on saturation).
 * Shapes: cer_enc_pc_supported(Cin, Cout, taps, stride, epi) != 0 - the "HR" encoder's: 32->32 3x3; 32->64 3x3 / 1x1 stride 2;
 * 64->64 3x3; 64->64 1x1 FMAP; 64->128 1x1 CTX.  Everything else returns CER_ESHAPE (use cer_enc_conv_f16x3).
 * Round 6 (ABI 1070), the FP6-correction form: with `flags & 8` and weights packed by cer_enc_conv_pack_f6 (same size and plane order as
 * cer_enc_conv_pack; the lo planes hold e2m3 K blocks instead) the two correction terms of a 32-channel tap run as ONE
 * v_mfma_scale_f32_32x32x64_f8f6f4 with both operands in e2m3 and one power-of-two scale per (pixel | output channel, 16-channel block):
 * half the matrix-pipe cycles of the three-term f16 form, ~15 instead of ~22 product bits in the correction terms (features 4e-5 relative L1
 * from the f16 form, 7e-6 on the final disparity: RAFT's "auto" calibration decides per set of weights).  Same shapes, epilogues and outputs. */
int cer_enc_pc_supported(int Cin, int Cout, int taps, int stride, int epi);
int cer_enc_pc_tiles(int ho, int wo, int Cout, int taps, int stride);
int cer_enc_conv_pack_f6(const float* w_oihw, void* packed, int Cout, int Cin, int taps);
int cer_enc_pc_conv(const float* srcA, const float* statsA, const float* srcB, const float* statsB, int flags, float* merged_out,
                    const void* packed_w, const float* bias, float* out, float* out2, float* stats_partial, int N, int h, int w,
                    int Cin, int Cout, int taps, int stride, int epi, int out_border, float out_scale, void* stream);

/* NCHW [C,h,w] -> NHWC [h*w,C] with a scale (feature maps: scale = 1/8, core/corr.py:30-31)
 * and NHWC -> NCHW; C % 4 == 0. */
int cer_nchw_to_nhwc_f32(const float* src, float* dst, int C, long P, float scale, void* stream);
int cer_nhwc_to_nchw_f32(const float* src, float* dst, int C, long P, float scale, void* stream);
/* Batched NCHW [N,C,h,w] -> channels-last [N,(h+2b)*(w+2b),C] * scale with a b-texel zero border (the source-map
 * layout of cer_cost_build_f32 with b = 2; reference: core/corr.py:29-35 permute, /8.0, contiguous). */
int cer_nchw_to_nhwc_border_f32(const float* src, float* dst, int N, int C, int h, int w, int border, float scale, void* stream);

/* ---- after the path: geometric-consistency filtering of the depth maps ------------------------
 * One reference depth map against S <= 10 source views, fused (reference: fusion.py:39-83 reproject_with_depth,
 * :86-106 check_geometric_consistency, :226-236 vote and averaged depth; utils/bilinear_sampler.py:32-41).
 *   depth_ref [h*w], depth_src [S, h*w] (device);  cams [S][CER_GEO_CAM_FLOATS] (device), per source view, row-major fp32:
 *     K_ref^-1 [9] | (E_src E_ref^-1) rows 0..2 [12] | K_src [9] | K_src^-1 [9] | (E_ref E_src^-1) rows 0..2 [12] | K_ref [9]
 *   thre1, thre2: the reference's Python floats (mask i, i = 2..10: dist < i/thre1 and |d_reproj - d|/d < i/thre2).
 * Outputs (any may be NULL): geo_mask [h*w] 0/1 (mask10 of all views, or mask_i of >= i views for some i < 1+S),
 * depth_est [h*w] = (sum of mask10-consistent reprojected depths + depth_ref) / (count + 1);
 * mask_count [CER_GEO_COUNTERS] (unsigned): the area of geo_mask is ADDED, spread over the counters (their sum is the area -
 * one hot address would serialise the blocks' atomics).
 * Per-view tensors of the reference's API, written only when the pointer is given: masks9 [9, S, h*w] (0/1),
 * depth_reprojected [S, h*w] (zero outside mask10), x_src, y_src, rel_diff [S, h*w]. */
#define CER_GEO_CAM_FLOATS 60
#define CER_GEO_COUNTERS 64
int cer_geo_consistency_f32(const float* depth_ref, const float* depth_src, const float* cams, int S, int h, int w,
                            double thre1, double thre2, unsigned char* geo_mask, float* depth_est, unsigned int* mask_count,
                            unsigned char* masks9, float* depth_reprojected, float* x_src, float* y_src, float* rel_diff,
                            void* stream);

/* Multi-resolution merge of two depth maps (reference: multires.py:16-40): out [h2, w2] = where(|r - im2| < th * r, im2, r) with
 * r = im1 [h1, w1] resized to [h2, w2] like cv2.resize(INTER_LINEAR) on float32; cer_resize_linear_f32 is that resize alone (the
 * reference's optional down_sample step).  Device pointers, fp32, row-major. */
int cer_multires_merge_f32(const float* im1, int h1, int w1, const float* im2, int h2, int w2, double th, float* out, void* stream);
int cer_resize_linear_f32(const float* src, int h, int w, float* dst, int ho, int wo, void* stream);

/* Multi-GPU row-slab exchange (cer-mvs_amd/slab.py): up to CER_COPY_MAX_SEG contiguous fp32 ranges copied by ONE launch -
 * the pack of a rank's (net, disp) border strips into its send buffer, and the refresh of its halo rows from the gathered
 * strips.  n[i] floats from src[i] to dst[i]; n[i] == 0 skips a segment.  Device pointers; ranges must not overlap. */
#define CER_COPY_MAX_SEG 4
typedef struct cer_copy_segments {
    const float* src[CER_COPY_MAX_SEG];
    float* dst[CER_COPY_MAX_SEG];
    long n[CER_COPY_MAX_SEG];
} cer_copy_segments;
int cer_copy_segments_f32(const cer_copy_segments* seg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CER_MVS_H */
