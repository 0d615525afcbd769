/* ORACLE (test infrastructure - NOT the product path).
 *
 * Plain-C, scalar, loop-for-loop restatement of the correlation half of the CER-MVS hot path,
 * written from a reading of the reference (file:line cited per function).  It exists to pin the one
 * piece of the reference that cannot be executed in the build container - the CUDA kernel
 * alt_cuda_corr/correlation_kernel.cu:18-119 - independently of the torch/grid_sample restatement
 * in oracle/cer_oracle.py: tests/test_oracle_c.py checks the two against each other and against the
 * golden captures.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; -ffp-contract=off so a*b+c rounds twice as written).
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

static int within(int h, int w, int H, int W) { return h >= 0 && h < H && w >= 0 && w < W; }

/* corr_forward_kernel, any radius (correlation_kernel.cu:59-116): for every texel (iy, ix) of the
 * (rd+1) x (rd+1) footprint, s = <f1[pixel], f2[texel]> accumulated in 32-channel chunks (:43,:88-90),
 * then scattered with the four bilinear weights nw/ne/sw/se into output channels
 * (iy-1)+rd*(ix-1), (iy-1)+rd*ix, iy+rd*(ix-1), iy+rd*ix (:92-114).  corr must be zero-filled (:273). */
void oracle_alt_corr_forward(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                             int B, int N, int H1, int W1, int H2, int W2, int C, int r) {
    const int rd = 2 * r + 1;
    memset(corr, 0, sizeof(float) * (size_t)B * N * rd * rd * H1 * W1);
    for (int b = 0; b < B; ++b)
        for (int c0 = 0; c0 < C; c0 += 32)
            for (int n = 0; n < N; ++n)
                for (int h1 = 0; h1 < H1; ++h1)
                    for (int w1 = 0; w1 < W1; ++w1) {
                        const float* cp = coords + ((((size_t)b * N + n) * H1 + h1) * W1 + w1) * 2;
                        const float x = cp[0], y = cp[1];
                        const float dx = x - floorf(x), dy = y - floorf(y);
                        const float* f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C;
                        float* out = corr + ((size_t)b * N + n) * rd * rd * H1 * W1 + (size_t)h1 * W1 + w1;
                        const size_t HW = (size_t)H1 * W1;
                        for (int iy = 0; iy < rd + 1; ++iy)
                            for (int ix = 0; ix < rd + 1; ++ix) {
                                const int h2 = (int)floorf(y) - r + iy, w2 = (int)floorf(x) - r + ix;
                                float s = 0.0f;
                                if (within(h2, w2, H2, W2)) {
                                    const float* f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C;
                                    const int cend = c0 + 32 < C ? c0 + 32 : C;
                                    for (int k = c0; k < cend; ++k) s += f1[k] * f2[k];
                                }
                                const float nw = s * dy * dx, ne = s * dy * (1 - dx), sw = s * (1 - dy) * dx, se = s * (1 - dy) * (1 - dx);
                                if (iy > 0 && ix > 0) out[HW * ((iy - 1) + rd * (ix - 1))] += nw;
                                if (iy > 0 && ix < rd) out[HW * ((iy - 1) + rd * ix)] += ne;
                                if (iy < rd && ix > 0) out[HW * (iy + rd * (ix - 1))] += sw;
                                if (iy < rd && ix < rd) out[HW * (iy + rd * ix)] += se;
                            }
                    }
}

/* CorrBlock.__init__ up to the per-view volume (core/corr.py:56-91; projective_ops.py:5-28; core/corr.py:28-43).
 * fmaps NCHW [V+1,C,h,w] (unscaled; the /8 of corr.py:30-31 is applied here), Pij [V,16], disp_in [h*w].
 * vol [V, P, D], origin [P]. */
void oracle_cost_volume(const float* fmaps, const float* Pij, const float* disp_in, float* vol, float* origin,
                        int V, int C, int h, int w, int D, double incre, int shift) {
    const int P = h * w;
    const float lim = (float)((D / 2) * incre), inc = (float)incre;
    for (int p = 0; p < P; ++p) origin[p] = (shift && disp_in[p] < lim) ? lim : disp_in[p];
    for (int v = 0; v < V; ++v) {
        const float* m = Pij + 16 * v;
        const float* f2 = fmaps + (size_t)(v + 1) * C * P;
        for (int p = 0; p < P; ++p) {
            const float px = (float)(p % w), py = (float)(p / w);
            for (int k = 0; k < D; ++k) {
                const float a = (float)(k - D / 2) * inc;
                const float d = a + origin[p];
                /* einsum('kh,...h->...k'): sequential 4-term dot, then divide by the z row (projective_ops.py:26-28) */
                float X = m[0] * px; X += m[1] * py; X += m[2] * 1.0f; X += m[3] * d;
                float Y = m[4] * px; Y += m[5] * py; Y += m[6] * 1.0f; Y += m[7] * d;
                float Z = m[8] * px; Z += m[9] * py; Z += m[10] * 1.0f; Z += m[11] * d;
                float x = X / Z, y = Y / Z;
                x = x < -1e4f ? -1e4f : (x > 1e4f ? 1e4f : x);          /* core/corr.py:88 */
                y = y < -1e4f ? -1e4f : (y > 1e4f ? 1e4f : y);
                const float fx = floorf(x), fy = floorf(y), dx = x - fx, dy = y - fy;
                float acc = 0.0f;
                for (int iy = 0; iy < 2; ++iy)
                    for (int ix = 0; ix < 2; ++ix) {
                        const int h2 = (int)fy + iy, w2 = (int)fx + ix;
                        if (!within(h2, w2, h, w)) continue;
                        float s = 0.0f;
                        for (int c = 0; c < C; ++c)
                            s += (fmaps[(size_t)c * P + p] / 8.0f) * (f2[(size_t)c * P + (size_t)h2 * w + w2] / 8.0f);
                        acc += s * (iy ? dy : 1 - dy) * (ix ? dx : 1 - dx);
                    }
                vol[((size_t)v * P + p) * D + k] = acc;
            }
        }
    }
}

/* core/corr.py:94-97: one avg_pool2d([1,2]) step, rows x n -> rows x n/2 */
void oracle_pool(const float* src, float* dst, long rows, int n) {
    const int m = n / 2;
    for (long r = 0; r < rows; ++r)
        for (int k = 0; k < m; ++k) dst[r * m + k] = (src[r * n + 2 * k] + src[r * n + 2 * k + 1]) / 2.0f;
}

/* CorrBlock.__call__ on one level (core/corr.py:107,123-137; bilinear_sampler.py:6-25, direct pixel-space lerp):
 * out[t, row] = lerp(level[row, :], c[row]/2^lv + (t - r)), zero outside. level [rows, n]; c [rows]; out [2r+1, rows] */
void oracle_lookup_level(const float* level, const float* c, float* out, long rows, int n, int lv, int r) {
    for (long row = 0; row < rows; ++row) {
        const float x0 = c[row] / (float)(1 << lv);
        for (int t = 0; t < 2 * r + 1; ++t) {
            const float x = x0 + (float)(t - r);
            const float fx = floorf(x), wgt = x - fx;
            float a = 0.0f, b = 0.0f;
            if (fx >= 0.0f && fx <= (float)(n - 1)) a = level[row * n + (long)fx];
            if (fx + 1.0f >= 0.0f && fx + 1.0f <= (float)(n - 1)) b = level[row * n + (long)fx + 1];
            out[(long)t * rows + row] = a * (1.0f - wgt) + b * wgt;
        }
    }
}
