// Micro-test: accuracy of the split-f16 product  x*w ~= xh*wh + xh*wl + xl*wh  on v_mfma_f32_32x32x16_f16 when
//   U) hi*hi and the (2^11-scaled) cross terms live in SEPARATE fp32 accumulators (round-1 kernels), versus
//   M) everything accumulates into ONE fp32 accumulator with unscaled lo parts (power-of-two operand scales keep the lo parts
//      inside f16's range; frees 16 VGPRs per 32x32 tile),
// against an fp64 reference, on the K = 1248 contraction of the z|r gate conv.  Also reports the plain f16 product (P) and an
// fp32 fmaf chain (F) for scale.  Decides whether the merged form is fp32-class (no truncation bias inside the MFMA adder).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_merge_acc.hip -o tools/ubench/mfma_merge_acc && tools/ubench/mfma_merge_acc
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// operands in fragment order: [k16-step][lane][8]
template <int MODE>   // 0 = U, 1 = M (hh, hl, lh), 2 = M cross terms first, 3 = plain
__global__ __launch_bounds__(64) void k(const _Float16* ah, const _Float16* al, const _Float16* bh, const _Float16* bl, float* out, int nk,
                                        float inv_scale) {
    const int lane = threadIdx.x;
    floatx16 m, l;
    for (int r = 0; r < 16; ++r) { m[r] = 0.f; l[r] = 0.f; }
    for (int s = 0; s < nk; ++s) {
        const half8 xh = *(const half8*)(ah + ((long)s * 64 + lane) * 8), xl = *(const half8*)(al + ((long)s * 64 + lane) * 8);
        const half8 wh = *(const half8*)(bh + ((long)s * 64 + lane) * 8), wl = *(const half8*)(bl + ((long)s * 64 + lane) * 8);
        if (MODE == 0) {
            m = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, m, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, l, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, l, 0, 0, 0);
        } else if (MODE == 1) {
            m = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, m, 0, 0, 0);
        } else if (MODE == 2) {
            m = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, m, 0, 0, 0);
        } else {
            m = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, m, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        out[row * 32 + col] = (MODE == 0) ? fmaf(l[r], 1.0f / 2048.0f, m[r]) : m[r] * inv_scale;
    }
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

int main() {
    const int K = 1248, NK = K / 16;
    srand(7);
    for (int trial = 0; trial < 4; ++trial) {
        // trial 0: activations U[-1,1], weights N(0, 0.05);  1: relu-like activations |N(0,1)| (positive: biased sums);
        // 2: tiny activations (1e-3 scale, disparity-feature like);  3: all-positive operands (worst case for truncation bias)
        std::vector<float> X(32 * K), W(K * 32);
        for (auto& v : X) v = trial == 0 ? (float)(2 * urand() - 1) : trial == 1 ? (float)fabs(nrand()) : trial == 2 ? (float)(1e-3 * nrand()) : (float)urand();
        for (auto& v : W) v = trial == 3 ? (float)(0.05 * urand()) : (float)(0.05 * nrand());
        const float sx = trial == 0 ? 16384.f : trial == 2 ? 64.f : 64.f, sw = 1024.f;      // powers of two (exact)
        auto pack = [&](int mode, std::vector<_Float16>& ah, std::vector<_Float16>& al, std::vector<_Float16>& bh, std::vector<_Float16>& bl) {
            ah.resize(NK * 512); al.resize(NK * 512); bh.resize(NK * 512); bl.resize(NK * 512);
            for (int s = 0; s < NK; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int kk = s * 16 + (lane >> 5) * 8 + e, i = lane & 31;
                        float x = X[i * K + kk], w = W[kk * 32 + i];
                        _Float16 h, l;
                        if (mode == 0) { h = (_Float16)x; l = (_Float16)((x - (float)h) * 2048.f); }
                        else { x *= sx; h = (_Float16)x; l = (_Float16)(x - (float)h); }
                        ah[(s * 64 + lane) * 8 + e] = h; al[(s * 64 + lane) * 8 + e] = l;
                        if (mode == 0) { h = (_Float16)w; l = (_Float16)((w - (float)h) * 2048.f); }
                        else { w *= sw; h = (_Float16)w; l = (_Float16)(w - (float)h); }
                        bh[(s * 64 + lane) * 8 + e] = h; bl[(s * 64 + lane) * 8 + e] = l;
                    }
        };
        std::vector<double> ref(1024), mag(1024);
        std::vector<float> f32(1024);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double a = 0, m = 0; float f = 0.f;
                for (int kk = 0; kk < K; ++kk) { a += (double)X[i * K + kk] * W[kk * 32 + j]; m += fabs((double)X[i * K + kk] * W[kk * 32 + j]); f = fmaf(X[i * K + kk], W[kk * 32 + j], f); }
                ref[i * 32 + j] = a; mag[i * 32 + j] = m; f32[i * 32 + j] = f;
            }
        _Float16 *dah, *dal, *dbh, *dbl; float* dout;
        hipMalloc(&dah, NK * 1024); hipMalloc(&dal, NK * 1024); hipMalloc(&dbh, NK * 1024); hipMalloc(&dbl, NK * 1024); hipMalloc(&dout, 4096);
        auto report = [&](const char* name, const float* o) {
            double rms = 0, mx = 0, bias = 0, rel_l1n = 0, rel_l1d = 0;
            for (int q = 0; q < 1024; ++q) {
                const double e = ((double)o[q] - ref[q]) / mag[q];
                rms += e * e; mx = fmax(mx, fabs(e)); bias += e;
                rel_l1n += fabs((double)o[q] - ref[q]); rel_l1d += fabs(ref[q]);
            }
            printf("  %-44s rms %.3e  max %.3e  mean(signed) %+.3e   [err / sum|x||w|];  rel-L1 %.3e\n", name, sqrt(rms / 1024), mx, bias / 1024, rel_l1n / rel_l1d);
        };
        printf("trial %d\n", trial);
        report("F  fp32 fmaf chain (host)", f32.data());
        std::vector<float> o(1024);
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<_Float16> ah, al, bh, bl;
            pack(mode == 0 ? 0 : 1, ah, al, bh, bl);
            if (mode == 3) pack(0, ah, al, bh, bl);
            hipMemcpy(dah, ah.data(), NK * 1024, hipMemcpyHostToDevice); hipMemcpy(dal, al.data(), NK * 1024, hipMemcpyHostToDevice);
            hipMemcpy(dbh, bh.data(), NK * 1024, hipMemcpyHostToDevice); hipMemcpy(dbl, bl.data(), NK * 1024, hipMemcpyHostToDevice);
            const float inv = 1.0f / (sx * sw);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dah, dal, dbh, dbl, dout, NK, inv);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dah, dal, dbh, dbl, dout, NK, inv);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dah, dal, dbh, dbl, dout, NK, inv);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, dah, dal, dbh, dbl, dout, NK, 1.0f);
            hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
            report(mode == 0 ? "U  separate accumulators, lo scaled 2^11" : mode == 1 ? "M  one accumulator, unscaled lo (hh,hl,lh)" : mode == 2 ? "M2 one accumulator, cross terms first" : "P  plain f16 operands", o.data());
        }
        hipFree(dah); hipFree(dal); hipFree(dbh); hipFree(dbl); hipFree(dout);
    }
    return 0;
}
