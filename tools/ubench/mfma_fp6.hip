// Micro-benchmark for VERDICT r5 item 1(i): the two correction terms of the split-f16 product on v_mfma_scale_f32_32x32x64_f8f6f4 with FP6
// (e2m3) operands - 8 passes instead of the 16 of the fp8 form when BOTH operands are FP6 / FP4.  The guide names the builtin, not the layout:
//   (1) operand layout of the FP6 form (which K a lane's i-th 6-bit field is) and what the per-lane E8M0 scale byte covers, against a host product;
//   (2) v_cvt_scalef32_pk32_fp6_f16: element order of the 32 packed results, whether it divides or multiplies by its scale operand, rounding;
//   (3) issue rate of the FP6 form against the fp8 form and v_mfma_f32_32x32x16_f16 (one wave per SIMD, four independent accumulators).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fp6.hip -o tools/ubench/mfma_fp6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef int intx6 __attribute__((ext_vector_type(6)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %d\n", (int)e_, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// e2m3: code = s | e(2) | m(3); e = 0: m / 8 (subnormal); else 2^(e - 1) * (1 + m / 8); max 7.5
static float e2m3_value(int code) {
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
    return s ? -v : v;
}
static int e2m3_encode(float x) {                          // round to nearest even, saturating
    const int s = x < 0 ? 32 : 0;
    const float a = fabsf(x);
    int best = 0;
    float bd = 1e30f;
    for (int c = 0; c < 32; ++c) {
        const float d = fabsf(e2m3_value(c) - a);
        if (d < bd || (d == bd && (c & 1) == 0 && (best & 1) == 1)) { bd = d; best = c; }
    }
    return s | best;
}
static void pack6(const int* codes, unsigned* out6) {       // 32 six-bit fields, field i at bit 6 i
    memset(out6, 0, 24);
    for (int i = 0; i < 32; ++i) {
        const int bit = 6 * i;
        out6[bit >> 5] |= (unsigned)codes[i] << (bit & 31);
        if ((bit & 31) > 26) out6[(bit >> 5) + 1] |= (unsigned)codes[i] >> (32 - (bit & 31));
    }
}

__global__ void one_mfma6(const intx8* a, const intx8* b, const int* sa, const int* sb, floatx16* c) {
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 2, 2, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    c[threadIdx.x] = acc;
}
// K index of field i of lane-half g: hypothesis 0: 32 g + i; 1: 16 g + (i & 15) + 32 (i >> 4) (the fp8 form's interleave)
static int kidx(int hyp, int g, int i) { return hyp == 0 ? 32 * g + i : 16 * g + (i & 15) + 32 * (i >> 4); }
static int check_layout() {
    static float A[32][64], B[64][32];
    srand(7);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = e2m3_value(rand() & 63);
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = e2m3_value(rand() & 63);
    void *da, *db, *dsa, *dsb, *dc;
    CHECK(hipMalloc(&da, 2048)); CHECK(hipMalloc(&db, 2048)); CHECK(hipMalloc(&dsa, 256)); CHECK(hipMalloc(&dsb, 256)); CHECK(hipMalloc(&dc, 4096));
    int ok = 0;
    for (int hyp = 0; hyp < 2; ++hyp)
        for (int sc = 0; sc < 2; ++sc) {
            unsigned ha[64][8], hb[64][8];
            int hsa[64], hsb[64];
            memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
            for (int l = 0; l < 64; ++l) {
                int ca[32], cb[32];
                for (int i = 0; i < 32; ++i) {
                    ca[i] = e2m3_encode(A[l & 31][kidx(hyp, l >> 5, i)]);
                    cb[i] = e2m3_encode(B[kidx(hyp, l >> 5, i)][l & 31]);
                }
                pack6(ca, ha[l]); pack6(cb, hb[l]);
                hsa[l] = 127 + (sc ? l % 3 : 0);
                hsb[l] = 127 - 5 + (sc ? l % 2 : 0);
            }
            CHECK(hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(one_mfma6, dim3(1), dim3(64), 0, 0, (const intx8*)da, (const intx8*)db, (const int*)dsa, (const int*)dsb, (floatx16*)dc);
            float hc[64][16];
            CHECK(hipMemcpy(hc, dc, 4096, hipMemcpyDeviceToHost));
            // scale hypotheses: 0: lane l's scale covers the 32 values lane l holds; 1: covers K block l >> 5 = K / 32
            for (int sh = 0; sh < 2; ++sh) {
                double worst = 0, big = 0;
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 16; ++r) {
                        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                        double ref = 0;
                        for (int g = 0; g < 2; ++g)
                            for (int i = 0; i < 32; ++i) {
                                const int k = kidx(hyp, g, i);
                                const int lg = sh == 0 ? g : k / 32;
                                const double fa = sc ? ldexp(1.0, (row + 32 * lg) % 3) : 1.0, fb = ldexp(1.0, -5 + (sc ? (col + 32 * lg) % 2 : 0));
                                ref += (double)A[row][k] * B[k][col] * fa * fb;
                            }
                        worst = fmax(worst, fabs(ref - hc[l][r])); big = fmax(big, fabs(ref));
                    }
                printf("FP6 layout hypothesis K%d, %s scales (scale hyp %d): max |diff| = %.3g of %.3g %s\n", hyp, sc ? "per-lane" : "uniform", sh, worst, big,
                       worst < 1e-5 * big ? "OK" : "");
                if (worst < 1e-5 * big) ++ok;
            }
        }
    return ok > 0;
}

// ---------------------------------------------------------------- (2) the conversion
__global__ void cvt32(const half32* in, const float* scale, intx6* out) {
    out[threadIdx.x] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(in[threadIdx.x], scale[threadIdx.x]);
}
static void check_cvt() {
    static _Float16 h[64][32];
    float sc[64];
    srand(11);
    for (int l = 0; l < 64; ++l) {
        sc[l] = ldexpf(1.0f, (l % 5) - 2);
        for (int i = 0; i < 32; ++i) h[l][i] = (_Float16)((float)((rand() % 2001) - 1000) / 1000.0f * 9.0f * sc[l]);
        h[l][0] = (_Float16)(0.1875f * sc[l]);             // a tie between 0.125 and 0.25 -> even mantissa (0.25: m = 2)
        h[l][1] = (_Float16)(100.0f * sc[l]);              // saturates to 7.5
        h[l][2] = (_Float16)(0.0f);
    }
    void *di, *ds, *dout;
    CHECK(hipMalloc(&di, sizeof(h))); CHECK(hipMalloc(&ds, 256)); CHECK(hipMalloc(&dout, 64 * 32));      // (sizeof(intx6) == 32: six ints padded to eight)
    CHECK(hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice)); CHECK(hipMemcpy(ds, sc, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt32, dim3(1), dim3(64), 0, 0, (const half32*)di, (const float*)ds, (intx6*)dout);
    unsigned got[64][8];
    CHECK(hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost));
    for (int mode = 0; mode < 2; ++mode) {                 // 0: value / scale, 1: value * scale
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            int codes[32];
            unsigned want[6];
            for (int i = 0; i < 32; ++i) codes[i] = e2m3_encode(mode == 0 ? (float)h[l][i] / sc[l] : (float)h[l][i] * sc[l]);
            pack6(codes, want);
            for (int q = 0; q < 6; ++q) if (want[q] != got[l][q]) { ++bad; break; }
        }
        printf("v_cvt_scalef32_pk32_fp6_f16: fields in element order, %s its scale, RNE, saturating: %d of 64 lanes differ %s\n", mode == 0 ? "DIVIDES by" : "MULTIPLIES by", bad,
               bad == 0 ? "OK" : "");
    }
    // first lanes decoded, for the record
    for (int l = 0; l < 2; ++l) {
        printf("  lane %d scale %g:", l, sc[l]);
        for (int i = 0; i < 6; ++i) {
            const int bit = 6 * i;
            unsigned long long w = got[l][bit >> 5] | ((unsigned long long)got[l][(bit >> 5) + 1] << 32);
            printf("  %g -> %g", (float)h[l][i], e2m3_value((int)((w >> (bit & 31)) & 63)));
        }
        printf("\n");
    }
}

// ---------------------------------------------------------------- (3) issue rates
template <int FORM>       // 0: f16 32x32x16, 1: fp8 32x32x64, 2: fp6 32x32x64
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    floatx16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    intx8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = 0x01010101 * (threadIdx.x & 3); b[q] = 0x02020202; }
    half8 ha, hb;
    for (int q = 0; q < 8; ++q) { ha[q] = (_Float16)0.5f; hb[q] = (_Float16)0.25f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (FORM == 0) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[m], 0, 0, 0);
            if (FORM == 1) acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[m], 0, 0, 0, 127, 0, 127);
            if (FORM == 2) acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[m], 2, 2, 0, 127, 0, 127);
        }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int FORM>
static void time_rate(const char* name, double flop_per_mfma) {
    float* d;
    CHECK(hipMalloc(&d, 1024 * 256 * 4));
    const int iters = 20000, blocks = 1024;
    hipLaunchKernelGGL(rate<FORM>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate<FORM>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)blocks * 4 * iters * 4;
    printf("%-28s %.3f ms: %.0f TF; per MFMA and SIMD %.1f cycles at 2.4 GHz (1024 SIMDs)\n", name, ms, n * flop_per_mfma / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (n / 1024.0));
}

int main() {
    const int ok = check_layout();
    check_cvt();
    time_rate<0>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16);
    time_rate<1>("f8f6f4 32x32x64, fp8 x fp8", 2.0 * 32 * 32 * 64);
    time_rate<2>("f8f6f4 32x32x64, fp6 x fp6", 2.0 * 32 * 32 * 64);
    return ok ? 0 : 1;
}
