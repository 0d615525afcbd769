// Micro-benchmark for DESIGN.md §7 "what comes next" item 1: the two correction terms of the split-f16 product (xh*wl + xl*wh)
// on the block-scaled fp8 matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4, twice the f16 rate) instead of two
// v_mfma_f32_32x32x16_f16, INTO THE SAME fp32 accumulator (uniform E8M0 scales carry the 2^-11 of the lo halves).
//   (1) operand layout / scale semantics of the instruction, checked against a host product;
//   (2) throughput of the update-block conv's K-loop shape (4-wave blocks, 2 per CU, 4 accumulators per wave, weights from
//       global memory in fragment order, activations from LDS with ds_read_b128) in the shipped form (3 f16 MFMAs per
//       16-channel tap step) and in the fp8-correction form (per 32-channel tap: 2 f16 MFMAs + 1 fp8 K = 64 MFMA).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fp8_corr.hip -o tools/ubench/mfma_fp8_corr
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %d\n", (int)e_, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- (1) layout
__global__ void one_mfma(const intx8* a, const intx8* b, const int* sa, const int* sb, floatx16* c) {
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    c[threadIdx.x] = acc;
}
static unsigned char e4m3(float v) {                      // exact for the small integers / powers of two used here
    if (v == 0.f) return 0;
    const unsigned s = v < 0 ? 0x80 : 0;
    v = fabsf(v);
    int e;
    const float m = frexpf(v, &e);                        // v = m * 2^e, m in [0.5, 1)
    int E = e - 1 + 7;
    int M = (int)lrintf((m * 2.f - 1.f) * 8.f);
    if (E <= 0) { M = (int)lrintf(v * 512.f); E = 0; }    // subnormal: M * 2^-9
    return (unsigned char)(s | (E << 3) | (M & 7));
}
// K index of byte q of lane-half g under hypothesis hyp: 0: 32 g + q;  1: 16-byte halves interleaved: 16 g + (q & 15) + 32 (q >> 4)
static int kidx(int hyp, int g, int q) { return hyp == 0 ? 32 * g + q : 16 * g + (q & 15) + 32 * (q >> 4); }
static int check_layout() {
    float A[32][64], B[64][32];
    srand(5);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = (float)(rand() % 9 - 4);
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = (float)(rand() % 9 - 4) * 0.5f;
    void *da, *db, *dsa, *dsb, *dc;
    CHECK(hipMalloc(&da, 2048)); CHECK(hipMalloc(&db, 2048)); CHECK(hipMalloc(&dsa, 256)); CHECK(hipMalloc(&dsb, 256)); CHECK(hipMalloc(&dc, 4096));
    int ok = 0;
    for (int hyp = 0; hyp < 2; ++hyp)
        for (int sc = 0; sc < 2; ++sc) {              // sc 0: uniform scales (1, 2^-11); 1: per-lane scales
            unsigned char ha[64][32], hb[64][32];
            int hsa[64], hsb[64];
            for (int l = 0; l < 64; ++l) {
                for (int q = 0; q < 32; ++q) {
                    ha[l][q] = e4m3(A[l & 31][kidx(hyp, l >> 5, q)]);
                    hb[l][q] = e4m3(B[kidx(hyp, l >> 5, q)][l & 31]);
                }
                hsa[l] = 127 + (sc ? l % 3 : 0);
                hsb[l] = 127 - 11 + (sc ? l % 2 : 0);
            }
            CHECK(hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, (const intx8*)da, (const intx8*)db, (const int*)dsa, (const int*)dsb, (floatx16*)dc);
            float hc[64][16];
            CHECK(hipMemcpy(hc, dc, 4096, hipMemcpyDeviceToHost));
            // scale hypotheses: 0: lane l's scale covers (row l & 31, the 32 values lane l holds); 1: covers K block l >> 5 = K / 32
            for (int sh = 0; sh < 2; ++sh) {
                double worst = 0, big = 0;
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 16; ++r) {
                        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                        double ref = 0;
                        for (int g = 0; g < 2; ++g)
                            for (int q = 0; q < 32; ++q) {
                                const int k = kidx(hyp, g, q);
                                const int lg = sh == 0 ? g : k / 32;          // which lane half's scale applies to element k
                                const double fa = sc ? ldexp(1.0, (row + 32 * lg) % 3) : 1.0, fb = ldexp(1.0, -11 + (sc ? (col + 32 * lg) % 2 : 0));
                                ref += (double)A[row][k] * B[k][col] * fa * fb;
                            }
                        worst = fmax(worst, fabs(ref - hc[l][r])); big = fmax(big, fabs(ref));
                    }
                printf("layout hypothesis K%d, %s scales (scale hyp %d): max |diff| = %.3g of %.3g %s\n", hyp, sc ? "per-lane" : "uniform", sh, worst, big, worst < 1e-6 ? "OK" : "");
                if (worst < 1e-6) ++ok;
            }
        }
    return ok > 0;
}

// ---------------------------------------------------------------- (2) K-loop shapes
#define ROWB (20 * 64)
// shipped form: per step 12 f16 MFMAs, 8 ds_read_b128 (one (hi, lo) pair per accumulator), 2 x 16 B of weights from global
template <int MT>
__global__ __launch_bounds__(256, 2) void loop_f16x3(const char* __restrict__ wts, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kg = lane >> 5;
    for (int i = tid; i < 2 * 10 * ROWB / 4; i += 256) reinterpret_cast<int*>(smem)[i] = 0x3c003c00 + (i & 3);
    __syncthreads();
    int xh[3], xl[3];
    for (int dx = 0; dx < 3; ++dx) {
        const int col = (li & 15) + dx, base = (li >> 4) * ROWB + col * 64, sl = kg ^ ((col >> 2) & 3);
        xh[dx] = base + sl * 16; xl[dx] = base + (sl ^ 2) * 16;
    }
    const char* wl = wts + ((long)(blockIdx.x & 1) * 4 + wave) * 2048 + lane * 16;
    floatx16 acc[MT];
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    half8 fh[MT], fl[MT], wh[3], wlo[3];
    for (int m = 0; m < MT; ++m) { fh[m] = *reinterpret_cast<const half8*>(smem + xh[0] + 2 * m * ROWB); fl[m] = *reinterpret_cast<const half8*>(smem + xl[0] + 2 * m * ROWB); }
    wh[0] = *reinterpret_cast<const half8*>(wl); wlo[0] = *reinterpret_cast<const half8*>(wl + 1024);
    wh[1] = *reinterpret_cast<const half8*>(wl + 16384); wlo[1] = *reinterpret_cast<const half8*>(wl + 16384 + 1024);
    for (int s = 0; s < steps; s += 9) {
        const int buf = ((s / 9) & 1) * 10 * ROWB;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int nt = (t + 1) % 9, dy = nt / 3, dx = nt % 3;
            const char* p = wl + (long)((s + t + 2) % 78) * 16384;
            wh[(t + 2) % 3] = *reinterpret_cast<const half8*>(p); wlo[(t + 2) % 3] = *reinterpret_cast<const half8*>(p + 1024);
            const half8 a = wh[t % 3], b = wlo[t % 3];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, fh[m], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, fh[m], acc[m], 0, 0, 0);
                fh[m] = *reinterpret_cast<const half8*>(smem + buf + xh[dx] + (2 * m + dy) * ROWB);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, fl[m], acc[m], 0, 0, 0);
                fl[m] = *reinterpret_cast<const half8*>(smem + buf + xl[dx] + (2 * m + dy) * ROWB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float sum = 0;
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) sum += acc[m][r];
    out[blockIdx.x * 256 + tid] = sum;
}
// fp8-correction form: per 32-channel tap ("pair" = 2 steps of the form above): 2 * MT f16 MFMAs (hi*hi of both 16-channel
// halves), MT fp8 K = 64 MFMAs; per accumulator 2 ds_read_b128 (f16 hi halves) + 2 ds_read_b128 (32 fp8 bytes: hi8 | lo8 of one
// half per K block); weights 2 x 16 B f16 hi + 32 B fp8 per lane.  Pixel stride 128 B (32 channels: 64 B f16 hi + 64 B fp8).
template <int MT>
__global__ __launch_bounds__(256, 2) void loop_fp8c(const char* __restrict__ wts, float* out, int pairs, int sa, int sb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RB = 20 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kg = lane >> 5;
    for (int i = tid; i < 2 * 10 * RB / 4; i += 256) reinterpret_cast<int*>(smem)[i] = (i & 16) ? 0x38383838 : 0x3c003c00 + (i & 3);
    __syncthreads();
    // 8 x 16-byte slots per pixel: f16 hi of half-chunk 0 (kg 0, 1), of half-chunk 1 (kg 0, 1), fp8 of K block 0 (2 slots), K block 1;
    // swizzled with (col >> 1) & 7 so that the 16 pixels x 2 rows a quarter-wave phase touches spread over the banks
    int xa[3], xb[3], xc[3], xd[3];
    for (int dx = 0; dx < 3; ++dx) {
        const int col = (li & 15) + dx, base = (li >> 4) * RB + col * 128, key = (col >> 1) & 7;
        xa[dx] = base + ((kg) ^ key) * 16; xb[dx] = base + ((2 + kg) ^ key) * 16;
        xc[dx] = base + ((4 + 2 * kg) ^ key) * 16; xd[dx] = base + ((5 + 2 * kg) ^ key) * 16;
    }
    const char* wl = wts + ((long)(blockIdx.x & 1) * 4 + wave) * 4096 + lane * 16;
    floatx16 acc[MT];
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    half8 f0[MT], f1[MT];
    union Q { intx8 v; uint4 q[2]; };
    Q f8[MT];
    struct W { half8 h0, h1; Q q; } w[3];
    auto load_w = [&](W& x, int pair) {
        const char* p = wl + (long)(pair % 39) * 32768;
        x.h0 = *reinterpret_cast<const half8*>(p); x.h1 = *reinterpret_cast<const half8*>(p + 1024);
        x.q.q[0] = *reinterpret_cast<const uint4*>(p + 2048); x.q.q[1] = *reinterpret_cast<const uint4*>(p + 3072);
    };
    for (int m = 0; m < MT; ++m) {
        f0[m] = *reinterpret_cast<const half8*>(smem + xa[0] + 2 * m * RB); f1[m] = *reinterpret_cast<const half8*>(smem + xb[0] + 2 * m * RB);
        f8[m].q[0] = *reinterpret_cast<const uint4*>(smem + xc[0] + 2 * m * RB); f8[m].q[1] = *reinterpret_cast<const uint4*>(smem + xd[0] + 2 * m * RB);
    }
    load_w(w[0], 0); load_w(w[1], 1);
    for (int s = 0; s < pairs; s += 9) {
        const int buf = ((s / 9) & 1) * 10 * RB;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int nt = (t + 1) % 9, dy = nt / 3, dx = nt % 3;
            load_w(w[(t + 2) % 3], s + t + 2);
            const W& c = w[t % 3];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c.h0, f0[m], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c.h1, f1[m], acc[m], 0, 0, 0);
                f0[m] = *reinterpret_cast<const half8*>(smem + buf + xa[dx] + (2 * m + dy) * RB);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(c.q.v, f8[m].v, acc[m], 0, 0, 0, sa, 0, sb);
                f1[m] = *reinterpret_cast<const half8*>(smem + buf + xb[dx] + (2 * m + dy) * RB);
                f8[m].q[0] = *reinterpret_cast<const uint4*>(smem + buf + xc[dx] + (2 * m + dy) * RB);
                f8[m].q[1] = *reinterpret_cast<const uint4*>(smem + buf + xd[dx] + (2 * m + dy) * RB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float sum = 0;
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) sum += acc[m][r];
    out[blockIdx.x * 256 + tid] = sum;
}

int main() {
    if (!check_layout()) return 1;
    char* w; float* out;
    CHECK(hipMalloc(&w, 8 << 20)); CHECK(hipMemset(w, 0x38, 8 << 20)); CHECK(hipMalloc(&out, 512 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 512, groups = 400;                 // 400 groups of 9 taps
    CHECK(hipFuncSetAttribute((const void*)loop_f16x3<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 10 * ROWB));
    CHECK(hipFuncSetAttribute((const void*)loop_fp8c<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 10 * 20 * 128));
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        // the same arithmetic: `groups` 32-channel chunks x 9 taps = 2 * groups * 9 sixteen-channel steps
        hipLaunchKernelGGL(loop_f16x3<4>, dim3(blocks), dim3(256), 2 * 10 * ROWB, 0, w, out, 90);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(loop_f16x3<4>, dim3(blocks), dim3(256), 2 * 10 * ROWB, 0, w, out, 2 * groups * 9);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double a = ms * 1e6 / (2.0 * groups * 9);
        printf("f16 x 3   : %.3f ms, %.1f ns per 16-channel tap step and CU (pipe floor 12 MFMAs x 2 waves x 16.8 ns = 403 at 1.9 GHz)\n", ms, a);
        hipLaunchKernelGGL(loop_fp8c<4>, dim3(blocks), dim3(256), 2 * 10 * 20 * 128, 0, w, out, 90, 0x7f7f7f7f, 0x74747474);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(loop_fp8c<4>, dim3(blocks), dim3(256), 2 * 10 * 20 * 128, 0, w, out, groups * 9, 0x7f7f7f7f, 0x74747474);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double b = ms * 1e6 / (2.0 * groups * 9);
        printf("f16 + fp8 : %.3f ms, %.1f ns per 16-channel tap step and CU (pipe floor 269)  -> %.2fx\n", ms, b, a / b);
    }
    return 0;
}
