// Micro-benchmark: LDS-fed v_mfma_f32_32x32x16_f16 at the operand ratios of the f16x3 conv (per k16-step of a wave with a
// WM x WN register tile: 2*WM A-fragment + 2*WN B-fragment ds_read_b128, 3*WM*WN MFMAs), fragments double-buffered in registers.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_lds.hip -o tools/ubench/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return; } } while (0)

template <int WM, int WN, int MODE>   // MODE 0: reads then MFMAs (compiler order); 1: sched_group_barrier interleave; 2: no LDS (registers)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kg = lane >> 5;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 1e-3f;
    __syncthreads();
    const char* A = smem + (wave & 3) * 34 * 144;                 // activation rows, 144-B pixel stride
    const char* B = smem + 48 * 1024 + (wave >> 2) * 8192;        // weight fragments, lane-linear
    floatx16 accm[WM][WN], accl[WM][WN];
    for (int m = 0; m < WM; ++m) for (int n = 0; n < WN; ++n) for (int r = 0; r < 16; ++r) { accm[m][n][r] = 0; accl[m][n][r] = 0; }
    half8 ah[2][WM], al[2][WM], bh[2][WN], bl[2][WN];
    auto load = [&](int buf, int it) {
        const int tap = it % 9;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const char* p = A + ((m + tap / 3) * 34 + li + tap % 3) * 144 + (it & 1) * 32 + kg * 16;
            ah[buf][m] = *reinterpret_cast<const half8*>(p);
            al[buf][m] = *reinterpret_cast<const half8*>(p + 64);
        }
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const char* q = B + ((n * 2 + (it & 1)) * 2) * 1024 + lane * 16;
            bh[buf][n] = *reinterpret_cast<const half8*>(q);
            bl[buf][n] = *reinterpret_cast<const half8*>(q + 1024);
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                accm[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[buf][m], bh[buf][n], accm[m][n], 0, 0, 0);
                accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[buf][m], bl[buf][n], accl[m][n], 0, 0, 0);
                accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[buf][m], bh[buf][n], accl[m][n], 0, 0, 0);
            }
    };
    load(0, 0);
    if (MODE == 2) load(1, 1);
    for (int it = 0; it < iters; it += 2) {
        if (MODE != 2) load(1, it + 1);
        mma(0);
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 2 * WM + 2 * WN; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 2) load(0, it + 2);
        mma(1);
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 2 * WM + 2 * WN; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int m = 0; m < WM; ++m) for (int n = 0; n < WN; ++n) for (int r = 0; r < 16; ++r) s += accm[m][n][r] + accl[m][n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WM, int WN, int MODE>
static void run(const char* name, int waves_per_block, int blocks_per_cu) {
    const int iters = 4000, blocks = 256 * blocks_per_cu, thr = waves_per_block * 64;
    const size_t smem = 72 * 1024;
    float* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 1024 * 4));
    CHECK(hipFuncSetAttribute((const void*)k<WM, WN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<WM, WN, MODE>), dim3(blocks), dim3(thr), smem, 0, out, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<WM, WN, MODE>), dim3(blocks), dim3(thr), smem, 0, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mf = (double)iters * 3 * WM * WN * waves_per_block * blocks_per_cu / 4.0;   // MFMAs per SIMD
    printf("%-44s %d x %d waves/CU: %.3f ms, %.1f ns per MFMA per SIMD (floor 16.8)\n", name, blocks_per_cu, waves_per_block, ms, ms * 1e6 / mf);
    CHECK(hipFree(out));
}

int main() {
    run<1, 2, 2>("1x2 tile, registers only", 8, 2);
    run<1, 2, 0>("1x2 tile (32px x 64ch), LDS, compiler order", 8, 2);
    run<1, 2, 1>("1x2 tile, LDS, MFMA/DS interleaved", 8, 2);
    run<2, 2, 0>("2x2 tile (64px x 64ch), LDS, compiler order", 8, 1);
    run<2, 2, 1>("2x2 tile, LDS, MFMA/DS interleaved", 8, 1);
    run<2, 2, 0>("2x2 tile, LDS, compiler order", 4, 2);
    run<2, 2, 1>("2x2 tile, LDS, MFMA/DS interleaved", 4, 2);
    run<2, 2, 0>("2x2 tile, LDS, compiler order", 4, 1);
    run<2, 2, 1>("2x2 tile, LDS, MFMA/DS interleaved", 4, 1);
    return 0;
}
