// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 under the accumulator-dependency patterns of the f16x3 conv.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_issue.hip -o tools/ubench/mfma_issue && tools/ubench/mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int PAT>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    half8 a0, a1, b0, b1, b2, b3;
    for (int e = 0; e < 8; ++e) {
        a0[e] = (_Float16)(threadIdx.x * 0.001f + e); a1[e] = (_Float16)(0.5f + e);
        b0[e] = (_Float16)(e * 0.25f); b1[e] = (_Float16)(1.f + e); b2[e] = (_Float16)(2.f - e); b3[e] = (_Float16)(0.125f * e);
    }
    floatx16 m0, m1, l0, l1, x0, x1;
    for (int r = 0; r < 16; ++r) { m0[r] = 0; m1[r] = 0; l0[r] = 0; l1[r] = 0; x0[r] = 0; x1[r] = 0; }
    for (int i = 0; i < iters; ++i) {
        if (PAT == 0) {          // conv order: per n-tile accm, accl, accl (dependent pair back to back)
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, m0, 0, 0, 0);
            l0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, l0, 0, 0, 0);
            l0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, l0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b2, m1, 0, 0, 0);
            l1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b3, l1, 0, 0, 0);
            l1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, l1, 0, 0, 0);
        } else if (PAT == 1) {   // dependent pairs 2 apart
            l0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, l0, 0, 0, 0);
            l1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b3, l1, 0, 0, 0);
            l0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, l0, 0, 0, 0);
            l1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, l1, 0, 0, 0);
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b2, m1, 0, 0, 0);
        } else {                 // six independent accumulators
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, m0, 0, 0, 0);
            l0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, l0, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, x0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b2, m1, 0, 0, 0);
            l1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b3, l1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, x1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += m0[r] + m1[r] + l0[r] + l1[r] + x0[r] + x1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int PAT>
static void run(const char* name, int waves_per_cu) {
    const int iters = 4000, blocks = 256, thr = waves_per_cu * 64;
    float* out; hipMalloc(&out, blocks * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(thr), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(thr), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)iters * 6 * waves_per_cu / 4.0;
    const double tf = (double)iters * 6 * waves_per_cu * blocks * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/CU %2d: %.3f ms, %.1f ns per MFMA per SIMD, %.0f TF (f16 dense)\n", name, waves_per_cu, ms, ms * 1e6 / mfmas_per_simd, tf);
    hipFree(out);
}

int main() {
    for (int w : {4, 8}) {
        run<0>("conv order (dep. pair adjacent)", w);
        run<1>("dep. pair 2 apart", w);
        run<2>("independent accumulators", w);
    }
    return 0;
}
