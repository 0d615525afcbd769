// Micro-test: is the B operand of a v_mfma_f32_32x32x16_f16 that waits behind other MFMAs protected against a later overwrite of its
// registers (VALU write / LDS load return)?  Four independent MFMAs back to back (the conv's multiply step), then after NOPS wait states
// `v_mov` (MODE 0) or a ds_read_b32 return (MODE 1) into the first B register of the LAST one.  A = B = 1.0: every result must be 16.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_src_war.hip -o tools/ubench/mfma_src_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <int MODE, int NOPS>
__global__ __launch_bounds__(512) void k(unsigned* bad_lanes, int iters) {
    __shared__ unsigned lds[256];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 256) lds[tid] = 0;
    __syncthreads();
    unsigned bad = 0;
    const unsigned ones = 0x3c003c00u, laddr = (unsigned)((tid & 255) * 4);
    for (int it = 0; it < iters; ++it) {
        float r0, r1;
        asm volatile(
            "v_mov_b32 v40, %2\n v_mov_b32 v41, %2\n v_mov_b32 v42, %2\n v_mov_b32 v43, %2\n"          // A (shared)
            "v_mov_b32 v44, %2\n v_mov_b32 v45, %2\n v_mov_b32 v46, %2\n v_mov_b32 v47, %2\n"          // B of MFMAs 0-2
            "v_mov_b32 v48, %2\n v_mov_b32 v49, %2\n v_mov_b32 v50, %2\n v_mov_b32 v51, %2\n"          // B of the last MFMA
            ".irp r,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127\n v_mov_b32 v\\r, 0\n .endr\n"
            "s_nop 7\n"
            "v_mfma_f32_32x32x16_f16 v[64:79], v[40:43], v[44:47], v[64:79]\n"
            "v_mfma_f32_32x32x16_f16 v[80:95], v[40:43], v[44:47], v[80:95]\n"
            "v_mfma_f32_32x32x16_f16 v[96:111], v[40:43], v[44:47], v[96:111]\n"
            "v_mfma_f32_32x32x16_f16 v[112:127], v[40:43], v[48:51], v[112:127]\n"
            ".rept %4\n s_nop 0\n .endr\n"
            ".if %5 == 0\n v_mov_b32 v48, 0\n .else\n ds_read_b32 v48, %3\n .endif\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_nop 15\n s_nop 15\n"
            "v_mov_b32 %0, v112\n v_mov_b32 %1, v127\n"
            : "=v"(r0), "=v"(r1) : "v"(ones), "v"(laddr), "n"(NOPS), "n"(MODE)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
              "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
              "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
              "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
              "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "memory");
        if (r0 != 16.0f || r1 != 16.0f) bad = 1;
    }
    if (bad) atomicOr(&bad_lanes[lane], 1u);
}

template <int MODE, int NOPS>
static int run(unsigned* d) {
    unsigned h[64];
    CHECK(hipMemset(d, 0, 256));
    hipLaunchKernelGGL((k<MODE, NOPS>), dim3(512), dim3(512), 0, 0, d, 200);
    CHECK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
    int n = 0;
    unsigned long long mask = 0;
    for (int l = 0; l < 64; ++l) if (h[l]) { ++n; mask |= 1ull << l; }
    printf("%s of the last MFMA's first B register %2d wait state(s) behind the group: %2d of 64 lanes hold a wrong result (lane mask %016llx)\n",
           MODE ? "LDS load into" : "VALU write to", NOPS, n, mask);
    return 0;
}

int main() {
    unsigned* d;
    CHECK(hipMalloc(&d, 256));
    run<0, 0>(d); run<0, 4>(d); run<0, 16>(d); run<0, 64>(d);
    run<1, 0>(d); run<1, 4>(d); run<1, 16>(d); run<1, 64>(d);
    return 0;
}
