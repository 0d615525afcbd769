// Micro-test: ds_write_b128 -> s_waitcnt lgkmcnt(0) -> s_barrier -> another wave's ds_read_b128 of the same bytes.  With NR reads queued in
// front of the store.  Every wave writes its own 1-KiB region and, after the barrier, reads the next wave's region.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_barrier_vis.hip -o tools/ubench/lds_barrier_vis
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <int NR>
__global__ __launch_bounds__(256) void k(unsigned* bad_lanes, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) lds[i] = 0;
    __syncthreads();
    const unsigned raddr = (unsigned)(tid * 16);
    const unsigned waddr = (unsigned)(32768 + tid * 16);
    const unsigned naddr = (unsigned)(32768 + (((wave + 1) & 3) * 64 + lane) * 16);
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned val = (unsigned)(0x1000 * (it + 1));
        unsigned g0, g3;
        asm volatile(
            "v_mov_b32 v10, %4\n v_mov_b32 v11, %4\n v_mov_b32 v12, %4\n v_mov_b32 v13, %4\n"
            "s_nop 4\n"
            ".rept %6\n ds_read_b128 v[20:23], %2\n ds_read_b128 v[24:27], %2 offset:4096\n .endr\n"
            "ds_write_b128 %3, v[10:13]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_barrier\n"
            "ds_read_b128 v[28:31], %5\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_mov_b32 %0, v28\n v_mov_b32 %1, v31\n"
            "s_barrier\n"
            : "=v"(g0), "=v"(g3) : "v"(raddr), "v"(waddr), "v"(val), "v"(naddr), "n"(NR / 2)
            : "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "memory");
        if (g0 != val || g3 != val) bad = 1;
    }
    if (bad) atomicOr(&bad_lanes[lane], 1u);
}

template <int NR>
static int run(unsigned* d) {
    unsigned h[64];
    CHECK(hipMemset(d, 0, 256));
    hipLaunchKernelGGL((k<NR>), dim3(1024), dim3(256), 0, 0, d, 400);
    CHECK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
    int n = 0;
    for (int l = 0; l < 64; ++l) n += h[l] != 0;
    printf("%2d LDS reads queued in front of the store: %2d of 64 lanes read stale data behind the barrier\n", NR, n);
    return 0;
}

int main() {
    unsigned* d;
    CHECK(hipMalloc(&d, 256));
    run<0>(d); run<8>(d); run<32>(d);
    return 0;
}
