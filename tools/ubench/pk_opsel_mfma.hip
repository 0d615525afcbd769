// Micro-reproducer (round 4): packed-fp32 VALU instructions whose operand selection crosses the halves of a 64-bit register pair
// (VOP3P op_sel / op_sel_hi) give intermittently wrong results on MI355X (gfx950) while waves of an f16 MFMA kernel (e.g. a hipBLASLt
// GEMM on another HIP stream of the same process) are resident on the same CU.  Each thread evaluates one instruction form on operands
// derived from (thread, iteration), compares both result halves with the scalar evaluation and counts mismatches.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC pk_opsel_mfma.hip -o libpkopsel.so ;  python tools/ubench/pk_opsel_mfma.py
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define FORM(ID, ASM, LO, HI)                                                                                       \
    if (form == ID) {                                                                                               \
        for (int it = 0; it < iters; ++it) {                                                                        \
            const float s = (float)(it & 1023);                                                                     \
            f2 a = (f2){base + s, base * 0.5f - s}, b = (f2){3.0f * base - s, 0.25f * base + 2.0f * s}, c = (f2){s, -s}, d;  \
            asm volatile("s_nop 0" : "+v"(a), "+v"(b), "+v"(c));                                                    \
            asm volatile(ASM : "=&v"(d) : "v"(a), "v"(b), "v"(c));                                                  \
            float lo = LO, hi = HI;                                                                                 \
            asm volatile("" : "+v"(lo), "+v"(hi));                                                                  \
            if (d.x != lo) ++bad_lo;                                                                                \
            if (d.y != hi) ++bad_hi;                                                                                \
        }                                                                                                           \
    }

// scalar references are single IEEE operations (no contraction: compile with -ffp-contract=off)
__global__ __launch_bounds__(256) void pk_opsel_kernel(int form, int iters, unsigned long long* out) {
    const float base = (float)(threadIdx.x + 256 * (blockIdx.x & 63)) * 0.001f + 1.0f;
    unsigned bad_lo = 0, bad_hi = 0;
    FORM(0, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]", a.x + b.y, a.y + b.x)            // halves of src1 swapped (the lookup's failing form)
    FORM(1, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]", a.y + b.x, a.x + b.y)            // halves of src0 swapped
    FORM(2, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]", a.x * b.y, a.y * b.x)            // mul, src1 swapped
    FORM(3, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]", fmaf(a.x, b.y, c.x), fmaf(a.y, b.x, c.y))   // fma, src1 swapped
    FORM(4, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]", a.x + b.y, a.y + b.y)            // src1 high half to both
    FORM(5, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]", a.x + b.x, a.y + b.x)                          // src1 low half to both
    FORM(6, "v_pk_add_f32 %0, %1, %2", a.x + b.x, a.y + b.y)                                          // control: no selection
    FORM(7, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]", fmaf(a.y, b.x, c.x), fmaf(a.y, b.y, c.y))    // src0 high half to both (shipped lookup's conv)
    FORM(8, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]", a.x * b.x, a.y * b.x)                          // src1 low half to both (very common)
    FORM(9, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]", a.y * b.x, a.y * b.y)                             // src0 high half to both
    FORM(10, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0]", a.x + b.y, a.y + b.x)   // (form 0 again)
    // packed f16 forms (one 32-bit register per operand; op_sel selects the 16-bit halves)
    if (form >= 11 && form <= 13) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        for (int it = 0; it < iters; ++it) {
            const float s = (float)(it & 255);
            h2 a = (h2){(_Float16)(base + s), (_Float16)(base * 0.5f - s)}, b = (h2){(_Float16)(3.0f * base - s), (_Float16)(0.25f * base + 2.0f * s)}, d;
            asm volatile("s_nop 0" : "+v"(a), "+v"(b));
            _Float16 lo, hi;
            if (form == 11) { asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b)); lo = a.x + b.y; hi = a.y + b.x; }
            else if (form == 12) { asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b)); lo = a.x * b.y; hi = a.y * b.x; }
            else { asm volatile("v_pk_add_f16 %0, %1, %2" : "=&v"(d) : "v"(a), "v"(b)); lo = a.x + b.x; hi = a.y + b.y; }
            asm volatile("" : "+v"(lo), "+v"(hi));
            if (d.x != lo) ++bad_lo;
            if (d.y != hi) ++bad_hi;
        }
    }
    if (bad_lo) atomicAdd(&out[0], (unsigned long long)bad_lo);
    if (bad_hi) atomicAdd(&out[1], (unsigned long long)bad_hi);
}

extern "C" int pk_opsel_launch(int form, int iters, int blocks, unsigned long long* out, void* stream) {
    hipLaunchKernelGGL(pk_opsel_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, form, iters, out);
    return (int)hipGetLastError();
}
