// Library-level entry points of libcermvs.so (include/cer_mvs.h).
#include "common.hpp"

extern "C" int cer_abi_version(void) { return 1040; }

extern "C" const char* cer_error_string(int code) {
    switch (code) {
        case CER_OK: return "ok";
        case CER_EINVAL: return "CER_EINVAL: null pointer or non-positive size";
        case CER_ESHAPE: return "CER_ESHAPE: shape not supported by this build";
        case CER_EALIGN: return "CER_EALIGN: pointer not 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown cer error";
    }
}

extern "C" int cer_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -(int)e - 100;
    return n;
}

// ---- sticky overflow flag: a caller-owned device int; kernels that saturate a split-f16 operand or into it (bit 1: feature rows of
// the cost volume, cer_feat_split_f16 takes its flag explicitly; bit 2: the hidden map of the fused delta head; cer_f16_scan_overflow
// sets the bit it is given).  One per process (one process per GPU); NULL (default) disables the in-kernel checks.
static int* g_overflow_flag = nullptr;
int* cer_overflow_flag_get() { return g_overflow_flag; }
extern "C" int cer_overflow_flag(int* flag) {
    g_overflow_flag = flag;
    return CER_OK;
}
