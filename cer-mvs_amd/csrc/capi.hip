// Library-level entry points of libcermvs.so (include/cer_mvs.h).
#include "common.hpp"

extern "C" int cer_abi_version(void) { return 1028; }

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
