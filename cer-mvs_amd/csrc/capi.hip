// Library-level entry points of libcermvs.so (include/cer_mvs.h).
#include "common.hpp"
#include <string.h>

extern "C" int cer_abi_version(void) { return 1070; }

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

// ---- sticky overflow flag: a caller-owned device int PER DEVICE; kernels that saturate a split-f16 operand or into it (bit 1:
// feature rows of the cost volume, cer_feat_split_f16 takes its flag explicitly; bit 2: the hidden map of the fused delta head;
// cer_f16_scan_overflow sets the bit it is given).  cer_overflow_flag registers the flag for the CURRENT device (hipGetDevice) and a
// launch uses the flag of the device it runs on - a process that drives two GPUs no longer lets the second registration redirect
// the first GPU's kernels into foreign memory.  NULL (default) disables the in-kernel checks on that device.
#define CER_MAX_DEVICES 64
static int* g_overflow_flag[CER_MAX_DEVICES];
int* cer_overflow_flag_get() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CER_MAX_DEVICES) return nullptr;
    return g_overflow_flag[dev];
}
extern "C" int cer_overflow_flag(int* flag) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return flag ? CER_EINVAL : CER_OK;      // (no device: nothing to register; clearing is a no-op)
    if (dev < 0 || dev >= CER_MAX_DEVICES) return CER_ESHAPE;
    g_overflow_flag[dev] = flag;
    return CER_OK;
}

// ---- number of CUs of the current device (cached): what persistent grids and tile-height choices are sized for
#include <atomic>
int cer_num_cus() {
    static std::atomic<int> dev_cus[CER_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CER_MAX_DEVICES) dev = 0;
    int n = dev_cus[dev].load();
    if (n == 0) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        dev_cus[dev].store(n);
    }
    return n;
}

// ---- LDS bytes per CU of the current device (cached): residency estimates of the persistent grids (160 KiB on gfx950)
int cer_lds_per_cu() {
    static std::atomic<int> dev_lds[CER_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CER_MAX_DEVICES) dev = 0;
    int n = dev_lds[dev].load();
    if (n == 0) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v > 0) ? v : 64 * 1024;
        // (some ROCm releases report the per-workgroup limit here: a gfx950 CU has 160 KiB whatever the attribute says)
        hipDeviceProp_t prop;
        if (n < 160 * 1024 && hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) n = 160 * 1024;
        dev_lds[dev].store(n);
    }
    return n;
}
