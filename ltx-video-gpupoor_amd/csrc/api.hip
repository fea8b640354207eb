// api.hip -- version / error plumbing of libltxmi.so (host code only).
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace ltxmi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LTXMI_ERR_LAUNCH;
    }
    return LTXMI_OK;
}

int reserve_lds(const void* kernel, int bytes, unsigned long long* done, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_error("%s: cannot query the current device", what);
        return LTXMI_ERR_LAUNCH;
    }
    const bool tracked = dev >= 0 && dev < 64;
    const unsigned long long bit = tracked ? (1ull << dev) : 0ull;
    if (tracked && (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit)) return LTXMI_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("%s: cannot reserve %d bytes of LDS: %s", what, bytes, hipGetErrorString(e));
        return LTXMI_ERR_LAUNCH;
    }
    if (tracked) __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
    return LTXMI_OK;
}

int device_cu_count(const char* what) {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_error("%s: cannot query the current device", what);
        return -1;
    }
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked) {
        const int c = __atomic_load_n(&cus[dev], __ATOMIC_ACQUIRE);
        if (c > 0) return c;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        set_error("%s: cannot query the CU count of device %d", what, dev);
        return -1;
    }
    if (tracked) __atomic_store_n(&cus[dev], n, __ATOMIC_RELEASE);
    return n;
}

}  // namespace ltxmi

extern "C" const char* ltxmi_version(void) { return "ltxmi 0.5.0 (round 4)"; }
extern "C" const char* ltxmi_last_error(void) { return ltxmi::g_err; }
extern "C" const char* ltxmi_arch(void) { return "gfx950"; }
