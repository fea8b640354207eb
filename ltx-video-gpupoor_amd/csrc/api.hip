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

}  // namespace ltxmi

extern "C" const char* ltxmi_version(void) { return "ltxmi 0.1.0 (round 1)"; }
extern "C" const char* ltxmi_last_error(void) { return ltxmi::g_err; }
extern "C" const char* ltxmi_arch(void) { return "gfx950"; }
