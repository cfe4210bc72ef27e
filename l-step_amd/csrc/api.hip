// Error plumbing + ABI version of liblstep_hip.so
#include <stdarg.h>
#include <stdio.h>

#include "lstep_common.h"

namespace lstep {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LSTEP_EHIP, "%s: %s", what, hipGetErrorString(e));
    return LSTEP_OK;
}

}  // namespace lstep

extern "C" int lstep_abi_version(void) { return LSTEP_ABI_VERSION; }
extern "C" const char* lstep_last_error(void) { return lstep::g_err; }
