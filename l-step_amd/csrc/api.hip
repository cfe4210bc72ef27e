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

// A HIP stream of the caller's own (hipStreamNonBlocking, like the framework's).  The host layer keeps ONE per role (update_pe, ring copies,
// weight-gradient products, edge re-gather, graph capture, pulled rows): PyTorch hands its streams out round-robin from a pool of 32 per
// device, so in a long process two roles -- or a role and the framework's own graph-capture stream -- end up on the SAME queue (round 4:
// the engine's update stream was torch.cuda.graph's capture stream in exactly the test that faulted).
extern "C" int lstep_stream_create(void** out_stream, int32_t priority) {
    if (!out_stream) return lstep::set_error(LSTEP_EINVAL, "lstep_stream_create: NULL pointer");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);      // (numerically: greatest <= least)
    if (e != hipSuccess) return lstep::set_error(LSTEP_EHIP, "lstep_stream_create: %s", hipGetErrorString(e));
    int prio = priority;
    if (prio < greatest) prio = greatest;
    if (prio > least) prio = least;
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio);
    if (e != hipSuccess) return lstep::set_error(LSTEP_EHIP, "lstep_stream_create: %s", hipGetErrorString(e));
    *out_stream = (void*)s;
    return LSTEP_OK;
}

extern "C" int lstep_stream_destroy(void* stream) {
    if (!stream) return LSTEP_OK;
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) return lstep::set_error(LSTEP_EHIP, "lstep_stream_destroy: %s", hipGetErrorString(e));
    return LSTEP_OK;
}
