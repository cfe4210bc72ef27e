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

#ifdef LSTEP_BOUNDS_CHECK
static void (*g_check_setters[64])(unsigned long long*);
static int g_num_check_setters = 0;
void register_check_setter(void (*fn)(unsigned long long*)) {
    if (g_num_check_setters < 64) g_check_setters[g_num_check_setters++] = fn;
}
static unsigned long long* g_check_buffer = nullptr;
static int check_buffer() {      // allocated on first use, handed to every translation unit
    if (g_check_buffer != nullptr) return LSTEP_OK;
    if (hipMalloc((void**)&g_check_buffer, 8 * sizeof(unsigned long long)) != hipSuccess) return set_error(LSTEP_EHIP, "lstep_debug: hipMalloc failed");
    if (hipMemset(g_check_buffer, 0, 8 * sizeof(unsigned long long)) != hipSuccess) return set_error(LSTEP_EHIP, "lstep_debug: hipMemset failed");
    for (int i = 0; i < g_num_check_setters; ++i) g_check_setters[i](g_check_buffer);
    return hipDeviceSynchronize() == hipSuccess ? LSTEP_OK : set_error(LSTEP_EHIP, "lstep_debug: set-up failed");
}
#endif

}  // namespace lstep

extern "C" int lstep_debug_bounds_check_enabled(void) {
#ifdef LSTEP_BOUNDS_CHECK
    return 1;
#else
    return 0;
#endif
}

extern "C" int lstep_debug_set_limits(int64_t node_rows, int64_t edge_rows) {
#ifdef LSTEP_BOUNDS_CHECK
    if (int rc = lstep::check_buffer()) return rc;
    const unsigned long long v[2] = {(unsigned long long)(node_rows > 0 ? node_rows : 0), (unsigned long long)(edge_rows > 0 ? edge_rows : 0)};
    if (hipMemcpy(lstep::g_check_buffer + 4, v, sizeof(v), hipMemcpyHostToDevice) != hipSuccess)
        return lstep::set_error(LSTEP_EHIP, "lstep_debug_set_limits: copy failed");
#else
    (void)node_rows; (void)edge_rows;
#endif
    return LSTEP_OK;
}

extern "C" int lstep_debug_device_error(int64_t out[4]) {
    if (!out) return lstep::set_error(LSTEP_EINVAL, "lstep_debug_device_error: NULL pointer");
    out[0] = out[1] = out[2] = out[3] = 0;
#ifdef LSTEP_BOUNDS_CHECK
    if (int rc = lstep::check_buffer()) return rc;
    unsigned long long v[4];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(v, lstep::g_check_buffer, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemset(lstep::g_check_buffer, 0, sizeof(v)) != hipSuccess)
        return lstep::set_error(LSTEP_EHIP, "lstep_debug_device_error: read-back failed");
    for (int i = 0; i < 4; ++i) out[i] = (int64_t)v[i];
#endif
    return LSTEP_OK;
}

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
