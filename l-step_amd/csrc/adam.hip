// Adam for the whole L-STEP parameter set as ONE short launch (train_LSTEP_link_prediction.py:283 `optimizer.step()`, utils/utils.py:49-67:
// plain torch.optim.Adam, no amsgrad).  The parameter set is ~45 small tensors (0.58 M floats in all); the framework's multi-tensor
// kernel hands a workgroup 64 K elements, so its run time is one workgroup's serial loop over the largest tensor (40 us, whatever the
// batch size).  Here a workgroup takes 1024 elements, the tensor table travels in the kernel arguments, and the launch lasts a few us.
#include "lstep_common.h"

namespace lstep {

struct AdamArgs {
    float* param[LSTEP_ADAM_MAX_TENSORS];
    const float* grad[LSTEP_ADAM_MAX_TENSORS];
    float* exp_avg[LSTEP_ADAM_MAX_TENSORS];
    float* exp_avg_sq[LSTEP_ADAM_MAX_TENSORS];
    int32_t first_chunk[LSTEP_ADAM_MAX_TENSORS + 1];   // workgroup index where each tensor starts
    int32_t numel[LSTEP_ADAM_MAX_TENSORS];
    int32_t step_index[LSTEP_ADAM_MAX_TENSORS];        // which entry of `steps` belongs to the tensor
    const float* steps;                                // step counts INCLUDING this step (device, float32 like torch's)
    int32_t n;
    float lr, beta1, beta2, eps, weight_decay;
};

constexpr int kAdamChunk = 1024;   // elements per workgroup: 256 threads x 4

__global__ __launch_bounds__(kBlock) void adam_kernel(const AdamArgs a) {
    // which tensor: the table is tiny and sorted
    int t = 0;
    const int b = blockIdx.x;
    while (t + 1 < a.n && a.first_chunk[t + 1] <= b) ++t;
    const int base = (b - a.first_chunk[t]) * kAdamChunk + threadIdx.x * 4;
    const int n = a.numel[t];
    if (base >= n) return;
    // torch's _fused_adam_ (fused_adam_utils.cuh): bias corrections in double, moments and the update in float
    const double step = (double)a.steps[a.step_index[t]];
    const float bc1 = (float)(1.0 - pow((double)a.beta1, step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, step));
    const float step_size = a.lr / bc1;
    float* p = a.param[t] + base;
    const float* g = a.grad[t] + base;
    float* m = a.exp_avg[t] + base;
    float* v = a.exp_avg_sq[t] + base;
    const int cnt = (n - base) < 4 ? (n - base) : 4;
    const bool vec = cnt == 4 && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    float pv[4], gv[4], mv[4], vv[4];
    if (vec) {
        const float4 p4 = ld4(p), g4 = ld4(g), m4 = ld4(m), v4 = ld4(v);
        pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
        vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
    } else {
        for (int i = 0; i < cnt; ++i) { pv[i] = p[i]; gv[i] = g[i]; mv[i] = m[i]; vv[i] = v[i]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= cnt) break;
        float gi = gv[i];
        if (a.weight_decay != 0.f) gi = fmaf(a.weight_decay, pv[i], gi);
        mv[i] = mv[i] + (1.f - a.beta1) * (gi - mv[i]);            // lerp(exp_avg, grad, 1 - beta1)
        vv[i] = a.beta2 * vv[i] + (1.f - a.beta2) * gi * gi;
        const float denom = sqrtf(vv[i]) / bc2_sqrt + a.eps;
        pv[i] -= step_size * mv[i] / denom;
    }
    if (vec) {
        st4(p, make_float4(pv[0], pv[1], pv[2], pv[3]));
        st4(m, make_float4(mv[0], mv[1], mv[2], mv[3]));
        st4(v, make_float4(vv[0], vv[1], vv[2], vv[3]));
    } else {
        for (int i = 0; i < cnt; ++i) { p[i] = pv[i]; m[i] = mv[i]; v[i] = vv[i]; }
    }
}

}  // namespace lstep

using namespace lstep;

extern "C" int lstep_adam_step(int32_t num_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, const float* steps, const int32_t* step_index, float lr, float beta1, float beta2, float eps,
                               float weight_decay, void* stream) {
    if (num_tensors < 0 || num_tensors > LSTEP_ADAM_MAX_TENSORS)
        return set_error(LSTEP_EINVAL, "lstep_adam_step: %d tensors (at most %d per call)", num_tensors, LSTEP_ADAM_MAX_TENSORS);
    if (num_tensors == 0) return LSTEP_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !steps) return set_error(LSTEP_EINVAL, "lstep_adam_step: NULL pointer");
    AdamArgs a;
    int64_t chunks = 0;
    for (int t = 0; t < num_tensors; ++t) {
        if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t] || numel[t] <= 0 || numel[t] > (int64_t)1 << 30)
            return set_error(LSTEP_EINVAL, "lstep_adam_step: tensor %d: NULL pointer or bad size", t);
        a.param[t] = params[t]; a.grad[t] = grads[t]; a.exp_avg[t] = exp_avg[t]; a.exp_avg_sq[t] = exp_avg_sq[t];
        a.numel[t] = (int32_t)numel[t];
        a.step_index[t] = step_index ? step_index[t] : t;
        a.first_chunk[t] = (int32_t)chunks;
        chunks += (numel[t] + kAdamChunk - 1) / kAdamChunk;
    }
    a.first_chunk[num_tensors] = (int32_t)chunks;
    a.steps = steps; a.n = num_tensors; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)chunks), dim3(kBlock), 0, (hipStream_t)stream, a);
    return check_launch("adam_kernel");
}
