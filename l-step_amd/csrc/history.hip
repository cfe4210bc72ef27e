// F: linear core of fourier_transform_pe (reference models/LSTEP.py:104-137) over a strided PE history.
//   fwd  out[u, :]   = sum_s coef[s, :] * hist[node_u, s, :]          streams U * t_len rows of P floats once
//   bwd  dcoef[s, :] = sum_u grad[u, :] * hist[node_u, s, :]          streams the same rows once more
// Both are pure HBM streams of 4P-byte rows (688 B for P = 172); coef / grad rows are re-read from L2.
#include "lstep_common.h"

namespace lstep {

struct HistView {
    const float* base;
    int64_t node_stride, time_stride;
    int slots, rot;
    __device__ __forceinline__ const float* row(int64_t node, int s) const {
        int ph = s + rot;
        if (ph >= slots) ph -= slots;
        return base + node * node_stride + (int64_t)ph * time_stride;
    }
};

constexpr int kHistInFlight = 8;

__global__ __launch_bounds__(kBlock) void history_filter_fwd_kernel(HistView h, int t_len, int P, const int64_t* __restrict__ ids,
                                                                     int64_t num_ids, const float* __restrict__ coef,
                                                                     float* __restrict__ out) {
    const int lane = lane_id();
    const int64_t u = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (u >= num_ids) return;
    if (lane >= (P >> 2)) return;
    const int64_t node = ids[u];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < t_len; s += kHistInFlight) {
        float4 x[kHistInFlight], c[kHistInFlight];
#pragma unroll
        for (int i = 0; i < kHistInFlight; ++i) {
            const bool live = s + i < t_len;
            x[i] = live ? ld4_stream(h.row(node, s + i) + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            c[i] = live ? ld4(coef + (int64_t)(s + i) * P + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < kHistInFlight; ++i) {
            acc.x = fmaf(c[i].x, x[i].x, acc.x);
            acc.y = fmaf(c[i].y, x[i].y, acc.y);
            acc.z = fmaf(c[i].z, x[i].z, acc.z);
            acc.w = fmaf(c[i].w, x[i].w, acc.w);
        }
    }
    st4(out + u * (int64_t)P + lane * 4, acc);
}

constexpr int kBwdNodesPerChunk = 64;
constexpr int kBwdTimeGroup = 8;

// wave = (node chunk, group of 8 time steps): 8 float4 accumulators, one pass over the chunk's nodes
__global__ __launch_bounds__(kBlock) void history_filter_bwd_kernel(HistView h, int t_len, int P, const int64_t* __restrict__ ids,
                                                                     int64_t num_ids, const float* __restrict__ grad,
                                                                     float* __restrict__ partial, int groups) {
    const int lane = lane_id();
    const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t chunk = w / groups;
    const int grp = (int)(w - chunk * groups);
    const int64_t u0 = chunk * kBwdNodesPerChunk;
    if (u0 >= num_ids) return;
    if (lane >= (P >> 2)) return;
    const int s0 = grp * kBwdTimeGroup;
    const int64_t u1 = (u0 + kBwdNodesPerChunk < num_ids) ? u0 + kBwdNodesPerChunk : num_ids;
    float4 acc[kBwdTimeGroup];
#pragma unroll
    for (int i = 0; i < kBwdTimeGroup; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t u = u0; u < u1; ++u) {
        const int64_t node = ids[u];
        const float4 g = ld4(grad + u * (int64_t)P + lane * 4);
        float4 x[kBwdTimeGroup];
#pragma unroll
        for (int i = 0; i < kBwdTimeGroup; ++i)
            x[i] = (s0 + i < t_len) ? ld4_stream(h.row(node, s0 + i) + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < kBwdTimeGroup; ++i) {
            acc[i].x = fmaf(g.x, x[i].x, acc[i].x);
            acc[i].y = fmaf(g.y, x[i].y, acc[i].y);
            acc[i].z = fmaf(g.z, x[i].z, acc[i].z);
            acc[i].w = fmaf(g.w, x[i].w, acc[i].w);
        }
    }
#pragma unroll
    for (int i = 0; i < kBwdTimeGroup; ++i)
        if (s0 + i < t_len) st4(partial + (chunk * t_len + s0 + i) * (int64_t)P + lane * 4, acc[i]);
}

static int check_hist(const char* who, const float* hist, int64_t node_stride, int64_t time_stride, int slots, int rot, int t_len, int P) {
    if (!hist) return set_error(LSTEP_EINVAL, "%s: NULL history", who);
    if (P <= 0 || (P & 3) || P > 4 * kMaxRowVec) return set_error(LSTEP_EINVAL, "%s: unsupported pe_dim %d", who, P);
    if ((node_stride & 3) || (time_stride & 3) || (((uintptr_t)hist) & 15)) return set_error(LSTEP_EINVAL, "%s: history rows must be 16-byte aligned", who);
    if (slots <= 0 || rot < 0 || rot >= slots || t_len < 0 || t_len > slots) return set_error(LSTEP_EINVAL, "%s: bad time window (slots=%d rot=%d t_len=%d)", who, slots, rot, t_len);
    return LSTEP_OK;
}

}  // namespace lstep

using namespace lstep;

extern "C" int lstep_history_filter_fwd(const float* hist, int64_t node_stride, int64_t time_stride, int32_t time_slots,
                                        int32_t time_rot, int32_t t_len, int32_t pe_dim, const int64_t* node_ids, int64_t num_ids,
                                        const float* coef, float* out, void* stream) {
    if (num_ids < 0) return set_error(LSTEP_EINVAL, "lstep_history_filter_fwd: negative count");
    if (num_ids == 0) return LSTEP_OK;
    if (int rc = check_hist("lstep_history_filter_fwd", hist, node_stride, time_stride, time_slots, time_rot, t_len, pe_dim)) return rc;
    if (!node_ids || !coef || !out) return set_error(LSTEP_EINVAL, "lstep_history_filter_fwd: NULL pointer");
    HistView h{hist, node_stride, time_stride, time_slots, time_rot};
    const unsigned grid = (unsigned)((num_ids + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(history_filter_fwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, h, (int)t_len, (int)pe_dim, node_ids,
                       num_ids, coef, out);
    return check_launch("history_filter_fwd_kernel");
}

extern "C" int64_t lstep_history_filter_bwd_chunks(int64_t num_ids) {
    return num_ids <= 0 ? 0 : (num_ids + kBwdNodesPerChunk - 1) / kBwdNodesPerChunk;
}

extern "C" int lstep_history_filter_bwd(const float* hist, int64_t node_stride, int64_t time_stride, int32_t time_slots,
                                        int32_t time_rot, int32_t t_len, int32_t pe_dim, const int64_t* node_ids, int64_t num_ids,
                                        const float* grad_out, float* out_partial, void* stream) {
    if (num_ids < 0) return set_error(LSTEP_EINVAL, "lstep_history_filter_bwd: negative count");
    if (num_ids == 0 || t_len == 0) return LSTEP_OK;
    if (int rc = check_hist("lstep_history_filter_bwd", hist, node_stride, time_stride, time_slots, time_rot, t_len, pe_dim)) return rc;
    if (!node_ids || !grad_out || !out_partial) return set_error(LSTEP_EINVAL, "lstep_history_filter_bwd: NULL pointer");
    HistView h{hist, node_stride, time_stride, time_slots, time_rot};
    const int groups = (t_len + kBwdTimeGroup - 1) / kBwdTimeGroup;
    const int64_t waves = lstep_history_filter_bwd_chunks(num_ids) * groups;
    const unsigned grid = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(history_filter_bwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, h, (int)t_len, (int)pe_dim, node_ids,
                       num_ids, grad_out, out_partial, groups);
    return check_launch("history_filter_bwd_kernel");
}
