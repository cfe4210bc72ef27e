// U1 / U2 message accumulation of update_pe without the dense [N+1, P+D] scatter target
//   reference: models/LSTEP.py:282-290 (phase 1: per batch edge, both directions), :319-322 (phase 2: per sampled
//   neighbour).  Entries are pre-grouped by destination row; one wave sums one segment in entry order.
#include "lstep_common.h"

namespace lstep {

constexpr int kSegInFlight = 4;

__global__ __launch_bounds__(kBlock) void segment_pe_time_sum_kernel(const float* __restrict__ pe, int P, const float* __restrict__ tw,
                                                                      const float* __restrict__ tb, int D,
                                                                      const int64_t* __restrict__ seg_begin, const int64_t* __restrict__ seg_end, int64_t num_segments,
                                                                      const int32_t* __restrict__ ent_row, const float* __restrict__ ent_dt,
                                                                      const uint8_t* __restrict__ ent_valid, float* __restrict__ out, int ld_out) {
    const int lane = lane_id();
    const int64_t s = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (s >= num_segments) return;
    const bool pa = lane < (P >> 2);
    const int64_t e0 = seg_begin[s], e1 = seg_end[s];
    const float w0 = lane < D ? tw[lane] : 0.f, b0 = lane < D ? tb[lane] : 0.f;
    const float w1 = lane + kWave < D ? tw[lane + kWave] : 0.f, b1 = lane + kWave < D ? tb[lane + kWave] : 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float t0 = 0.f, t1 = 0.f;
    for (int64_t c0 = e0; c0 < e1; c0 += kWave) {
        const int m = (int)((e1 - c0) < kWave ? (e1 - c0) : kWave);
        int r = 0, ok = 0;
        float dt = 0.f;
        if (lane < m) {
            r = ent_row[c0 + lane];
            dt = ent_dt[c0 + lane];
            ok = ent_valid ? (int)ent_valid[c0 + lane] : 1;
        }
        for (int j = 0; j < m; j += kSegInFlight) {
            float4 x[kSegInFlight];
#pragma unroll
            for (int u = 0; u < kSegInFlight; ++u) {
                const bool live = j + u < m;
                const int64_t rj = bcast_i32(r, live ? j + u : m - 1);
                x[u] = (pa && live) ? ld4(pe + rj * P + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < kSegInFlight; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
        for (int j = 0; j < m; ++j) {
            if (bcast_i32(ok, j) == 0) continue;
            const float dj = bcast_f32(dt, j);
            if (lane < D) t0 += time_feat(dj, w0, b0);
            if (lane + kWave < D) t1 += time_feat(dj, w1, b1);
        }
    }
    float* o = out + s * (int64_t)ld_out;
    if (pa) st4(o + lane * 4, acc);
    if (lane < D) o[P + lane] = t0;
    if (lane + kWave < D) o[P + lane + kWave] = t1;
    const int c = P + D + lane * 4;  // zero the padding columns
    if (c < ld_out) st4(o + c, make_float4(0.f, 0.f, 0.f, 0.f));
}

__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(float* __restrict__ table, int W, const int64_t* __restrict__ ids,
                                                               int64_t num_ids, const float* __restrict__ rows) {
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (i >= num_ids) return;
    const int64_t r = ids[i];
    for (int c = lane; c < (W >> 2); c += kWave) st4(table + r * W + c * 4, ld4(rows + i * W + c * 4));
}

// table[ids[i], :] += tanh(z[i, :])   (residual PE update of models/LSTEP.py:299-303 and :335-339, fused with the write)
__global__ __launch_bounds__(kBlock) void residual_tanh_rows_kernel(float* __restrict__ table, int W, const int64_t* __restrict__ ids,
                                                                     int64_t num_ids, const float* __restrict__ z, int ld_z) {
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (i >= num_ids) return;
    const int64_t r = ids[i];
    for (int c = lane; c < (W >> 2); c += kWave) {
        const float4 a = ld4(table + r * W + c * 4);
        const float4 b = ld4(z + i * (int64_t)ld_z + c * 4);
        st4(table + r * W + c * 4, make_float4(a.x + tanhf(b.x), a.y + tanhf(b.y), a.z + tanhf(b.z), a.w + tanhf(b.w)));
    }
}

}  // namespace lstep

using namespace lstep;

extern "C" int lstep_segment_pe_time_sum(const float* pe, int32_t pe_dim, const float* time_w, const float* time_b, int32_t time_dim,
                                         const int64_t* seg_begin, const int64_t* seg_end, int64_t num_segments, const int32_t* ent_row, const float* ent_dt,
                                         const uint8_t* ent_valid, float* out, int32_t ld_out, void* stream) {
    if (num_segments < 0) return set_error(LSTEP_EINVAL, "lstep_segment_pe_time_sum: negative count");
    if (num_segments == 0) return LSTEP_OK;
    if (pe_dim <= 0 || (pe_dim & 3) || pe_dim > 4 * kMaxRowVec || time_dim <= 0 || (time_dim & 3) || time_dim > kMaxTimeDim)
        return set_error(LSTEP_EINVAL, "lstep_segment_pe_time_sum: unsupported widths P=%d D=%d", pe_dim, time_dim);
    if (!pe || !time_w || !time_b || !seg_begin || !seg_end || !ent_row || !ent_dt || !out) return set_error(LSTEP_EINVAL, "lstep_segment_pe_time_sum: NULL pointer");
    if (ld_out == 0) ld_out = pe_dim + time_dim;
    if (ld_out < pe_dim + time_dim || (ld_out & 3) || ld_out - (pe_dim + time_dim) > 4 * kWave)
        return set_error(LSTEP_EINVAL, "lstep_segment_pe_time_sum: bad output row stride %d", ld_out);
    const unsigned grid = (unsigned)((num_segments + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(segment_pe_time_sum_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, pe, (int)pe_dim, time_w, time_b,
                       (int)time_dim, seg_begin, seg_end, num_segments, ent_row, ent_dt, ent_valid, out, (int)ld_out);
    return check_launch("segment_pe_time_sum_kernel");
}

extern "C" int lstep_scatter_rows(float* table, int32_t width, const int64_t* ids, int64_t num_ids, const float* rows, void* stream) {
    if (num_ids < 0 || width <= 0 || (width & 3)) return set_error(LSTEP_EINVAL, "lstep_scatter_rows: bad sizes");
    if (num_ids == 0) return LSTEP_OK;
    if (!table || !ids || !rows) return set_error(LSTEP_EINVAL, "lstep_scatter_rows: NULL pointer");
    const unsigned grid = (unsigned)((num_ids + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, table, (int)width, ids, num_ids, rows);
    return check_launch("scatter_rows_kernel");
}

extern "C" int lstep_residual_tanh_rows(float* table, int32_t width, const int64_t* ids, int64_t num_ids, const float* z, int32_t ld_z,
                                        void* stream) {
    if (num_ids < 0 || width <= 0 || (width & 3)) return set_error(LSTEP_EINVAL, "lstep_residual_tanh_rows: bad sizes");
    if (num_ids == 0) return LSTEP_OK;
    if (!table || !ids || !z) return set_error(LSTEP_EINVAL, "lstep_residual_tanh_rows: NULL pointer");
    if (ld_z == 0) ld_z = width;
    if (ld_z < width || (ld_z & 3)) return set_error(LSTEP_EINVAL, "lstep_residual_tanh_rows: bad row stride %d", ld_z);
    const unsigned grid = (unsigned)((num_ids + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(residual_tanh_rows_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, table, (int)width, ids, num_ids, z, (int)ld_z);
    return check_launch("residual_tanh_rows_kernel");
}
