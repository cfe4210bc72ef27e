// Segmented row sums: U1 / U2 message accumulation of update_pe without the dense [N+1, P+D] scatter target
//   (reference models/LSTEP.py:282-290: per batch edge, both directions; :319-322: per sampled neighbour), and the
//   sort-based PE-gradient reduction of the gather backward.  Entries are pre-grouped by destination segment; the
//   entry array is cut into fixed 128-entry chunks (one wave each) so hub segments cannot serialise the kernel.
#include "lstep_common.h"

namespace lstep {

constexpr int kSegInFlight = 8;
constexpr int kChunk = 128;  // entries per wave: bounds the work of one wave whatever the segment-length distribution is

// flush one run of a segment: plain store when this chunk holds the whole segment, float atomics (contiguous dwords
// per wave-instruction) when the segment is split over several chunks (long segments = hub nodes)
__device__ __forceinline__ void flush_run(float* __restrict__ o, const float4& acc, float t0, float t1, int W, int D, bool whole,
                                          int lane) {
    const bool wa = lane < (W >> 2);
    if (whole) {
        if (wa) st4(o + lane * 4, acc);
        if (lane < D) o[W + lane] = t0;
        if (lane + kWave < D) o[W + lane + kWave] = t1;
    } else {
        if (wa) {
            atomicAdd(o + lane * 4 + 0, acc.x);
            atomicAdd(o + lane * 4 + 1, acc.y);
            atomicAdd(o + lane * 4 + 2, acc.z);
            atomicAdd(o + lane * 4 + 3, acc.w);
        }
        if (lane < D) atomicAdd(o + W + lane, t0);
        if (lane + kWave < D) atomicAdd(o + W + lane + kWave, t1);
    }
}

// Entries are sorted by segment; wave c owns entries [c * kChunk, (c + 1) * kChunk).  Inside the chunk the wave walks the
// runs of equal segment id, summing table rows (8 in flight) and time features, and flushes each run.
__global__ __launch_bounds__(kBlock) void segment_rows_sum_kernel(const float* __restrict__ table, int W, int ld_table,
                                                                   const float* __restrict__ tw, const float* __restrict__ tb, int D,
                                                                   const int64_t* __restrict__ seg_begin, const int64_t* __restrict__ seg_end,
                                                                   const int32_t* __restrict__ ent_seg, const int32_t* __restrict__ ent_row,
                                                                   const float* __restrict__ ent_dt, int64_t num_entries,
                                                                   float* __restrict__ out, int ld_out) {
    const int lane = lane_id();
    const int64_t chunk = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t e0 = chunk * kChunk;
    if (e0 >= num_entries) return;
    const int64_t e1 = (e0 + kChunk < num_entries) ? e0 + kChunk : num_entries;
    const bool wa = lane < (W >> 2);
    const float w0 = lane < D ? tw[lane] : 0.f, b0 = lane < D ? tb[lane] : 0.f;
    const float w1 = lane + kWave < D ? tw[lane + kWave] : 0.f, b1 = lane + kWave < D ? tb[lane + kWave] : 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float t0 = 0.f, t1 = 0.f;
    int cur = -1;  // segment of the open run
    for (int64_t c0 = e0; c0 < e1; c0 += kWave) {
        const int m = (int)((e1 - c0) < kWave ? (e1 - c0) : kWave);
        int sg = 0, r = 0;
        float dt = 0.f;
        if (lane < m) {
            sg = ent_seg[c0 + lane];
            r = ent_row[c0 + lane];
            if (D > 0) dt = ent_dt[c0 + lane];
        }
        settle(sg ^ r ^ __float_as_int(dt));
        int j = 0;
        while (j < m) {
            const int sj = bcast_i32(sg, j);
            if (sj != cur) {
                if (cur >= 0) {
                    const bool whole = seg_begin[cur] >= e0 && seg_end[cur] <= e1;
                    flush_run(out + (int64_t)cur * ld_out, acc, t0, t1, W, D, whole, lane);
                }
                cur = sj;
                acc = make_float4(0.f, 0.f, 0.f, 0.f);
                t0 = t1 = 0.f;
            }
            // length of this run inside the 64-entry window, capped at the rows-in-flight group size
            int n = 1;
            while (j + n < m && n < kSegInFlight && bcast_i32(sg, j + n) == sj) ++n;
            float4 x[kSegInFlight];
            if (wa) {
#pragma unroll
                for (int u = 0; u < kSegInFlight; ++u) {
                    const int64_t rj = bcast_i32(r, (u < n) ? (j + u) : j);   // tail slots re-read the first row, weight 0
                    x[u] = ld4(table + rj * ld_table + lane * 4);
                }
            }
            if (D > 0) {
                for (int u = 0; u < n; ++u) {
                    const float dj = bcast_f32(dt, j + u);
                    if (lane < D) t0 += time_feat(dj, w0, b0);
                    if (lane + kWave < D) t1 += time_feat(dj, w1, b1);
                }
            }
            if (wa) {
#pragma unroll
                for (int u = 0; u < kSegInFlight; ++u) fma4(acc, (u < n) ? 1.f : 0.f, x[u]);
            }
            j += n;
        }
    }
    if (cur >= 0) {
        const bool whole = seg_begin[cur] >= e0 && seg_end[cur] <= e1;
        flush_run(out + (int64_t)cur * ld_out, acc, t0, t1, W, D, whole, lane);
    }
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

extern "C" int lstep_segment_rows_sum(const float* table, int32_t width, int32_t ld_table, const float* time_w, const float* time_b,
                                      int32_t time_dim, const int64_t* seg_begin, const int64_t* seg_end, int64_t num_segments,
                                      const int32_t* ent_seg, const int32_t* ent_row, const float* ent_dt, int64_t num_entries, float* out,
                                      int32_t ld_out, void* stream) {
    if (num_segments < 0 || num_entries < 0) return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum: negative count");
    if (num_segments == 0 || num_entries == 0) return LSTEP_OK;
    if (ld_table == 0) ld_table = width;
    if (ld_out == 0) ld_out = width + time_dim;
    if (width <= 0 || (width & 3) || width > 4 * kMaxRowVec || time_dim < 0 || (time_dim & 3) || time_dim > kMaxTimeDim || ld_table < width ||
        (ld_table & 3) || ld_out < width + time_dim || (ld_out & 3))
        return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum: unsupported widths W=%d D=%d ld_table=%d ld_out=%d", width, time_dim, ld_table, ld_out);
    if (!table || !seg_begin || !seg_end || !ent_seg || !ent_row || !out || (time_dim > 0 && (!time_w || !time_b || !ent_dt)))
        return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum: NULL pointer");
    const int64_t chunks = (num_entries + kChunk - 1) / kChunk;
    const unsigned grid = (unsigned)((chunks + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(segment_rows_sum_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, table, (int)width, (int)ld_table, time_w,
                       time_b, (int)time_dim, seg_begin, seg_end, ent_seg, ent_row, ent_dt, num_entries, out, (int)ld_out);
    return check_launch("segment_rows_sum_kernel");
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
