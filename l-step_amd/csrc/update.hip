// update_pe's message lists (models/LSTEP.py:277-290, 305-324) built on the device in one kernel each, instead of a dozen framework
// index / cat / cast launches per phase (each of which queues behind the backward pass's big kernels on the update stream).
//   phase 1: every batch edge sends cat[pe[other endpoint], time_feat(now - t)] to both endpoints; the entries are the positions of
//            cat[src, dst] grouped by receiving endpoint (lstep_group_by_key's `order`).
//   phase 2: every sampled neighbour slot (row r of the batch-node set, slot j) sends cat[pe[bn[r]], time_feat(now - nt[r, j])] to the
//            neighbour; slots with neighbour 0 are padding (they feed row 0, lstep_padding_rows_sum).
#include "lstep_common.h"

namespace lstep {

// ent_row[e] = the OTHER endpoint of entry order[e] of cat[src, dst];  ent_dt[e] = float(double(now32) - t)   (LSTEP.py:277: float32 now)
__global__ void update_entries_p1_kernel(const int32_t* __restrict__ order, int64_t n2, const int64_t* __restrict__ src,
                                         const int64_t* __restrict__ dst, const double* __restrict__ t, const float* __restrict__ now32,
                                         int64_t b, int32_t* __restrict__ ent_row, float* __restrict__ ent_dt) {
    const double now = (double)now32[0];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = order[e];
        const bool first = o < b;                    // entry of src: the message comes from dst
        const int64_t i = first ? o : o - b;
        ent_row[e] = (int32_t)(first ? dst[i] : src[i]);
        ent_dt[e] = (float)(now - t[i]);
    }
}

// keys[i] = nbr[i] if it is a real neighbour owned by this shard, else `sentinel` (sorts last, dropped by lstep_group_by_key's limit)
__global__ void update_keys_p2_kernel(const int64_t* __restrict__ nbr, int64_t n, int32_t sentinel, int32_t world, int32_t rank,
                                      int32_t* __restrict__ keys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = nbr[i];
        const bool real = k != 0 && (world <= 1 || (k % world) == rank);
        keys[i] = real ? (int32_t)k : sentinel;
    }
}

// for the n_real leading (grouped) slots: source row, time delta (float32 - float32, LSTEP.py:314), segment (+ shift when row 0 takes
// segment 0); and the touched-row list: [0 if shift] + uniq[:nseg]
__global__ void update_entries_p2_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ seg, int64_t n_real,
                                         const int64_t* __restrict__ bn, const float* __restrict__ nt, const float* __restrict__ now32, int32_t k,
                                         int32_t shift, const int32_t* __restrict__ uniq, int64_t nseg, int32_t* __restrict__ ent_row,
                                         float* __restrict__ ent_dt, int32_t* __restrict__ ent_seg, int64_t* __restrict__ touched) {
    const float now = now32[0];
    const int64_t total = n_real > nseg ? n_real : nseg;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < n_real) {
            const int64_t o = order[e];
            ent_row[e] = (int32_t)bn[o / k];
            ent_dt[e] = now - nt[o];
            ent_seg[e] = seg[e] + shift;
        }
        if (e < nseg) touched[shift + e] = uniq[e];
        if (e == 0 && shift) touched[0] = 0;
    }
}

// The same with every count on the device (no host round trip between the grouping and the message sums):
//   summary = lstep_group_by_key's {., n_real, nseg} of the capacity-sized key list, live_rows = number of batch nodes (the key list has
//   capacity_rows * k slots; slots of rows >= *live_rows are padding by construction).  Segment 0 is ALWAYS reserved for row 0:
//   ent_seg = seg + 1, touched = [0, uniq[0 .. nseg), 0 ...] up to touched_capacity, counts_out = {nseg, row-0 flag}: row 0 takes part
//   iff some slot of a LIVE row is padding (models/LSTEP.py:324: 0 is in unique(neighbour ids)), i.e. live_rows * k > n_real.
__global__ void update_entries_p2_dev_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ seg, const int32_t* __restrict__ summary,
                                             const int32_t* __restrict__ live_rows, int64_t capacity, int64_t touched_capacity,
                                             const int64_t* __restrict__ bn, const float* __restrict__ nt, const float* __restrict__ now32, int32_t k,
                                             const int32_t* __restrict__ uniq, int32_t* __restrict__ ent_row, float* __restrict__ ent_dt,
                                             int32_t* __restrict__ ent_seg, int64_t* __restrict__ touched, int32_t* __restrict__ counts_out) {
    const float now = now32[0];
    const int64_t n_real = summary[1], nseg = summary[2];
    const int64_t total = capacity > touched_capacity ? capacity : touched_capacity;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < n_real) {
            const int64_t o = order[e];
            ent_row[e] = bn ? (int32_t)bn[o / k] : (int32_t)(o / k);   // (no id list: the row's position in it, for a table laid out in list order)
            ent_dt[e] = now - nt[o];
            ent_seg[e] = seg[e] + 1;
        }
        if (e + 1 < touched_capacity) touched[e + 1] = e < nseg ? (int64_t)uniq[e] : 0;
        if (e == 0) {
            touched[0] = 0;
            counts_out[0] = (int32_t)nseg;
            counts_out[1] = ((int64_t)live_rows[0] * k > n_real) ? 1 : 0;
        }
    }
}

// One launch for what an engine iteration derives from its batch before anything else runs: the gather rows cat[src, dst, neg] with their
// times cat[t, t, t] (train_LSTEP_link_prediction.py:233-251: three combining_pe_raw_feat calls on the same times), the int32 grouping keys
// cat[src, dst] (train:221-222) and float32(max t) (models/LSTEP.py:277: torch.Tensor([current_time]) rounds to float32 first).  The last
// workgroup reduces the maximum; the others copy.
__global__ __launch_bounds__(256) void batch_prepare_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, const int64_t* __restrict__ neg,
                                                            const double* __restrict__ t, int64_t b, int64_t* __restrict__ ids3, double* __restrict__ t3,
                                                            int32_t* __restrict__ keys, float* __restrict__ now32) {
    if (blockIdx.x == gridDim.x - 1) {
        __shared__ double part[256];
        double m = -1.7976931348623157e308;
        // (eight independent loads per round: one load per round is b / 256 dependent memory latencies -- 25 us at b = 16384, at the head of the step)
        for (int64_t i0 = threadIdx.x; i0 < b; i0 += 8 * 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i0 + u * 256 < b ? t[i0 + u * 256] : m;
#pragma unroll
            for (int u = 0; u < 8; ++u) m = v[u] > m ? v[u] : m;
        }
        part[threadIdx.x] = m;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) part[threadIdx.x] = part[threadIdx.x + off] > part[threadIdx.x] ? part[threadIdx.x + off] : part[threadIdx.x];
            __syncthreads();
        }
        if (threadIdx.x == 0) now32[0] = (float)part[0];
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < b; i += (int64_t)(gridDim.x - 1) * 256) {
        const int64_t s = src[i], d = dst[i];
        const double ti = t[i];
        ids3[i] = s; ids3[b + i] = d;
        t3[i] = ti; t3[b + i] = ti;
        if (neg) { ids3[2 * b + i] = neg[i]; t3[2 * b + i] = ti; }
        keys[i] = (int32_t)s; keys[b + i] = (int32_t)d;
    }
}

// row[:width] = sum over the blocks of partial[blk, :width], row[width:row_width] = 0: the padding row's aggregate of update_pe phase 2
// (lstep_padding_rows_sum's per-block sums; its time part is zero, models/LSTEP.py:316).  One workgroup of 16 waves: wave w adds its
// contiguous share of the blocks in block order, eight loads in flight, lane = one float4 of the row; the 16 wave sums meet in LDS and are added
// in wave order -- a fixed order, so the result is a function of the inputs alone.  (One thread per column walking all the blocks is
// `blocks` dependent memory latencies: 196 us for the 512 partial rows of a 32 768-row update, on update_pe's chain.)
constexpr int kFinishWaves = 16;
__global__ __launch_bounds__(kFinishWaves * kWave) void padding_rows_finish_kernel(const float* __restrict__ partial, int64_t blocks, int width,
                                                                                   float* __restrict__ row, int row_width) {
    __shared__ float4 sh[kFinishWaves][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int64_t per = (blocks + kFinishWaves - 1) / kFinishWaves;
    const int64_t k0 = wave * per, k1 = k0 + per < blocks ? k0 + per : blocks;
    for (int c0 = 0; c0 < row_width; c0 += 4 * kWave) {       // (rows wider than 256 floats: another pass)
        const int c = c0 + 4 * lane;
        const bool on = c + 4 <= width;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) {
            for (int64_t k = k0; k < k1; k += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = k + u < k1 ? ld4(partial + (k + u) * width + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
        sh[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && c < row_width) {
            float4 t = sh[0][lane];
#pragma unroll
            for (int w = 1; w < kFinishWaves; ++w) { const float4 v = sh[w][lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            const float out[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < row_width) row[c + e] = c + e < width ? out[e] : 0.f;
        }
        __syncthreads();
    }
}

// ---- owner-sharded PE table (lstep_amd/parallel.py, form "pull"): the request lists of one gather.
// keys[i] = owner(id) * num_rows + id for the ids this rank does NOT own among {neighbour slots, the rows themselves, the padding row 0},
// `sentinel` (= world * num_rows: dropped by the grouping) for the ones it owns: sorting the keys groups the distinct ids by owner.
__global__ void pull_keys_kernel(const int64_t* __restrict__ nbr, int64_t n_nbr, const int64_t* __restrict__ ids, int64_t n_ids, int32_t world,
                                 int32_t rank, int64_t num_rows, int32_t* __restrict__ keys) {
    const int64_t total = n_nbr + n_ids + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = i < n_nbr ? nbr[i] : (i < n_nbr + n_ids ? ids[i - n_nbr] : 0);
        const int64_t owner = id % world;
        keys[i] = owner == rank ? (int32_t)(world * num_rows) : (int32_t)(owner * num_rows + id);
    }
}

// From the grouped keys (uniq sorted by (owner, id), summary[2] of them below the sentinel): count[p] = ids requested from owner p and the
// fixed-capacity request blocks req[p, :capacity] = those ids (global), -1 beyond.  One workgroup per owner; the block boundaries are found by
// binary search.
__global__ __launch_bounds__(256) void pull_blocks_kernel(const int32_t* __restrict__ uniq, const int32_t* __restrict__ summary, int32_t world,
                                                          int64_t num_rows, int64_t capacity, int32_t* __restrict__ req, int32_t* __restrict__ count) {
    const int p = blockIdx.x;
    const int64_t n = summary[2];
    auto lower = [&](int64_t key) {      // first position whose key >= `key`
        int64_t lo = 0, hi = n;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)uniq[mid] < key) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const int64_t b = lower((int64_t)p * num_rows), e = lower((int64_t)(p + 1) * num_rows);
    if (threadIdx.x == 0) count[p] = (int32_t)(e - b);
    for (int64_t i = threadIdx.x; i < capacity; i += 256)
        req[(int64_t)p * capacity + i] = (b + i < e) ? (int32_t)((int64_t)uniq[b + i] - (int64_t)p * num_rows) : -1;
}

static unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace lstep

using namespace lstep;

extern "C" int lstep_update_entries_p1(const int32_t* order, int64_t num_entries, const int64_t* src, const int64_t* dst, const double* times,
                                       const float* now32, int64_t batch, int32_t* ent_row, float* ent_dt, void* stream) {
    if (num_entries < 0 || batch < 0 || num_entries > 2 * batch) return set_error(LSTEP_EINVAL, "lstep_update_entries_p1: bad sizes");
    if (num_entries == 0) return LSTEP_OK;
    if (!order || !src || !dst || !times || !now32 || !ent_row || !ent_dt) return set_error(LSTEP_EINVAL, "lstep_update_entries_p1: NULL pointer");
    hipLaunchKernelGGL(update_entries_p1_kernel, dim3(grid_for(num_entries)), dim3(256), 0, (hipStream_t)stream, order, num_entries, src, dst, times,
                       now32, batch, ent_row, ent_dt);
    return check_launch("update_entries_p1_kernel");
}

extern "C" int lstep_update_keys_p2(const int64_t* nbr, int64_t n, int32_t sentinel, int32_t world, int32_t rank, int32_t* keys, void* stream) {
    if (n < 0 || world < 1 || rank < 0 || rank >= world) return set_error(LSTEP_EINVAL, "lstep_update_keys_p2: bad sizes");
    if (n == 0) return LSTEP_OK;
    if (!nbr || !keys) return set_error(LSTEP_EINVAL, "lstep_update_keys_p2: NULL pointer");
    hipLaunchKernelGGL(update_keys_p2_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, nbr, n, sentinel, world, rank, keys);
    return check_launch("update_keys_p2_kernel");
}

extern "C" int lstep_update_entries_p2(const int32_t* order, const int32_t* seg, int64_t n_real, const int64_t* bn, const float* nt,
                                       const float* now32, int32_t num_neighbors, int32_t shift, const int32_t* uniq, int64_t nseg, int32_t* ent_row,
                                       float* ent_dt, int32_t* ent_seg, int64_t* touched, void* stream) {
    if (n_real < 0 || nseg < 0 || num_neighbors <= 0 || shift < 0 || shift > 1) return set_error(LSTEP_EINVAL, "lstep_update_entries_p2: bad sizes");
    if (n_real == 0 && nseg == 0 && !shift) return LSTEP_OK;
    if (!now32 || !touched || (n_real > 0 && (!order || !seg || !bn || !nt || !ent_row || !ent_dt || !ent_seg)) || (nseg > 0 && !uniq))
        return set_error(LSTEP_EINVAL, "lstep_update_entries_p2: NULL pointer");
    const int64_t total = (n_real > nseg ? n_real : nseg) > 0 ? (n_real > nseg ? n_real : nseg) : 1;
    hipLaunchKernelGGL(update_entries_p2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, order, seg, n_real, bn, nt, now32,
                       num_neighbors, shift, uniq, nseg, ent_row, ent_dt, ent_seg, touched);
    return check_launch("update_entries_p2_kernel");
}

extern "C" int lstep_update_entries_p2_dev(const int32_t* order, const int32_t* seg, const int32_t* summary, const int32_t* live_rows,
                                           int64_t capacity, int64_t touched_capacity, const int64_t* bn, const float* nt, const float* now32,
                                           int32_t num_neighbors, const int32_t* uniq, int32_t* ent_row, float* ent_dt, int32_t* ent_seg,
                                           int64_t* touched, int32_t* counts_out, void* stream) {
    if (capacity <= 0 || touched_capacity <= 0 || num_neighbors <= 0) return set_error(LSTEP_EINVAL, "lstep_update_entries_p2_dev: bad sizes");
    if (!order || !seg || !summary || !live_rows || !nt || !now32 || !uniq || !ent_row || !ent_dt || !ent_seg || !touched || !counts_out)
        return set_error(LSTEP_EINVAL, "lstep_update_entries_p2_dev: NULL pointer");
    const int64_t total = capacity > touched_capacity ? capacity : touched_capacity;
    hipLaunchKernelGGL(update_entries_p2_dev_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, order, seg, summary, live_rows,
                       capacity, touched_capacity, bn, nt, now32, num_neighbors, uniq, ent_row, ent_dt, ent_seg, touched, counts_out);
    return check_launch("update_entries_p2_dev_kernel");
}


extern "C" int lstep_batch_prepare(const int64_t* src, const int64_t* dst, const int64_t* neg, const double* times, int64_t batch, int64_t* ids3,
                                   double* t3, int32_t* keys, float* now32, void* stream) {
    if (batch <= 0) return set_error(LSTEP_EINVAL, "lstep_batch_prepare: empty batch");
    if (!src || !dst || !times || !ids3 || !t3 || !keys || !now32) return set_error(LSTEP_EINVAL, "lstep_batch_prepare: NULL pointer");
    const unsigned copy_blocks = (unsigned)((batch + 255) / 256 < 1024 ? (batch + 255) / 256 : 1024);
    hipLaunchKernelGGL(batch_prepare_kernel, dim3(copy_blocks + 1), dim3(256), 0, (hipStream_t)stream, src, dst, neg, times, batch, ids3, t3, keys, now32);
    return check_launch("batch_prepare_kernel");
}

extern "C" int lstep_padding_rows_finish(const float* partial, int64_t blocks, int32_t width, float* row, int32_t row_width, void* stream) {
    if (blocks < 0 || width <= 0 || row_width < width) return set_error(LSTEP_EINVAL, "lstep_padding_rows_finish: bad sizes");
    if (!row || (blocks > 0 && !partial)) return set_error(LSTEP_EINVAL, "lstep_padding_rows_finish: NULL pointer");
    if (width & 3) return set_error(LSTEP_EINVAL, "lstep_padding_rows_finish: width must be a multiple of 4");
    hipLaunchKernelGGL(padding_rows_finish_kernel, dim3(1), dim3(kFinishWaves * kWave), 0, (hipStream_t)stream, partial, blocks, (int)width, row,
                       (int)row_width);
    return check_launch("padding_rows_finish_kernel");
}


extern "C" int lstep_pull_keys(const int64_t* nbr, int64_t n_nbr, const int64_t* ids, int64_t n_ids, int32_t world, int32_t rank, int64_t num_rows,
                               int32_t* keys, void* stream) {
    if (n_nbr < 0 || n_ids < 0 || world < 1 || rank < 0 || rank >= world || num_rows <= 0 || (int64_t)world * num_rows >= ((int64_t)1 << 31))
        return set_error(LSTEP_EINVAL, "lstep_pull_keys: bad sizes (world * num_rows must fit int32)");
    if (!keys || (n_nbr > 0 && !nbr) || (n_ids > 0 && !ids)) return set_error(LSTEP_EINVAL, "lstep_pull_keys: NULL pointer");
    hipLaunchKernelGGL(pull_keys_kernel, dim3(grid_for(n_nbr + n_ids + 1)), dim3(256), 0, (hipStream_t)stream, nbr, n_nbr, ids, n_ids, world, rank, num_rows,
                       keys);
    return check_launch("pull_keys_kernel");
}

extern "C" int lstep_pull_blocks(const int32_t* uniq, const int32_t* summary, int32_t world, int64_t num_rows, int64_t capacity, int32_t* req,
                                 int32_t* count, void* stream) {
    if (world < 1 || num_rows <= 0 || capacity <= 0) return set_error(LSTEP_EINVAL, "lstep_pull_blocks: bad sizes");
    if (!uniq || !summary || !req || !count) return set_error(LSTEP_EINVAL, "lstep_pull_blocks: NULL pointer");
    hipLaunchKernelGGL(pull_blocks_kernel, dim3((unsigned)world), dim3(256), 0, (hipStream_t)stream, uniq, summary, world, num_rows, capacity, req, count);
    return check_launch("pull_blocks_kernel");
}
