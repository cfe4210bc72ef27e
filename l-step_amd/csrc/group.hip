// "Group by key" on the device: the plumbing every batch needs three times
//   (1) cat[src, dst] of the batch -> sorted unique batch nodes + per-node message segments   (train:221-222, LSTEP.py:282-290)
//   (2) sampled neighbour ids of update_pe phase 2 -> touched rows + their message segments   (LSTEP.py:319-324)
//   (3) spliced-row hits of the gather backward -> gradient segments
// One stable radix sort (hipCUB) + head flags + one scan instead of ~50 small framework launches per use.
#include <stdlib.h>

#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include "lstep_common.h"

namespace lstep {

__global__ void iota_kernel(int32_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)i;
}

__global__ void head_flag_kernel(const int32_t* __restrict__ sk, int32_t* __restrict__ flag, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        flag[i] = (i > 0 && sk[i] != sk[i - 1]) ? 1 : 0;
}

// uniq[seg] = key of the segment's first entry; begin[seg] = its position; summary = {num_unique, first index with key >= limit}
__global__ void unique_kernel(const int32_t* __restrict__ sk, const int32_t* __restrict__ seg, int64_t n, int32_t limit,
                              int32_t* __restrict__ uniq, int32_t* __restrict__ summary) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool head = i == 0 || sk[i] != sk[i - 1];
        if (head) uniq[seg[i]] = sk[i];
        if (i == n - 1) summary[0] = seg[i] + 1;
        // boundary between keys < limit and keys >= limit (sorted): exactly one thread sees it
        const bool below = sk[i] < limit;
        if (below && (i == n - 1 || sk[i + 1] >= limit)) { summary[1] = (int32_t)(i + 1); summary[2] = seg[i] + 1; }
        if (i == 0 && !below) { summary[1] = 0; summary[2] = 0; }
    }
}

// out[i] = i < *count ? (int64) ids32[i] : 0 -- a capacity-sized id list whose dead tail is the padding node 0 (no history, no
// neighbours: every consumer that only READS per-id state treats it as a no-op)
__global__ void widen_ids_kernel(const int32_t* __restrict__ ids32, int64_t capacity, const int32_t* __restrict__ count, int64_t* __restrict__ out) {
    const int64_t n = *count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = i < n ? (int64_t)ids32[i] : 0;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- up to 4096 keys: the whole grouping in ONE workgroup (block radix sort, head flags, block scan, unique, summary).  The reference's
// own batch sizes (200 / 600 edges: 400 / 1200 endpoints) otherwise pay for ~12 dependent library launches of a few microseconds each at
// the very start of every iteration.  Same outputs as the multi-launch path (the sort is stable in both).
constexpr int kSmallThreads = 256;
template <int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void group_small_kernel(const int32_t* __restrict__ keys, int n, int bits, int32_t limit,
                                                              int32_t* __restrict__ sorted_keys, int32_t* __restrict__ order,
                                                              int32_t* __restrict__ seg, int32_t* __restrict__ uniq,
                                                              int32_t* __restrict__ summary) {
    using Sort = rocprim::block_radix_sort<uint32_t, THREADS, ITEMS, int32_t>;
    using Scan = rocprim::block_scan<int32_t, THREADS>;
    __shared__ union {
        typename Sort::storage_type sort;
        typename Scan::storage_type scan;
        uint32_t sk[THREADS * ITEMS + 2];       // the sorted keys, shifted by one: sk[i + 1] = key of entry i (written after the sort is done with its storage)
    } tmp;
    const int t = threadIdx.x;
    uint32_t k[ITEMS];
    int32_t v[ITEMS];
    // padding sorts last: one bit above the real keys when there is room (then the sort needs bits + 1 bit ranges, not 32)
    const uint32_t pad = bits < 31 ? (1u << bits) : 0xFFFFFFFFu;
    // Global memory is touched in STRIPED order only (consecutive lanes, consecutive words), staged through LDS (round 5): with the blocked
    // arrangement the sort wants -- thread t owns entries t * ITEMS .. -- every load / store instruction of a wave hit 64 different cache
    // lines, and at 24 items per thread those 96 instructions were most of the kernel's 105 us.
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = j * THREADS + t;
        tmp.sk[idx] = idx < n ? (uint32_t)keys[idx] : pad;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = t * ITEMS + j;                     // blocked arrangement: stability = input order
        k[j] = tmp.sk[idx];
        v[j] = idx;
    }
    __syncthreads();                                       // (the sort reuses the storage)
    Sort().sort(k, v, tmp.sort, 0u, bits < 31 ? (unsigned)(bits + 1) : 32u);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) tmp.sk[t * ITEMS + j + 1] = k[j];
    if (t == 0) { tmp.sk[0] = 0xFFFFFFFFu; tmp.sk[THREADS * ITEMS + 1] = 0xFFFFFFFFu; }
    __syncthreads();
    int32_t flag[ITEMS], local = 0;
    uint32_t next[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = t * ITEMS + j;
        flag[j] = (idx > 0 && idx < n && tmp.sk[idx] != k[j]) ? 1 : 0;     // sk[idx] = the key before entry idx
        next[j] = idx + 1 < n ? tmp.sk[idx + 2] : 0xFFFFFFFFu;             // the key after entry idx
        local += flag[j];
    }
    __syncthreads();                                       // (the scan reuses the storage)
    int32_t before = 0;
    Scan().exclusive_scan(local, before, 0, tmp.scan);
    int32_t run = before;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = t * ITEMS + j;
        run += flag[j];
        if (idx < n) {
            const int32_t key = (int32_t)k[j];
            if (idx == 0 || flag[j]) uniq[run] = key;
            if (idx == n - 1) summary[0] = run + 1;
            const bool below = key < limit;
            if (below && (idx == n - 1 || (int64_t)next[j] >= (int64_t)limit)) { summary[1] = idx + 1; summary[2] = run + 1; }
            if (idx == 0 && !below) { summary[1] = 0; summary[2] = 0; }
        }
    }
    // the three per-entry outputs, one after the other through the same LDS block: blocked in, striped out
    int32_t* stage = reinterpret_cast<int32_t*>(tmp.sk);
    auto flush = [&](int32_t* __restrict__ dst) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = j * THREADS + t;
            if (idx < n) dst[idx] = stage[idx];
        }
        __syncthreads();
    };
    __syncthreads();                                       // (the scan is done with the storage)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) stage[t * ITEMS + j] = (int32_t)k[j];
    flush(sorted_keys);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) stage[t * ITEMS + j] = v[j];
    flush(order);
    run = before;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        run += flag[j];
        stage[t * ITEMS + j] = run;
    }
    flush(seg);
}

// lstep_sort_live_bounded for up to 65536 keys and a capacity of up to 32768: compaction of the live (non-negative) keys, sentinel
// padding and the stable sort in ONE workgroup instead of ~9 dependent library launches (at the reference's batch sizes the gradient-hit
// lists are 12 000 - 36 000 slots long).  Same outputs as the multi-launch path.
constexpr int kLiveThreads = 1024;
constexpr int kLiveTile = 8;       // keys per thread and compaction round
template <int ITEMS>
__global__ __launch_bounds__(kLiveThreads) void sort_live_small_kernel(const int32_t* __restrict__ keys, int n, int bits, int32_t sentinel, int capacity,
                                                                       int32_t* __restrict__ sorted_keys, int32_t* __restrict__ order,
                                                                       int32_t* __restrict__ live_index, int32_t* __restrict__ count) {
    using Sort = rocprim::block_radix_sort<uint32_t, kLiveThreads, ITEMS, int32_t>;
    using Scan = rocprim::block_scan<int32_t, kLiveThreads>;
    __shared__ union {
        typename Sort::storage_type sort;
        typename Scan::storage_type scan;
    } tmp;
    __shared__ int32_t total_sh;
    const int t = threadIdx.x;
    int32_t running = 0;
    for (int base = 0; base < n; base += kLiveThreads * kLiveTile) {
        int32_t live[kLiveTile], mine = 0;
#pragma unroll
        for (int j = 0; j < kLiveTile; ++j) {
            const int idx = base + t * kLiveTile + j;
            live[j] = (idx < n && keys[idx] >= 0) ? 1 : 0;
            mine += live[j];
        }
        int32_t before = 0, tile_total = 0;
        Scan().exclusive_scan(mine, before, 0, tile_total, tmp.scan);
        int32_t pos = running + before;
#pragma unroll
        for (int j = 0; j < kLiveTile; ++j)
            if (live[j]) live_index[pos++] = base + t * kLiveTile + j;
        running += tile_total;
        __syncthreads();
    }
    if (t == 0) { count[0] = running; total_sh = running; }
    __threadfence_block();
    __syncthreads();
    const int total = total_sh;
    uint32_t k[ITEMS];
    int32_t v[ITEMS];
    const uint32_t beyond = bits < 31 ? ((1u << bits) - 1u) : 0x7FFFFFFFu;     // slots past `capacity`: largest key of the sorted bit range
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int i = t * ITEMS + j;
        if (i < capacity) {
            const bool is_live = i < total;
            const int32_t e = is_live ? live_index[i] : 0;
            k[j] = is_live ? (uint32_t)keys[e] : (uint32_t)sentinel;
            v[j] = e;
        } else {
            k[j] = beyond;
            v[j] = 0;
        }
    }
    Sort().sort(k, v, tmp.sort, 0u, (unsigned)bits);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int i = t * ITEMS + j;
        if (i < capacity) { sorted_keys[i] = (int32_t)k[j]; order[i] = v[j]; }
    }
}

// Stable LSD radix sort of (key, value) pairs on the low `bits` key bits.  The library's default switches to a merge sort below
// 1 M items; measured here (MI355X, 15-20 key bits): merge sort 8 launches / 36 us at 32 k items but 21 launches / 250 us at 330 k,
// Onesweep ~31 us per 6-8-bit pass whatever the size.  So: merge sort up to 64 k items, Onesweep above.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;

static hipError_t sort_pairs(void* temp, size_t& temp_bytes, const int32_t* keys_in, int32_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                             int64_t n, int bits, hipStream_t s) {
    // keys are non-negative int32: sorted as unsigned (rocprim's signed-key bit flip only touches the sign bit, outside `bits`)
    return rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, reinterpret_cast<const uint32_t*>(keys_in), reinterpret_cast<uint32_t*>(keys_out),
                                                 vals_in, vals_out, (size_t)n, 0u, (unsigned)bits, s);
}

struct NonNegative {   // select predicate over entry indices: keep i if keys[i] >= 0
    const int32_t* keys;
    __host__ __device__ bool operator()(const int32_t& i) const { return keys[i] >= 0; }
};

__global__ void take_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                            int32_t* __restrict__ out) {
    const int64_t n = *count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = src[idx[i]];
}

// live prefix + sentinel padding up to `capacity`: keys_out / idx_out are the (fixed-size) input of the sort
__global__ void take_pad_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int64_t capacity,
                                int32_t sentinel, int32_t* __restrict__ keys_out, int32_t* __restrict__ idx_out) {
    const int64_t n = *count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (int64_t)gridDim.x * blockDim.x) {
        const bool live = i < n;
        const int32_t e = live ? idx[i] : 0;
        keys_out[i] = live ? src[e] : sentinel;
        idx_out[i] = e;
    }
}

static size_t select_temp_bytes(int64_t n) {
    size_t a = 0;
    NonNegative pred{nullptr};
    (void)hipcub::DeviceSelect::If(nullptr, a, hipcub::CountingInputIterator<int32_t>(0), (int32_t*)nullptr, (int32_t*)nullptr, (int)n, pred,
                                   (hipStream_t)0);
    return a;
}

static size_t cub_temp_bytes(int64_t n, int bits) {
    size_t a = 0, b = 0;
    (void)sort_pairs(nullptr, a, nullptr, nullptr, nullptr, nullptr, n, bits, (hipStream_t)0);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, (hipStream_t)0);
    return a > b ? a : b;
}

}  // namespace lstep

using namespace lstep;

extern "C" int64_t lstep_group_by_key_workspace(int64_t n, int32_t key_bits) {
    if (n <= 0) return 256;
    return (int64_t)(2 * align256((size_t)n * 4) + align256(cub_temp_bytes(n, key_bits)) + 256);
}

extern "C" int lstep_group_by_key(const int32_t* keys, int64_t n, int32_t key_bits, int32_t limit, void* workspace, int64_t workspace_bytes,
                                  int32_t* sorted_keys, int32_t* order, int32_t* seg, int32_t* uniq, int32_t* summary, void* stream) {
    if (n < 0 || key_bits <= 0 || key_bits > 31) return set_error(LSTEP_EINVAL, "lstep_group_by_key: bad sizes");
    if (n >= ((int64_t)1 << 31)) return set_error(LSTEP_EINVAL, "lstep_group_by_key: more than 2^31 - 1 entries");
    if (!summary) return set_error(LSTEP_EINVAL, "lstep_group_by_key: NULL summary");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(summary, 0, 3 * sizeof(int32_t), s) != hipSuccess) return set_error(LSTEP_EHIP, "lstep_group_by_key: memset failed");
        return LSTEP_OK;
    }
    if (!keys || !workspace || !sorted_keys || !order || !seg || !uniq) return set_error(LSTEP_EINVAL, "lstep_group_by_key: NULL pointer");
    if (workspace_bytes < lstep_group_by_key_workspace(n, key_bits)) return set_error(LSTEP_EINVAL, "lstep_group_by_key: workspace too small");
    const char* no_small = getenv("LSTEP_GROUP_NO_SMALL");      // A/B and the parity test: read per call
    if (n <= 1024 * 24 && !(no_small && no_small[0] == '1')) {
#define LSTEP_GROUP_SMALL(TH, IT) hipLaunchKernelGGL((group_small_kernel<TH, IT>), dim3(1), dim3(TH), 0, s, keys, (int)n, key_bits, limit, sorted_keys, order, seg, uniq, summary)
        if (n <= kSmallThreads * 2) LSTEP_GROUP_SMALL(kSmallThreads, 2);
        else if (n <= kSmallThreads * 8) LSTEP_GROUP_SMALL(kSmallThreads, 8);
        else if (n <= kSmallThreads * 16) LSTEP_GROUP_SMALL(kSmallThreads, 16);
        else if (n <= 1024 * 8) LSTEP_GROUP_SMALL(1024, 8);
        else if (n <= 1024 * 16) LSTEP_GROUP_SMALL(1024, 16);
        else LSTEP_GROUP_SMALL(1024, 24);      // (32 keys per thread spill: 24 still fit the 128-register budget of a 1024-thread workgroup)
#undef LSTEP_GROUP_SMALL
        return check_launch("group_small_kernel");
    }
    char* ws = (char*)workspace;
    int32_t* idx = (int32_t*)ws;
    int32_t* flag = (int32_t*)(ws + align256((size_t)n * 4));
    void* temp = ws + 2 * align256((size_t)n * 4);
    size_t temp_bytes = cub_temp_bytes(n, key_bits);
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(256), 0, s, idx, n);
    if (sort_pairs(temp, temp_bytes, keys, sorted_keys, idx, order, n, key_bits, s) != hipSuccess)
        return set_error(LSTEP_EHIP, "lstep_group_by_key: radix sort failed");
    hipLaunchKernelGGL(head_flag_kernel, dim3(grid), dim3(256), 0, s, sorted_keys, flag, n);
    temp_bytes = cub_temp_bytes(n, key_bits);
    if (hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, flag, seg, (int)n, s) != hipSuccess)
        return set_error(LSTEP_EHIP, "lstep_group_by_key: scan failed");
    hipLaunchKernelGGL(unique_kernel, dim3(grid), dim3(256), 0, s, sorted_keys, seg, n, limit, uniq, summary);
    return check_launch("lstep_group_by_key");
}

extern "C" int lstep_widen_ids(const int32_t* ids32, int64_t capacity, const int32_t* count, int64_t* out, void* stream) {
    if (capacity < 0) return set_error(LSTEP_EINVAL, "lstep_widen_ids: negative capacity");
    if (capacity == 0) return LSTEP_OK;
    if (!ids32 || !count || !out) return set_error(LSTEP_EINVAL, "lstep_widen_ids: NULL pointer");
    const unsigned grid = (unsigned)((capacity + 255) / 256 < 1024 ? (capacity + 255) / 256 : 1024);
    hipLaunchKernelGGL(widen_ids_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids32, capacity, count, out);
    return check_launch("widen_ids_kernel");
}

// ---- "sort the live entries": keys < 0 are dropped BEFORE the sort.  The gradient hits of the gather backward are ~95 % dead
// (only neighbours that are batch nodes themselves carry gradient), and the library sorts ~1 M pairs with ~20 merge passes.
extern "C" int64_t lstep_sort_live_workspace(int64_t n, int32_t key_bits) {
    if (n <= 0) return 256;
    const size_t temp = cub_temp_bytes(n, key_bits) > select_temp_bytes(n) ? cub_temp_bytes(n, key_bits) : select_temp_bytes(n);
    return (int64_t)(2 * align256((size_t)n * 4) + align256(temp) + 512);
}

extern "C" int lstep_sort_live(const int32_t* keys, int64_t n, int32_t key_bits, void* workspace, int64_t workspace_bytes, int32_t* sorted_keys,
                               int32_t* order, int64_t* num_live, void* stream) {
    if (n < 0 || key_bits <= 0 || key_bits > 31 || !num_live) return set_error(LSTEP_EINVAL, "lstep_sort_live: bad arguments");
    if (n >= ((int64_t)1 << 31)) return set_error(LSTEP_EINVAL, "lstep_sort_live: more than 2^31 - 1 entries");
    *num_live = 0;
    if (n == 0) return LSTEP_OK;
    if (!keys || !workspace || !sorted_keys || !order) return set_error(LSTEP_EINVAL, "lstep_sort_live: NULL pointer");
    if (workspace_bytes < lstep_sort_live_workspace(n, key_bits)) return set_error(LSTEP_EINVAL, "lstep_sort_live: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int32_t* idx = (int32_t*)ws;                                   // live entry indices, ascending
    int32_t* live_keys = (int32_t*)(ws + align256((size_t)n * 4));
    int32_t* count = (int32_t*)(ws + 2 * align256((size_t)n * 4));
    void* temp = ws + 2 * align256((size_t)n * 4) + 256;
    size_t temp_bytes = select_temp_bytes(n);
    NonNegative pred{keys};
    if (hipcub::DeviceSelect::If(temp, temp_bytes, hipcub::CountingInputIterator<int32_t>(0), idx, count, (int)n, pred, s) != hipSuccess)
        return set_error(LSTEP_EHIP, "lstep_sort_live: select failed");
    const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(take_kernel, dim3(grid), dim3(256), 0, s, keys, idx, count, live_keys);
    int32_t host_count = 0;   // the one host round trip: the sort (and the caller's segment sum) are sized by it
    if (hipMemcpyAsync(&host_count, count, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return set_error(LSTEP_EHIP, "lstep_sort_live: count read-back failed");
    *num_live = host_count;
    if (host_count == 0) return LSTEP_OK;
    temp_bytes = cub_temp_bytes(host_count, key_bits);
    if (sort_pairs(temp, temp_bytes, live_keys, sorted_keys, idx, order, host_count, key_bits, s) != hipSuccess)
        return set_error(LSTEP_EHIP, "lstep_sort_live: radix sort failed");
    return check_launch("lstep_sort_live");
}

// lstep_sort_live without the host round trip: the sort runs on a FIXED number of items (`capacity`, chosen by the caller from earlier
// batches): the live keys, padded with `sentinel` (a value above every live key that still fits key_bits).  *count (device) = number of
// live entries, live_index[0 .. *count) = their indices in ascending order.  If *count > capacity only the first `capacity` live entries are
// in the sorted output: the caller handles live_index[capacity .. *count) separately (lstep_scatter_add_overflow).
extern "C" int64_t lstep_sort_live_bounded_workspace(int64_t n, int64_t capacity, int32_t key_bits) {
    if (n <= 0 || capacity <= 0) return 0;
    const size_t a = select_temp_bytes(n), b = cub_temp_bytes(capacity, key_bits);
    return (int64_t)(2 * align256((size_t)capacity * 4) + 256 + align256(a > b ? a : b));
}

extern "C" int lstep_sort_live_bounded(const int32_t* keys, int64_t n, int32_t key_bits, int32_t sentinel, int64_t capacity, void* workspace,
                                       int64_t workspace_bytes, int32_t* sorted_keys, int32_t* order, int32_t* live_index, int32_t* count,
                                       void* stream) {
    if (n <= 0 || capacity <= 0 || key_bits <= 0 || key_bits > 31 || sentinel < 0 || (key_bits < 31 && sentinel >= (1 << key_bits)))
        return set_error(LSTEP_EINVAL, "lstep_sort_live_bounded: bad arguments");
    if (n >= ((int64_t)1 << 31) || capacity >= ((int64_t)1 << 31)) return set_error(LSTEP_EINVAL, "lstep_sort_live_bounded: more than 2^31 - 1 entries");
    if (!keys || !workspace || !sorted_keys || !order || !live_index || !count) return set_error(LSTEP_EINVAL, "lstep_sort_live_bounded: NULL pointer");
    if (workspace_bytes < lstep_sort_live_bounded_workspace(n, capacity, key_bits))
        return set_error(LSTEP_EINVAL, "lstep_sort_live_bounded: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const char* no_small = getenv("LSTEP_GROUP_NO_SMALL");
    if (n <= 65536 && capacity <= kLiveThreads * 24 && !(no_small && no_small[0] == '1')) {
#define LSTEP_LIVE_SMALL(IT) hipLaunchKernelGGL((sort_live_small_kernel<IT>), dim3(1), dim3(kLiveThreads), 0, s, keys, (int)n, key_bits, sentinel, (int)capacity, sorted_keys, order, live_index, count)
        if (capacity <= kLiveThreads * 8) LSTEP_LIVE_SMALL(8);
        else if (capacity <= kLiveThreads * 16) LSTEP_LIVE_SMALL(16);
        else LSTEP_LIVE_SMALL(24);
#undef LSTEP_LIVE_SMALL
        return check_launch("sort_live_small_kernel");
    }
    char* ws = (char*)workspace;
    int32_t* live_keys = (int32_t*)ws;
    int32_t* live_idx = (int32_t*)(ws + align256((size_t)capacity * 4));
    void* temp = ws + 2 * align256((size_t)capacity * 4) + 256;
    size_t temp_bytes = select_temp_bytes(n);
    NonNegative pred{keys};
    if (hipcub::DeviceSelect::If(temp, temp_bytes, hipcub::CountingInputIterator<int32_t>(0), live_index, count, (int)n, pred, s) != hipSuccess)
        return set_error(LSTEP_EHIP, "lstep_sort_live_bounded: select failed");
    const unsigned grid = (unsigned)((capacity + 255) / 256 < 2048 ? (capacity + 255) / 256 : 2048);
    hipLaunchKernelGGL(take_pad_kernel, dim3(grid), dim3(256), 0, s, keys, live_index, count, capacity, sentinel, live_keys, live_idx);
    temp_bytes = cub_temp_bytes(capacity, key_bits);
    if (sort_pairs(temp, temp_bytes, live_keys, sorted_keys, live_idx, order, capacity, key_bits, s) != hipSuccess)
        return set_error(LSTEP_EHIP, "lstep_sort_live_bounded: radix sort failed");
    return check_launch("lstep_sort_live_bounded");
}
