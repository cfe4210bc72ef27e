// Owner-sharded execution across GPUs (lstep_amd/parallel.py; node n is owned by rank n % W) with every data-dependent size left on the
// device, so that a rank's iteration is a fixed launch sequence with fixed-capacity collectives (and can be replayed as a HIP graph):
//   lstep_owner_partition    the sorted batch-node list (capacity-sized, live count on the device) split by owner into W blocks of C slots
//   lstep_scatter_owner_rows the all-gathered [W, C, P] row blocks written into the PE table (and numbered in slot_of)
//   lstep_rows_by_id         gather / scatter of table rows through an int32 id list with holes (-1): both ends of the row pull
// The reference has no counterpart (single process); these replace host-sized torch.argsort / bincount / index_copy_ chains.
#include "lstep_common.h"

namespace lstep {

constexpr int kPartTile = 1024;      // entries per workgroup (16 waves)
constexpr int kPartWaves = kPartTile / kWave;
constexpr int kMaxWorld = 16;

// Pass 1: tile_counts[tile, p] = live entries of the tile owned by rank p; also zero-fills the output blocks (dead slots = node 0, position 0).
__global__ __launch_bounds__(kPartTile) void owner_count_kernel(const int64_t* __restrict__ bn, int64_t cap, const int32_t* __restrict__ n_live,
                                                                int32_t world, int32_t* __restrict__ tile_counts, int64_t* __restrict__ ids_out,
                                                                int32_t* __restrict__ pos_out, int64_t out_slots) {
    __shared__ int32_t sh[kMaxWorld];
    const int lane = threadIdx.x & (kWave - 1);
    if (threadIdx.x < kMaxWorld) sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t live = n_live[0] < cap ? n_live[0] : cap;
    const int64_t i = (int64_t)blockIdx.x * kPartTile + threadIdx.x;
    const int owner = i < live ? (int)(bn[i] % world) : -1;
    for (int p = 0; p < world; ++p) {
        const int c = __popcll(__ballot(owner == p));
        if (lane == 0 && c) atomicAdd(&sh[p], c);
    }
    __syncthreads();
    if ((int)threadIdx.x < world) tile_counts[(int64_t)blockIdx.x * world + threadIdx.x] = sh[threadIdx.x];
    for (int64_t k = (int64_t)blockIdx.x * kPartTile + threadIdx.x; k < out_slots; k += (int64_t)gridDim.x * kPartTile) {
        ids_out[k] = 0;
        pos_out[k] = 0;
    }
}

// Pass 2: entry i (owner p, the r-th live entry of that owner in list order) goes to slot p * C + r: ids_out = its node id, pos_out = i.
// The list is sorted by id, so every block is sorted by id too.  The last tile writes counts[p] = min(total_p, C) and ORs the overflow word
// when some owner has more than C entries (the caller's capacity was too small: entries beyond C are dropped, the step's result is invalid).
__global__ __launch_bounds__(kPartTile) void owner_scatter_kernel(const int64_t* __restrict__ bn, int64_t cap, const int32_t* __restrict__ n_live,
                                                                  int32_t world, int64_t C, const int32_t* __restrict__ tile_counts,
                                                                  int64_t* __restrict__ ids_out, int32_t* __restrict__ pos_out,
                                                                  int32_t* __restrict__ counts, int32_t* __restrict__ overflow) {
    __shared__ int32_t base[kMaxWorld];
    __shared__ int32_t wave_cnt[kPartWaves][kMaxWorld];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (threadIdx.x < kMaxWorld) base[threadIdx.x] = 0;
    __syncthreads();
    // entries of every owner in the tiles before this one
    for (int64_t k = threadIdx.x; k < (int64_t)blockIdx.x * world; k += kPartTile) {
        const int32_t c = tile_counts[k];
        if (c) atomicAdd(&base[k % world], c);
    }
    const int64_t live = n_live[0] < cap ? n_live[0] : cap;
    const int64_t i = (int64_t)blockIdx.x * kPartTile + threadIdx.x;
    const int64_t id = i < live ? bn[i] : 0;
    const int owner = i < live ? (int)(id % world) : -1;
    int rank_in_wave = 0;
    for (int p = 0; p < world; ++p) {
        const unsigned long long m = __ballot(owner == p);
        if (owner == p) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave][p] = __popcll(m);
    }
    __syncthreads();
    if (owner >= 0) {
        int64_t r = base[owner] + rank_in_wave;
        for (int w = 0; w < wave; ++w) r += wave_cnt[w][owner];
        if (r < C) {
            ids_out[(int64_t)owner * C + r] = id;
            pos_out[(int64_t)owner * C + r] = (int32_t)i;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < world) {
        int64_t total = base[threadIdx.x];
        for (int w = 0; w < kPartWaves; ++w) total += wave_cnt[w][threadIdx.x];
        counts[threadIdx.x] = (int32_t)(total < C ? total : C);
        if (total > C) atomicOr(overflow, 1);
    }
}

// table[ids[p * C + i], :width] = rows[p * C + i, :width] for i < counts[p]; slot_of[id] = p * C + i (optional).  One wave per slot.
__global__ __launch_bounds__(kBlock) void scatter_owner_rows_kernel(const float* __restrict__ rows, int ld_rows, const int64_t* __restrict__ ids,
                                                                     const int32_t* __restrict__ counts, int32_t world, int64_t C,
                                                                     float* __restrict__ table, int width, int32_t* __restrict__ slot_of) {
    const int lane = lane_id();
    const int64_t e = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (e >= (int64_t)world * C) return;
    const int64_t p = e / C, i = e - p * C;
    if (i >= counts[p]) return;
    const int64_t id = LSTEP_CHECKED(ids[e], LSTEP_NODE_ROWS(), kCheckOwnerRowsId);      // (checked builds only: lstep_common.h)
    for (int c = lane; c < (width >> 2); c += kWave) st4(table + id * width + c * 4, ld4(rows + e * (int64_t)ld_rows + c * 4));
    if (slot_of && lane == 0) slot_of[id] = (int32_t)e;
}

// direction 0: buf[e, :width] = table[ids[e], :width];  direction 1: table[ids[e], :width] = buf[e, :width];  entries with ids[e] < 0 are holes.
__global__ __launch_bounds__(kBlock) void rows_by_id_kernel(const int32_t* __restrict__ ids, int64_t n, float* __restrict__ table, int width,
                                                             float* __restrict__ buf, int direction) {
    const int lane = lane_id();
    const int64_t e = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (e >= n) return;
    const int64_t id = ids[e];
    if (id < 0) return;
#ifdef LSTEP_BOUNDS_CHECK
    if (!check_index(id, LSTEP_NODE_ROWS(), kCheckRowsById)) return;      // (checked builds: an id past the table is skipped and recorded)
#endif
    float* t = table + id * width;
    float* b = buf + e * (int64_t)width;
    for (int c = lane; c < (width >> 2); c += kWave) {
        if (direction == 0) st4(b + c * 4, ld4(t + c * 4)); else st4(t + c * 4, ld4(b + c * 4));
    }
}

}  // namespace lstep

using namespace lstep;

extern "C" int64_t lstep_owner_partition_workspace(int64_t capacity, int32_t world) {
    if (capacity <= 0 || world < 1) return 0;
    return ((capacity + kPartTile - 1) / kPartTile) * (int64_t)world * (int64_t)sizeof(int32_t);
}

extern "C" int lstep_owner_partition(const int64_t* ids, int64_t capacity, const int32_t* num_live, int32_t world, int64_t block_slots,
                                     void* workspace, int64_t workspace_bytes, int64_t* ids_by_owner, int32_t* pos_by_owner, int32_t* counts,
                                     int32_t* overflow, void* stream) {
    if (capacity <= 0 || world < 1 || world > kMaxWorld || block_slots <= 0 || capacity >= ((int64_t)1 << 31))
        return set_error(LSTEP_EINVAL, "lstep_owner_partition: bad sizes (world <= %d)", kMaxWorld);
    if (!ids || !num_live || !workspace || !ids_by_owner || !pos_by_owner || !counts || !overflow)
        return set_error(LSTEP_EINVAL, "lstep_owner_partition: NULL pointer");
    if (workspace_bytes < lstep_owner_partition_workspace(capacity, world)) return set_error(LSTEP_EINVAL, "lstep_owner_partition: workspace too small");
    const unsigned tiles = (unsigned)((capacity + kPartTile - 1) / kPartTile);
    hipLaunchKernelGGL(owner_count_kernel, dim3(tiles), dim3(kPartTile), 0, (hipStream_t)stream, ids, capacity, num_live, world, (int32_t*)workspace,
                       ids_by_owner, pos_by_owner, (int64_t)world * block_slots);
    hipLaunchKernelGGL(owner_scatter_kernel, dim3(tiles), dim3(kPartTile), 0, (hipStream_t)stream, ids, capacity, num_live, world, block_slots,
                       (const int32_t*)workspace, ids_by_owner, pos_by_owner, counts, overflow);
    return check_launch("owner_partition kernels");
}

extern "C" int lstep_scatter_owner_rows(const float* rows, int32_t ld_rows, const int64_t* ids_by_owner, const int32_t* counts, int32_t world,
                                        int64_t block_slots, float* table, int32_t width, int32_t* slot_of, void* stream) {
    if (world < 1 || block_slots <= 0 || width <= 0 || (width & 3) || ld_rows < width || (ld_rows & 3))
        return set_error(LSTEP_EINVAL, "lstep_scatter_owner_rows: bad sizes");
    if (!rows || !ids_by_owner || !counts || !table) return set_error(LSTEP_EINVAL, "lstep_scatter_owner_rows: NULL pointer");
    const int64_t n = (int64_t)world * block_slots;
    hipLaunchKernelGGL(scatter_owner_rows_kernel, dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, rows,
                       (int)ld_rows, ids_by_owner, counts, world, block_slots, table, (int)width, slot_of);
    return check_launch("scatter_owner_rows_kernel");
}

extern "C" int lstep_rows_by_id(const int32_t* ids, int64_t n, float* table, int32_t width, float* buf, int32_t direction, void* stream) {
    if (n < 0 || width <= 0 || (width & 3) || (direction != 0 && direction != 1)) return set_error(LSTEP_EINVAL, "lstep_rows_by_id: bad arguments");
    if (n == 0) return LSTEP_OK;
    if (!ids || !table || !buf) return set_error(LSTEP_EINVAL, "lstep_rows_by_id: NULL pointer");
    hipLaunchKernelGGL(rows_by_id_kernel, dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, ids, n, table,
                       (int)width, buf, (int)direction);
    return check_launch("rows_by_id_kernel");
}
