// F: linear core of fourier_transform_pe (reference models/LSTEP.py:104-137) over a strided PE history.
//   fwd  out[u, :]   = sum_s coef[s, :] * hist[node_u, s, :]          streams U * t_len rows of P floats once
//   bwd  dcoef[s, :] = sum_u grad[u, :] * hist[node_u, s, :]          streams the same rows once more
// Both are pure HBM streams of 4P-byte rows (688 B for P = 172); coef / grad rows are re-read from L2.
#include <cstdlib>

#include "lstep_common.h"

namespace lstep {

struct HistView {
    const float* base;
    int64_t node_stride, time_stride;
    int slots, rot;
    const float* oldest;   // optional [rows, node_stride] table holding the window's oldest snapshot (rings whose slots only store changed rows)
    const int32_t* rot_dev = nullptr;   // lstep_ring_ref_t: the rotation lives on the device (rot = (*rot_dev + rot_add) % slots)
    int rot_add = 0;
    __device__ __forceinline__ void resolve() {
        if (rot_dev) rot = (*rot_dev + rot_add) % slots;
    }
    __device__ __forceinline__ const float* row(int64_t node, int s) const {
        int ph = s + rot;
        if (ph >= slots) ph -= slots;
        return base + node * node_stride + (int64_t)ph * time_stride;
    }
    // The row of the run that begins at snapshot s.  s = 0, the window's oldest snapshot: its slot holds the row if that batch wrote it
    // (`first_written`), otherwise the row is older than the window and lives in `oldest` -- which may lag one window slide behind:
    // the rows it still has to take over are exactly the ones read from the slot here.
    __device__ __forceinline__ const float* begin_row(int64_t node, int s, bool first_written) const {
        return (s == 0 && oldest && !first_written) ? oldest + node * node_stride : row(node, s);
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
    const int64_t node = LSTEP_CHECKED(ids[u], LSTEP_NODE_ROWS(), kCheckFilterNode);      // (checked builds only: lstep_common.h)
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

constexpr int kBwdNodesPerChunk = 128;
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
        const int64_t node = LSTEP_CHECKED(ids[u], LSTEP_NODE_ROWS(), kCheckFilterNode);
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

// ---- change-aware variants ------------------------------------------------------------------------------------------------
// The reference appends a CLONE of the whole PE table per batch (train_link_prediction.py:229,301), so a node's row in snapshot s is
// bit-identical to its row in snapshot s-1 unless the batch in between wrote it (splice, update_pe phase 1 / 2).  The device ring
// records exactly that in a per-node bit mask over its PHYSICAL slots (bit ph = "slot ph differs from slot ph-1"), and the filter
// only reads the first row of every run of equal rows:
//   fwd  out[u]   = sum_runs (cpre[run end] - cpre[run begin]) * row(run begin),   cpre[s] = sum_{s' < s} coef[s']  (float64 table)
//   bwd  D[b]     = sum_u g[u] * (row(b) - row(previous run's begin))  at every run begin b;   dcoef = cumsum_s(D)
// Same sums as the dense kernels, one row read per run instead of one per snapshot.

struct ChangeBits {   // logical-order view of a node's change mask: bit s = "snapshot s of the window differs from snapshot s-1"
    uint64_t lo, hi;
    __device__ __forceinline__ bool test(int s) const { return ((s < 64 ? lo >> s : hi >> (s - 64)) & 1ull) != 0; }
    // last set bit at a position in [1, s] (s >= 1), or 0 when there is none: the begin of the run that snapshot s belongs to
    __device__ __forceinline__ int prev(int s) const {
        if (s >= 64) {
            const uint64_t m = s - 64 >= 63 ? hi : hi & ((2ull << (s - 64)) - 1ull);
            if (m) return 127 - __builtin_clzll(m);
            s = 63;
        }
        const uint64_t m = (s >= 63 ? lo : lo & ((2ull << s) - 1ull)) & ~1ull;
        return m ? 63 - __builtin_clzll(m) : 0;
    }
    // first set bit at position >= s, or `limit` when there is none below it
    __device__ __forceinline__ int next(int s, int limit) const {
        if (s >= limit) return limit;
        if (s < 64) {
            const uint64_t m = lo >> s;
            if (m) { const int r = s + __builtin_ctzll(m); return r < limit ? r : limit; }
            s = 64;
            if (s >= limit) return limit;
        }
        const uint64_t m = hi >> (s - 64);
        if (!m) return limit;
        const int r = s + __builtin_ctzll(m);
        return r < limit ? r : limit;
    }
};

__device__ __forceinline__ void shr128(uint64_t& lo, uint64_t& hi, int n) {   // 0 <= n < 128
    if (n >= 64) { lo = hi >> (n - 64); hi = 0; }
    else if (n > 0) { lo = (lo >> n) | (hi << (64 - n)); hi >>= n; }
}
__device__ __forceinline__ void shl128(uint64_t& lo, uint64_t& hi, int n) {   // 0 <= n <= 128
    if (n >= 128) { lo = 0; hi = 0; }
    else if (n >= 64) { hi = lo << (n - 64); lo = 0; }
    else if (n > 0) { hi = (hi << n) | (lo >> (64 - n)); lo <<= n; }
}

// every lane of the wave holds the same node id: move it to scalar registers, so row addresses are scalar base + lane offset
__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// all lanes of the wave read the same 4 * words bytes; the result lives in scalar registers
__device__ __forceinline__ ChangeBits load_change_bits(const uint32_t* __restrict__ mask, int words, int64_t node, int slots, int rot) {
    const uint32_t* m = mask + node * words;
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = i < words ? (uint32_t)__builtin_amdgcn_readfirstlane((int)m[i]) : 0u;
    uint64_t alo = (uint64_t)w[0] | ((uint64_t)w[1] << 32), ahi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    uint64_t blo = alo, bhi = ahi;
    shr128(alo, ahi, rot);            // physical slots rot .. slots-1  -> logical 0 ..
    shl128(blo, bhi, slots - rot);    // physical slots 0 .. rot-1      -> logical slots-rot ..
    return ChangeBits{alo | blo, ahi | bhi};
}

struct alignas(16) double2_ { double x, y; };

constexpr int kRunsInFlight = 8;
constexpr int kRunsLdsLimit = 160 * 1024 - 512;   // bytes of LDS one workgroup may take on gfx950 (160 KB per CU)
constexpr int kRunsFwdBlocks = 256 * 4;   // persistent grid of the forward kernel: 4 workgroups (16 waves: 118 registers each) per CU


// persistent waves: wave w handles nodes w, w + waves, ...; the next node's id and mask are fetched while the current node's rows travel.
// cpre = the float64 prefix table of coef (coef_prefix_kernel).  IN_LDS: the workgroup (16 waves, one per CU) first copies it into LDS
// ((t_len + 1) * P * 8 bytes, 139 KB for T = 100, P = 172) and reads the run weights from there: with the table in L2 every run costs
// 1376 B of L1 fills next to its 688-B row, and the kernel was bound by that (245 us against 155 us with the weights faked).
template <bool IN_LDS>
__global__ __launch_bounds__(IN_LDS ? 1024 : kBlock) void history_filter_runs_fwd_kernel(HistView h, int t_len, int P, const uint32_t* __restrict__ mask,
                                                                                         int words, const int64_t* __restrict__ ids, int64_t num_ids,
                                                                                         const float* __restrict__ coef, const double* __restrict__ cpre,
                                                                                         float* __restrict__ out, float* __restrict__ table_out,
                                                                                         int32_t* __restrict__ slot_of, const int32_t* __restrict__ num_live) {
    extern __shared__ double cpre_lds[];
    h.resolve();
    if (num_live) {      // ids is a capacity-sized list: only its first min(*num_live, num_ids) entries are nodes of this batch
        const int64_t live = *num_live;
        if (live < num_ids) num_ids = live;
    }
    const int lane = lane_id();
    const int waves_per_block = (int)(blockDim.x >> 6);
    if (IN_LDS) {
        // copy the prefix table coef_prefix_kernel made (139 KB, L2-resident) with all 1024 threads, 16 bytes each: a few microseconds,
        // where building it here kept 172 threads busy with 100 dependent steps each (~30 us of every launch, whatever the batch size)
        const int n2 = ((t_len + 1) * P) >> 1;               // (P is a multiple of 4: the table is a whole number of double2)
        const double2_* src = reinterpret_cast<const double2_*>(cpre);
        double2_* dst = reinterpret_cast<double2_*>(cpre_lds);
        for (int k = threadIdx.x; k < n2; k += blockDim.x) dst[k] = src[k];
        __syncthreads();
    }
    const int64_t waves = (int64_t)gridDim.x * waves_per_block;
    int64_t u = (int64_t)blockIdx.x * waves_per_block + wave_in_block();
    if (u >= num_ids) return;
    const bool active = lane < (P >> 2);
    const int col = active ? lane * 4 : 0;
    int64_t node = uniform_i64(LSTEP_CHECKED(ids[u], LSTEP_NODE_ROWS(), kCheckFilterNode));
    ChangeBits bits = load_change_bits(mask, words, node, h.slots, h.rot);
    for (; u < num_ids; u += waves) {
        const int64_t node_now = node;
        const ChangeBits b = bits;
        if (u + waves < num_ids) {
            node = uniform_i64(LSTEP_CHECKED(ids[u + waves], LSTEP_NODE_ROWS(), kCheckFilterNode));
            bits = load_change_bits(mask, words, node, h.slots, h.rot);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cur = 0;                                  // the oldest snapshot always starts a run
        while (cur < t_len) {
            int st[kRunsInFlight + 1];
            st[0] = cur;
#pragma unroll
            for (int i = 1; i <= kRunsInFlight; ++i) st[i] = b.next(st[i - 1] + 1, t_len);
            float4 x[kRunsInFlight];
            double2_ c[kRunsInFlight + 1][2];
#pragma unroll
            for (int i = 0; i < kRunsInFlight; ++i) {
                x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (st[i] < t_len) x[i] = ld4_stream(h.begin_row(node_now, st[i], b.test(0)) + col);
            }
#pragma unroll
            for (int i = 0; i <= kRunsInFlight; ++i) {      // the prefix table has t_len + 1 rows
                const double2_* cp = IN_LDS ? reinterpret_cast<const double2_*>(cpre_lds + st[i] * P + col)
                                            : reinterpret_cast<const double2_*>(cpre + (int64_t)st[i] * P + col);
                c[i][0] = cp[0];
                c[i][1] = cp[1];
            }
#pragma unroll
            for (int i = 0; i < kRunsInFlight; ++i) {
                acc.x = fmaf((float)(c[i + 1][0].x - c[i][0].x), x[i].x, acc.x);
                acc.y = fmaf((float)(c[i + 1][0].y - c[i][0].y), x[i].y, acc.y);
                acc.z = fmaf((float)(c[i + 1][1].x - c[i][1].x), x[i].z, acc.z);
                acc.w = fmaf((float)(c[i + 1][1].y - c[i][1].y), x[i].w, acc.w);
            }
            cur = st[kRunsInFlight];
        }
        if (active) {
            st4(out + u * (int64_t)P + col, acc);
            if (table_out) st4(table_out + node_now * h.node_stride + col, acc);     // the splice of train:230: pe[batch nodes] = filtered rows
        }
        if (slot_of && lane == 0) slot_of[node_now] = (int32_t)u;
    }
}

// cpre[s, p] = sum_{s' < s} coef[s', p] in float64, rows 0 .. t_len.  One wave per column: lane l owns snapshots [l * per, (l + 1) * per)
// (per = ceil(t_len / 64)), a wave-wide exclusive scan of the lane sums gives each lane its offset.
__global__ __launch_bounds__(kBlock) void coef_prefix_kernel(const float* __restrict__ coef, int t_len, int P, double* __restrict__ cpre) {
    const int lane = lane_id();
    const int p = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (p >= P) return;
    const int per = (t_len + kWave - 1) / kWave;
    const int s0 = lane * per;
    double sum = 0.0;
    for (int i = 0; i < per; ++i)
        if (s0 + i < t_len) sum += (double)coef[(int64_t)(s0 + i) * P + p];
    double incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const double up = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += up;
    }
    double run = incl - sum;                    // exclusive prefix of this lane's block
    if (lane == 0) cpre[p] = 0.0;
    for (int i = 0; i < per; ++i)
        if (s0 + i < t_len) {
            run += (double)coef[(int64_t)(s0 + i) * P + p];
            cpre[(int64_t)(s0 + i + 1) * P + p] = run;
        }
}

constexpr int kRunsTimeGroup = 16;   // snapshots per wave of the backward pass: one forced read per group, runs in between

// wave = (node chunk, group of 16 snapshots).  partial[chunk, s] = sum over the chunk's nodes of g * (row(s) - previous run's row) at the
// run begins s of the group (the group's first snapshot counts as a run begin with "previous row" = 0), zero elsewhere.
__global__ __launch_bounds__(kBlock) void history_filter_runs_bwd_kernel(HistView h, int t_len, int P, const uint32_t* __restrict__ mask, int words,
                                                                          const int64_t* __restrict__ ids, int64_t num_ids,
                                                                          const float* __restrict__ grad, float* __restrict__ partial, int groups,
                                                                          int nodes_per_chunk) {
    const int lane = lane_id();
    h.resolve();
    const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t chunk = w / groups;
    const int grp = (int)(w - chunk * groups);
    const int64_t u0 = chunk * nodes_per_chunk;
    if (u0 >= num_ids) return;
    const int s0 = grp * kRunsTimeGroup;
    const int64_t u1 = (u0 + nodes_per_chunk < num_ids) ? u0 + nodes_per_chunk : num_ids;
    const bool active = lane < (P >> 2);
    const int col = active ? lane * 4 : 0;
    float4 acc[kRunsTimeGroup];
#pragma unroll
    for (int i = 0; i < kRunsTimeGroup; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t valid = t_len - s0 >= kRunsTimeGroup ? 0xFFFFu : (1u << (t_len - s0)) - 1u;   // s0 < t_len: groups = ceil(t_len / 16)
    int64_t node = uniform_i64(LSTEP_CHECKED(ids[u0], LSTEP_NODE_ROWS(), kCheckFilterNode));
    ChangeBits bits = load_change_bits(mask, words, node, h.slots, h.rot);
    for (int64_t u = u0; u < u1; ++u) {
        const int64_t node_now = node;
        const ChangeBits b = bits;
        if (u + 1 < u1) {     // next node's id and mask travel while this node's rows do
            node = uniform_i64(LSTEP_CHECKED(ids[u + 1], LSTEP_NODE_ROWS(), kCheckFilterNode));
            bits = load_change_bits(mask, words, node, h.slots, h.rot);
        }
        const float4 g = ld4(grad + u * (int64_t)P + col);
        float4 x[kRunsTimeGroup];
        // run begins of this group as one scalar bit field (s0 is a multiple of 16: the field never straddles the two mask halves)
        const uint32_t begins = (((uint32_t)(s0 < 64 ? b.lo >> s0 : b.hi >> (s0 - 64)) & valid) | 1u);
        // the group's first snapshot: its own row if it begins a run (or if every slot holds a full clone), else the row its run began with
        const int first = (h.oldest == nullptr || s0 == 0 || b.test(s0)) ? s0 : b.prev(s0);
#pragma unroll
        for (int i = 0; i < kRunsTimeGroup; ++i) {
            x[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // (defined on both paths: otherwise the merge costs a copy and a wait per load)
            if ((begins >> i) & 1u) x[i] = ld4_stream(h.begin_row(node_now, i == 0 ? first : s0 + i, b.test(0)) + col);
        }
        float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < kRunsTimeGroup; ++i) {
            if ((begins >> i) & 1u) {
                acc[i].x = fmaf(g.x, x[i].x - prev.x, acc[i].x);
                acc[i].y = fmaf(g.y, x[i].y - prev.y, acc[i].y);
                acc[i].z = fmaf(g.z, x[i].z - prev.z, acc[i].z);
                acc[i].w = fmaf(g.w, x[i].w - prev.w, acc[i].w);
                prev = x[i];
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int i = 0; i < kRunsTimeGroup; ++i)
        if (s0 + i < t_len) st4(partial + (chunk * t_len + s0 + i) * (int64_t)P + col, acc[i]);
}

// dcoef[s] = sum of D[s'] over the snapshots s' <= s of s's group (float64 running sum)
__global__ __launch_bounds__(kBlock) void history_runs_finish_kernel(const float* __restrict__ dsum, int t_len, int P, int groups, float* __restrict__ dcoef) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= groups * P) return;
    const int grp = i / P, p = i - grp * P;
    double run = 0.0;
    for (int s = grp * kRunsTimeGroup; s < (grp + 1) * kRunsTimeGroup && s < t_len; ++s) {
        run += (double)dsum[(int64_t)s * P + p];
        dcoef[(int64_t)s * P + p] = (float)run;
    }
}

// slot index from a device-resident ring position (lstep_ring_ref_t), or the host value
struct SlotRef {
    const int32_t* start;
    int add, slots;
    int64_t stride;
    __device__ __forceinline__ int slot(int host_slot) const { return start ? (*start + add) % slots : host_slot; }
};
static SlotRef slot_ref(const lstep_ring_ref_t* r) { return r ? SlotRef{r->start, r->add, r->slots, r->slot_stride} : SlotRef{nullptr, 0, 1, 0}; }
static int check_ring(const char* who, const lstep_ring_ref_t* r) {
    if (r && (!r->start || r->slots <= 0 || r->add < 0 || r->slot_stride < 0)) return set_error(LSTEP_EINVAL, "%s: bad ring reference", who);
    return LSTEP_OK;
}

__global__ void ring_tick_kernel(int32_t* start, int slots) { *start = (*start + 1) % slots; }

__global__ __launch_bounds__(kBlock) void history_mark_kernel(uint32_t* __restrict__ mask, int words, int64_t num_rows, int slot,
                                                               const int64_t* __restrict__ ids, int64_t num_ids, int world, int rank, SlotRef ring) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= num_ids) return;
    slot = ring.slot(slot);
    int64_t node = ids[i];
    // the padding node: history_slot_bits_kernel sets its bit for every slot, and capacity-sized id lists end in a long run of zeros
    // (hundreds of thousands of atomics on ONE word took 4.5 ms per batch)
    if (node == 0) return;
    if (world > 1) {
        if (node % world != rank) return;
        node /= world;
    }
    if (node < 0 || node >= num_rows) return;
    atomicOr(mask + node * words + (slot >> 5), 1u << (slot & 31));
}

// dst[r] = src[r] for the listed rows r (wave per row; rows outside [0, num_rows) are ignored)
__global__ __launch_bounds__(kBlock) void copy_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, int width, int64_t ld,
                                                            const int64_t* __restrict__ ids, int64_t num_ids, int64_t num_rows, SlotRef ring) {
    const int lane = lane_id();
    if (ring.start) dst += (int64_t)ring.slot(0) * ring.stride;
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (i >= num_ids) return;
    const int64_t r = ids[i];
    if (r < 0 || r >= num_rows) return;
    for (int c = lane * 4; c < width; c += kWave * 4) st4(dst + r * ld + c, ld4(src + r * ld + c));
}

// oldest[r] = slot_rows[r] for every row whose bit of `slot` is set: the window's oldest snapshot moves on by one batch.
// A wave takes 64 consecutive rows: one coalesced read of their mask words, a ballot, and the changed rows (~21 % on the c4 workload) are
// copied eight at a time -- all eight loads in flight before the first store.  (One wave per ROW, as in rounds 1-3, is a million waves of
// which four in five read one word and leave: 162 us for 0.42 GB = 2.6 TB/s, VERDICT r3 "HBM-side stragglers".)
constexpr int kAdvanceInFlight = 8;
__global__ __launch_bounds__(kBlock) void history_advance_oldest_kernel(float* __restrict__ oldest, const float* __restrict__ slot_rows, int width, int64_t ld,
                                                                         const uint32_t* __restrict__ mask, int words, int slot, int64_t num_rows,
                                                                         SlotRef ring) {
    const int lane = lane_id();
    const int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wave_in_block()) * kWave;
    if (base >= num_rows) return;
    if (ring.start) {
        slot = ring.slot(slot);
        slot_rows += (int64_t)slot * ring.stride;
    }
    const int64_t r = base + lane;
    const bool hit = r < num_rows && ((mask[r * words + (slot >> 5)] >> (slot & 31)) & 1u);
    unsigned long long todo = __ballot(hit);
    const bool on = lane * 4 < width;
    while (todo) {
        int rows[kAdvanceInFlight];
        int n = 0;
#pragma unroll
        for (int u = 0; u < kAdvanceInFlight; ++u) {
            if (todo) {
                rows[u] = __builtin_ctzll(todo);
                todo &= todo - 1;
                n = u + 1;
            } else {
                rows[u] = rows[0];          // (re-read the first row of the group: a cache hit, never stored)
            }
        }
        float4 v[kAdvanceInFlight];
        if (on) {
#pragma unroll
            for (int u = 0; u < kAdvanceInFlight; ++u) v[u] = ld4_stream(slot_rows + (base + rows[u]) * ld + lane * 4);
#pragma unroll
            for (int u = 0; u < kAdvanceInFlight; ++u)
                if (u < n) st4(oldest + (base + rows[u]) * ld + lane * 4, v[u]);
        }
    }
}

__global__ __launch_bounds__(kBlock) void history_slot_bits_kernel(uint32_t* __restrict__ mask, int words, int64_t num_rows, int slot, int value,
                                                                    SlotRef ring) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= num_rows) return;
    slot = ring.slot(slot);
    uint32_t* w = mask + i * words + (slot >> 5);
    const uint32_t bit = 1u << (slot & 31);
    // atomics: a word holds the bits of 32 slots, and history_mark_kernel may be setting another slot's bit of the same word from
    // another stream (a clone prefetched on the copy stream while the batch's writers mark theirs); row 0 = the padding row, which
    // every update_pe rewrites (LSTEP.py:317)
    if (value || i == 0) atomicOr(w, bit); else atomicAnd(w, ~bit);
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
    HistView h{hist, node_stride, time_stride, time_slots, time_rot, nullptr};
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
    HistView h{hist, node_stride, time_stride, time_slots, time_rot, nullptr};
    const int groups = (t_len + kBwdTimeGroup - 1) / kBwdTimeGroup;
    const int64_t waves = lstep_history_filter_bwd_chunks(num_ids) * groups;
    const unsigned grid = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(history_filter_bwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, h, (int)t_len, (int)pe_dim, node_ids,
                       num_ids, grad_out, out_partial, groups);
    return check_launch("history_filter_bwd_kernel");
}

static int check_mask(const char* who, const uint32_t* mask, int32_t words, int32_t slots) {
    if (!mask) return set_error(LSTEP_EINVAL, "%s: NULL change mask", who);
    if (words < 1 || words > 4 || slots > 32 * words) return set_error(LSTEP_EINVAL, "%s: %d mask words cannot hold %d slots (at most 128)", who, words, slots);
    return LSTEP_OK;
}

extern "C" int lstep_history_mark(uint32_t* mask, int32_t mask_words, int64_t num_rows, int32_t slot, const int64_t* ids, int64_t num_ids,
                                  int32_t world, int32_t rank, const lstep_ring_ref_t* ring, void* stream) {
    if (num_ids < 0 || num_rows < 0) return set_error(LSTEP_EINVAL, "lstep_history_mark: negative count");
    if (num_ids == 0) return LSTEP_OK;
    if (int rc = check_ring("lstep_history_mark", ring)) return rc;
    if (ring) slot = ring->slots - 1;     // (the mask must hold every slot the device may pick)
    if (int rc = check_mask("lstep_history_mark", mask, mask_words, slot + 1)) return rc;
    if (slot < 0 || !ids || world < 1 || rank < 0 || rank >= world) return set_error(LSTEP_EINVAL, "lstep_history_mark: bad argument");
    const unsigned grid = (unsigned)((num_ids + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(history_mark_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, mask, (int)mask_words, num_rows, (int)slot, ids, num_ids,
                       (int)world, (int)rank, slot_ref(ring));
    return check_launch("history_mark_kernel");
}

extern "C" int lstep_history_slot_bits(uint32_t* mask, int32_t mask_words, int64_t num_rows, int32_t slot, int32_t value,
                                       const lstep_ring_ref_t* ring, void* stream) {
    if (num_rows < 0) return set_error(LSTEP_EINVAL, "lstep_history_slot_bits: negative count");
    if (num_rows == 0) return LSTEP_OK;
    if (int rc = check_ring("lstep_history_slot_bits", ring)) return rc;
    if (ring) slot = ring->slots - 1;
    if (int rc = check_mask("lstep_history_slot_bits", mask, mask_words, slot + 1)) return rc;
    if (slot < 0) return set_error(LSTEP_EINVAL, "lstep_history_slot_bits: negative slot");
    const unsigned grid = (unsigned)((num_rows + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(history_slot_bits_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, mask, (int)mask_words, num_rows, (int)slot, (int)value,
                       slot_ref(ring));
    return check_launch("history_slot_bits_kernel");
}

extern "C" int64_t lstep_history_filter_runs_workspace(int32_t t_len, int32_t pe_dim) {
    return t_len < 0 || pe_dim <= 0 ? 0 : (int64_t)(t_len + 1) * pe_dim * (int64_t)sizeof(double);
}

extern "C" int lstep_history_filter_runs_fwd(const float* hist, int64_t node_stride, int64_t time_stride, int32_t time_slots, int32_t time_rot,
                                             int32_t t_len, int32_t pe_dim, const uint32_t* mask, int32_t mask_words, const float* oldest,
                                             const int64_t* node_ids, int64_t num_ids, const float* coef, void* workspace, float* out,
                                             float* table_out, int32_t* slot_of, const int32_t* num_live, const lstep_ring_ref_t* ring, void* stream) {
    if (num_ids < 0) return set_error(LSTEP_EINVAL, "lstep_history_filter_runs_fwd: negative count");
    if (num_ids == 0) return LSTEP_OK;
    if (int rc = check_ring("lstep_history_filter_runs_fwd", ring)) return rc;
    if (int rc = check_hist("lstep_history_filter_runs_fwd", hist, node_stride, time_stride, time_slots, time_rot, t_len, pe_dim)) return rc;
    if (int rc = check_mask("lstep_history_filter_runs_fwd", mask, mask_words, time_slots)) return rc;
    if (!node_ids || !coef || !out || !workspace || (((uintptr_t)workspace) & 15) || (((uintptr_t)oldest) & 15))
        return set_error(LSTEP_EINVAL, "lstep_history_filter_runs_fwd: NULL or misaligned pointer");
    if (((uintptr_t)table_out) & 15) return set_error(LSTEP_EINVAL, "lstep_history_filter_runs_fwd: misaligned table_out");
    HistView h{hist, node_stride, time_stride, time_slots, time_rot, oldest};
    if (ring) { h.rot_dev = ring->start; h.rot_add = ring->add; }
    double* cpre = (double*)workspace;
    const size_t lds_bytes = (size_t)(t_len + 1) * pe_dim * sizeof(double);
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&history_filter_runs_fwd_kernel<true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kRunsLdsLimit) == hipSuccess;
    if (big_lds && lds_bytes <= (size_t)kRunsLdsLimit && num_ids >= 2048 && !getenv("LSTEP_RUNS_NO_LDS")) {
        // one 16-wave workgroup per CU, prefix table in LDS
        int64_t blocks = (num_ids + 15) / 16;
        if (blocks > 256) blocks = 256;
        hipLaunchKernelGGL(coef_prefix_kernel, dim3((unsigned)((pe_dim + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                           coef, (int)t_len, (int)pe_dim, cpre);
        hipLaunchKernelGGL(history_filter_runs_fwd_kernel<true>, dim3((unsigned)blocks), dim3(1024), lds_bytes, (hipStream_t)stream, h, (int)t_len,
                           (int)pe_dim, mask, (int)mask_words, node_ids, num_ids, coef, (const double*)cpre, out, table_out, slot_of, num_live);
        return check_launch("history_filter_runs_fwd_kernel<lds>");
    }
    hipLaunchKernelGGL(coef_prefix_kernel, dim3((unsigned)((pe_dim + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       coef, (int)t_len, (int)pe_dim, cpre);
    int64_t blocks = (num_ids + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > kRunsFwdBlocks) blocks = kRunsFwdBlocks;
    hipLaunchKernelGGL(history_filter_runs_fwd_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, h, (int)t_len, (int)pe_dim,
                       mask, (int)mask_words, node_ids, num_ids, coef, (const double*)cpre, out, table_out, slot_of, num_live);
    return check_launch("history_filter_runs_fwd_kernel");
}

// Nodes per wave of the run backward.  A wave walks its nodes one after the other -- one round trip per node: its run rows of the wave's
// 16-snapshot group, ~4.5 loads -- so the launch takes (nodes per wave) x (a memory latency), whatever the number of waves, as long as they
// are all resident: 143 registers = 3 waves per SIMD = 3072 on the chip.  Rounds 1-4 used 128 nodes per wave (1 792 waves for the
// 1 M-node workload's 32 768 batch nodes: 1.75 per SIMD, 128 round trips: 272 us alone); halving the chunk doubles the waves past what
// is resident and the second round costs what the shorter chunks gain (64: 277 us, 32: 302 us with the extra partials, round 5).  So: as
// few nodes per wave as keeps every wave resident in ONE round -- ceil(nodes x groups / 3072), e.g. 75 for 32 768 nodes -- and
// never fewer than 4 (small batches: 400 nodes in 4 chunks of 128 took 250 us where 8-node chunks take 20).
static int runs_bwd_nodes_per_chunk(int64_t num_ids, int32_t t_len) {
    const int groups = (t_len + kRunsTimeGroup - 1) / kRunsTimeGroup;
    if (const char* e = getenv("LSTEP_RUNS_BWD_CHUNK")) {      // tuning knob (tools/history_bench.py)
        const int v = atoi(e);
        if (v >= 4 && v <= 1024) return v;
    }
    static const int resident = [] {
        int cus = 256;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        return cus * 4 * 3;       // SIMDs x waves per SIMD at this kernel's register count
    }();
    int64_t per = (num_ids * groups + resident - 1) / resident;
    per = (per + 3) / 4 * 4;
    if (per < 4) per = 4;
    if (per > 1024) per = 1024;
    return (int)per;
}

extern "C" int64_t lstep_history_filter_runs_bwd_chunks(int64_t num_ids, int32_t t_len) {
    if (num_ids <= 0 || t_len <= 0) return 0;
    const int per = runs_bwd_nodes_per_chunk(num_ids, t_len);
    return (num_ids + per - 1) / per;
}

extern "C" int lstep_history_filter_runs_bwd(const float* hist, int64_t node_stride, int64_t time_stride, int32_t time_slots, int32_t time_rot,
                                             int32_t t_len, int32_t pe_dim, const uint32_t* mask, int32_t mask_words, const float* oldest,
                                             const int64_t* node_ids, int64_t num_ids, const float* grad_out, float* out_partial,
                                             const lstep_ring_ref_t* ring, void* stream) {
    if (num_ids < 0) return set_error(LSTEP_EINVAL, "lstep_history_filter_runs_bwd: negative count");
    if (num_ids == 0 || t_len == 0) return LSTEP_OK;
    if (int rc = check_hist("lstep_history_filter_runs_bwd", hist, node_stride, time_stride, time_slots, time_rot, t_len, pe_dim)) return rc;
    if (int rc = check_mask("lstep_history_filter_runs_bwd", mask, mask_words, time_slots)) return rc;
    if (!node_ids || !grad_out || !out_partial || (((uintptr_t)oldest) & 15)) return set_error(LSTEP_EINVAL, "lstep_history_filter_runs_bwd: NULL or misaligned pointer");
    if (int rc = check_ring("lstep_history_filter_runs_bwd", ring)) return rc;
    HistView h{hist, node_stride, time_stride, time_slots, time_rot, oldest};
    if (ring) { h.rot_dev = ring->start; h.rot_add = ring->add; }
    const int groups = (t_len + kRunsTimeGroup - 1) / kRunsTimeGroup;
    const int per = runs_bwd_nodes_per_chunk(num_ids, t_len);
    const int64_t waves = ((num_ids + per - 1) / per) * groups;
    const unsigned grid = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(history_filter_runs_bwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, h, (int)t_len, (int)pe_dim, mask,
                       (int)mask_words, node_ids, num_ids, grad_out, out_partial, groups, per);
    return check_launch("history_filter_runs_bwd_kernel");
}

extern "C" int lstep_history_filter_runs_finish(const float* partial_sum, int32_t t_len, int32_t pe_dim, float* dcoef, void* stream) {
    if (t_len < 0 || pe_dim <= 0) return set_error(LSTEP_EINVAL, "lstep_history_filter_runs_finish: bad shape");
    if (t_len == 0) return LSTEP_OK;
    if (!partial_sum || !dcoef) return set_error(LSTEP_EINVAL, "lstep_history_filter_runs_finish: NULL pointer");
    const int groups = (t_len + kRunsTimeGroup - 1) / kRunsTimeGroup;
    const unsigned grid = (unsigned)((groups * pe_dim + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(history_runs_finish_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, partial_sum, (int)t_len, (int)pe_dim, groups, dcoef);
    return check_launch("history_runs_finish_kernel");
}

extern "C" int lstep_copy_rows(float* dst, const float* src, int32_t width, int64_t ld, const int64_t* ids, int64_t num_ids, int64_t num_rows,
                               const lstep_ring_ref_t* ring, void* stream) {
    if (num_ids < 0 || num_rows < 0) return set_error(LSTEP_EINVAL, "lstep_copy_rows: negative count");
    if (num_ids == 0) return LSTEP_OK;
    if (int rc = check_ring("lstep_copy_rows", ring)) return rc;
    if (ring && (ring->slot_stride & 3)) return set_error(LSTEP_EINVAL, "lstep_copy_rows: slot stride must keep rows 16-byte aligned");
    if (!dst || !src || !ids || width <= 0 || (width & 3) || ld < width || (ld & 3) || (((uintptr_t)dst | (uintptr_t)src) & 15))
        return set_error(LSTEP_EINVAL, "lstep_copy_rows: rows must be 16-byte aligned, width a multiple of 4");
    const unsigned grid = (unsigned)((num_ids + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, dst, src, (int)width, ld, ids, num_ids, num_rows,
                       slot_ref(ring));
    return check_launch("copy_rows_kernel");
}

extern "C" int lstep_history_advance_oldest(float* oldest, const float* slot_rows, int32_t width, int64_t ld, const uint32_t* mask,
                                            int32_t mask_words, int32_t slot, int64_t num_rows, const lstep_ring_ref_t* ring, void* stream) {
    if (num_rows < 0) return set_error(LSTEP_EINVAL, "lstep_history_advance_oldest: negative count");
    if (num_rows == 0) return LSTEP_OK;
    if (int rc = check_ring("lstep_history_advance_oldest", ring)) return rc;
    if (ring && (ring->slot_stride & 3)) return set_error(LSTEP_EINVAL, "lstep_history_advance_oldest: slot stride must keep rows 16-byte aligned");
    if (ring) slot = ring->slots - 1;
    if (int rc = check_mask("lstep_history_advance_oldest", mask, mask_words, slot + 1)) return rc;
    if (!oldest || !slot_rows || slot < 0 || width <= 0 || (width & 3) || ld < width || (ld & 3) || (((uintptr_t)oldest | (uintptr_t)slot_rows) & 15))
        return set_error(LSTEP_EINVAL, "lstep_history_advance_oldest: rows must be 16-byte aligned, width a multiple of 4");
    if (width > 4 * kWave) return set_error(LSTEP_EINVAL, "lstep_history_advance_oldest: rows wider than 256 floats are not supported");
    const int64_t waves = (num_rows + kWave - 1) / kWave;
    const unsigned grid = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(history_advance_oldest_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, oldest, slot_rows, (int)width, ld, mask,
                       (int)mask_words, (int)slot, num_rows, slot_ref(ring));
    return check_launch("history_advance_oldest_kernel");
}

extern "C" int lstep_ring_tick(int32_t* start, int32_t slots, void* stream) {
    if (!start || slots <= 0) return set_error(LSTEP_EINVAL, "lstep_ring_tick: bad arguments");
    hipLaunchKernelGGL(ring_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, start, (int)slots);
    return check_launch("ring_tick_kernel");
}
