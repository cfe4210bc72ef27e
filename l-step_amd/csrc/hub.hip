// Node channel of the gather stage for HUB nodes (round 5): models/LSTEP.py:177-211 for a node that occurs many times in one batch.
//
// The node channel of row b sums the node-feature rows of the last v = min(c, time_gap) interactions of node n_b before t_b
// (gather.hip).  On a power-law graph one node collects a fifth of a batch's endpoints: thousands of rows of ONE launch walk windows of
// 2000 slots over the SAME adjacency row, each shifted by a few interactions against the last -- 39 GB of algorithmic bytes per launch on
// the Zipf-1.2 variant of the 1 M-node workload, served from L2 at 21 TB/s, and the gather kernel 4 x its uniform time (round 4).
// Consecutive windows overlap in all but a few slots, so the algorithmic bytes themselves can go:
//
//   * the batch rows cat[src, dst] are already grouped by node (the engine's one group-by-key per batch); a node with >= min_occ
//     occurrences is cut into WORK ITEMS of up to 32 consecutive occurrences (lstep_hub_worklist; its rows are marked in `served`, and the
//     gather kernel skips their node channel);
//   * one workgroup per item (lstep_hub_node_sums): every occurrence j finds its window [lo_j, hi_j) of the adjacency row; the union
//     [R0, R1) of the item's windows -- time_gap + a few hundred slots -- is cut into eight contiguous pieces, one per wave; a wave adds the
//     node rows of its piece in order and keeps a copy of the running sum whenever it passes one of the 64 positions lo_j / hi_j;
//     with P[x] = (sum of the pieces before x's) + (that copy), the window sum of occurrence j is P[hi_j] - P[lo_j].
//
// ~2100 row reads per item instead of 32 x 2000.  Every sum has a fixed order (pieces in wave order, rows in adjacency order): the result
// is a function of the inputs alone.  It is NOT the summation order of the gather kernel (a window's rows are added as a difference of
// two prefixes): the node channel's output -- the window sum divided by valid x time_gap -- moves by ~1e-7 relative, far inside the
// 5e-5 bar (tests/test_hip_parity.py::test_hub_node_sums_vs_gather_kernel).
#include "lstep_common.h"

namespace lstep {

constexpr int kHubOcc = 32;        // occurrences per work item (their 2 x 32 window ends are one position per lane)
constexpr int kHubWaves = 8;       // waves per work item
constexpr int kHubInFlight = 16;   // node rows in flight per wave
constexpr int kHubRowVec = 44;     // float4 per row held in LDS: feature width <= 176 (51 KB of LDS per workgroup)

// seg_start[s] = first sorted position of segment s; seg_start[last segment + 1] = n2
__global__ void hub_seg_start_kernel(const int32_t* __restrict__ seg, int64_t n2, int32_t* __restrict__ seg_start) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const int32_t s = seg[i];
    if (i == 0 || seg[i - 1] != s) seg_start[s] = (int32_t)i;
    if (i == n2 - 1) seg_start[s + 1] = (int32_t)n2;
}

// segments of >= min_occ entries: mark their entries in `served`, emit one work item (first sorted position, occurrences) per 32 entries
__global__ void hub_items_kernel(const int32_t* __restrict__ seg, const int32_t* __restrict__ order, const int32_t* __restrict__ seg_start, int64_t n2,
                                 int32_t min_occ, uint8_t* __restrict__ served, int32_t* __restrict__ work, int32_t* __restrict__ nwork, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const int32_t s = seg[i];
    const int32_t st = seg_start[s], len = seg_start[s + 1] - st;
    if (len < min_occ) return;
    const int32_t off = (int32_t)i - st;
    served[order[i]] = 1;
    if (off % kHubOcc == 0) {
        const int32_t slot = atomicAdd(nwork, 1);      // (the ORDER of the items is arbitrary; every item is independent of the others)
        if (slot < cap) {                              // (cap >= lstep_hub_capacity(n2, min_occ) cannot overflow)
            work[2 * slot] = (int32_t)i;
            work[2 * slot + 1] = len - off < kHubOcc ? len - off : kHubOcc;
        }
    }
}

struct HubParams {
    lstep_csr_t csr;
    const float* node_raw;
    int F, G;
    const int64_t* ids;
    const double* times;
    const int32_t* order;
    const int32_t* work;
    const int32_t* nwork;
    float* out_node;
    int ld_node;
};

__global__ __launch_bounds__(kHubWaves* kWave) void hub_node_sums_kernel(HubParams p) {
    const int item = blockIdx.x;
    if (item >= *p.nwork) return;                      // (block-uniform: no barrier is skipped by part of a workgroup)
    __shared__ float4 snap[2 * kHubOcc][kHubRowVec];   // running sum of the owning wave's piece at each of the 64 positions
    __shared__ int snap_cnt[2 * kHubOcc];              // ... and the number of non-padding neighbours in it
    __shared__ float4 tot[kHubWaves][kHubRowVec];      // every wave's whole piece
    __shared__ int tot_cnt[kHubWaves];
    const int lane = lane_id(), wv = wave_in_block();
    const int F = p.F;
    const bool fa = lane < (F >> 2);
    const int pos0 = p.work[2 * item], m = p.work[2 * item + 1];
    // lane l: occurrence j = l & 31 (clamped); lanes 0..31 carry the window END hi_j, lanes 32..63 the window BEGIN lo_j
    const int j = (lane & (kHubOcc - 1)) < m ? (lane & (kHubOcc - 1)) : m - 1;
    const bool lane_live = (lane & (kHubOcc - 1)) < m;
    const int64_t e = p.order[pos0 + j];
    const int64_t node = p.ids[e];
    const bool in_range = node >= 0 && node < p.csr.num_rows;
    const double t = p.times[e];
    int64_t lo_n = 0, hi_n = 0;
    if (in_range) { lo_n = p.csr.indptr[node]; hi_n = p.csr.indptr[node + 1]; }
    int64_t a = lo_n, b = hi_n;                        // number of interactions strictly before t (np.searchsorted, side 'left': utils/utils.py:140)
    while (a < b) {
        const int64_t mid = a + ((b - a) >> 1);
        if (p.csr.ts[mid] < t) a = mid + 1; else b = mid;
    }
    const int64_t hi = a;
    const int64_t lo = hi - lo_n > p.G ? hi - p.G : lo_n;
    const int64_t q = lane < kHubOcc ? hi : lo;        // this lane's position
    // union of the item's windows (the same values in every lane after the butterflies)
    int64_t r0 = lo, r1 = hi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int64_t o0 = __shfl_xor(r0, off, kWave), o1 = __shfl_xor(r1, off, kWave);
        r0 = o0 < r0 ? o0 : r0;
        r1 = o1 > r1 ? o1 : r1;
    }
    const int64_t len = r1 - r0;
    const int64_t piece = (len + kHubWaves - 1) / kHubWaves;
    const int64_t begin = r0 + wv * piece;
    int64_t end = begin + piece;
    if (end > r1) end = r1;
    // a position x is kept by the wave whose piece holds row x - 1 (begin < x <= end); x == r0 needs no copy: P[r0] = 0
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int64_t c0 = begin; c0 < end; c0 += kWave) {
        const int mc = (int)((end - c0) < kWave ? (end - c0) : kWave);
        int idx = lane < mc ? p.csr.nbr[c0 + lane] : 0;
        idx = LSTEP_CHECKED(idx, p.csr.num_rows, kCheckGatherFwdNbrGap);
        settle(idx);
        for (int g0 = 0; g0 < mc; g0 += kHubInFlight) {
            int rid[kHubInFlight];
            float4 rn[kHubInFlight];
#pragma unroll
            for (int u = 0; u < kHubInFlight; ++u) {
                const int r = bcast_i32(idx, (g0 + u) < mc ? (g0 + u) : (mc - 1));
                rid[u] = (g0 + u) < mc ? r : -1;
                rn[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (fa) rn[u] = ld4(p.node_raw + (int64_t)(r > 0 ? r : 0) * F + lane * 4);
            }
#pragma unroll
            for (int u = 0; u < kHubInFlight; ++u) {
                if (rid[u] < 0) continue;                                  // wave-uniform (past the chunk)
                if (rid[u] > 0) {                                          // (neighbour id 0 = padding: not a valid slot, models/LSTEP.py:183-186)
                    acc.x += rn[u].x; acc.y += rn[u].y; acc.z += rn[u].z; acc.w += rn[u].w;
                    cnt += 1;
                }
                const int64_t cur = c0 + g0 + u + 1;                        // rows [begin, cur) are in acc now
                unsigned long long want = __ballot(lane_live && q == cur);
                while (want) {
                    const int l = __builtin_ctzll(want);
                    want &= want - 1;
                    if (fa) snap[l][lane] = acc;
                    if (lane == 0) snap_cnt[l] = cnt;
                }
            }
        }
    }
    if (fa) tot[wv][lane] = acc;
    if (lane == 0) tot_cnt[wv] = cnt;
    __syncthreads();
    // P[x] = pieces before x's, in wave order, + the owning wave's copy
    auto prefix_at = [&](int64_t x, int slot, float4& s, int& c) {
        s = make_float4(0.f, 0.f, 0.f, 0.f);
        c = 0;
        if (x <= r0 || piece == 0) return;
        const int w = (int)((x - 1 - r0) / piece);
        for (int w2 = 0; w2 < w; ++w2) {
            if (fa) { const float4 v = tot[w2][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            c += tot_cnt[w2];
        }
        if (fa) { const float4 v = snap[slot][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        c += snap_cnt[slot];
    };
    const float invG = 1.0f / (float)p.G;
    for (int jj = wv; jj < m; jj += kHubWaves) {
        const int64_t hi_j = bcast_i64(q, jj), lo_j = bcast_i64(q, kHubOcc + jj);
        float4 sh, sl;
        int ch, cl;
        prefix_at(hi_j, jj, sh, ch);
        prefix_at(lo_j, kHubOcc + jj, sl, cl);
        const int valid = ch - cl;
        const int64_t row = p.order[pos0 + jj];
        if (fa) {
            float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid > 0) {
                const float sc = (1.0f / (float)valid) / (float)p.G;       // softmax over the valid slots, then the mean over time_gap slots (LSTEP.py:186,208)
                res = make_float4((sh.x - sl.x) * sc, (sh.y - sl.y) * sc, (sh.z - sl.z) * sc, (sh.w - sl.w) * sc);
            } else {                                                       // all slots padded: uniform 1 / time_gap over copies of row 0, then / time_gap
                const float4 z = ld4(p.node_raw + lane * 4);
                res = make_float4(z.x * invG, z.y * invG, z.z * invG, z.w * invG);
            }
            float4 self = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in_range) self = ld4(p.node_raw + node * F + lane * 4);
            st4(p.out_node + row * (int64_t)p.ld_node + lane * 4, make_float4(res.x + self.x, res.y + self.y, res.z + self.z, res.w + self.w));
        }
        // padding columns [F, F rounded up to 16), clipped to the row stride (the gather kernel's zero_tail)
        const int zend = p.ld_node < ((F + 15) & ~15) ? p.ld_node : ((F + 15) & ~15);
        const int zc = F + lane * 4;
        if (zc < zend) st4(p.out_node + row * (int64_t)p.ld_node + zc, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

}  // namespace lstep

using namespace lstep;

// work items a batch of n2 grouped entries can produce: sum over the segments of >= min_occ entries of ceil(len / 32) <= n2 / 32 + n2 / min_occ
extern "C" int64_t lstep_hub_capacity(int64_t n2, int32_t min_occ) {
    if (min_occ < 2) min_occ = 2;
    return n2 <= 0 ? 1 : n2 / 32 + n2 / min_occ + 64;
}

extern "C" int lstep_hub_worklist(const int32_t* seg, const int32_t* order, int64_t n2, int32_t min_occ, int32_t* seg_start, uint8_t* served,
                                  int64_t num_rows, int32_t* work, int32_t* nwork, int64_t capacity, void* stream) {
    if (n2 < 0 || num_rows < n2 || min_occ < 2 || capacity < lstep_hub_capacity(n2, min_occ))
        return set_error(LSTEP_EINVAL, "lstep_hub_worklist: bad sizes (min_occ >= 2, capacity >= lstep_hub_capacity(n2, min_occ))");
    if (!served || !nwork || (n2 > 0 && (!seg || !order || !seg_start || !work))) return set_error(LSTEP_EINVAL, "lstep_hub_worklist: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(served, 0, (size_t)num_rows, s) != hipSuccess || hipMemsetAsync(nwork, 0, sizeof(int32_t), s) != hipSuccess)
        return set_error(LSTEP_EHIP, "lstep_hub_worklist: memset failed");
    if (n2 == 0) return LSTEP_OK;
    const unsigned grid = (unsigned)((n2 + 255) / 256);
    hipLaunchKernelGGL(hub_seg_start_kernel, dim3(grid), dim3(256), 0, s, seg, n2, seg_start);
    hipLaunchKernelGGL(hub_items_kernel, dim3(grid), dim3(256), 0, s, seg, order, (const int32_t*)seg_start, n2, min_occ, served, work, nwork, capacity);
    return check_launch("hub_items_kernel");
}

extern "C" int lstep_hub_node_sums(const lstep_csr_t* csr, const float* node_raw, int32_t feat_dim, const int64_t* node_ids, const double* times,
                                   int32_t time_gap, const int32_t* order, const int32_t* work, const int32_t* nwork, int64_t capacity, float* out_node,
                                   int32_t ld_node, void* stream) {
    if (capacity <= 0) return LSTEP_OK;
    if (feat_dim <= 0 || (feat_dim & 3) || feat_dim > 4 * kHubRowVec || time_gap <= 0)
        return set_error(LSTEP_EINVAL, "lstep_hub_node_sums: unsupported widths (feature width: a multiple of 4, <= 176)");
    if (!csr || !csr->indptr || !csr->nbr || !csr->ts || !node_raw || !node_ids || !times || !order || !work || !nwork || !out_node)
        return set_error(LSTEP_EINVAL, "lstep_hub_node_sums: NULL pointer");
    if (ld_node == 0) ld_node = feat_dim;
    if (ld_node < feat_dim || (ld_node & 3)) return set_error(LSTEP_EINVAL, "lstep_hub_node_sums: bad row stride");
    HubParams p{*csr, node_raw, feat_dim, time_gap, node_ids, times, order, work, nwork, out_node, ld_node};
    hipLaunchKernelGGL(hub_node_sums_kernel, dim3((unsigned)capacity), dim3(kHubWaves * kWave), 0, (hipStream_t)stream, p);
    return check_launch("hub_node_sums_kernel");
}
