// A + N + C gather stage of combining_pe_raw_feat, forward and backward.
//   reference: models/LSTEP.py:147-158 (edge rows + time features), :177-211 (node rows over time_gap neighbours),
//              :223-238 (PE rows + time features); sampler semantics utils/utils.py:129-146,199-208.
//
// Mapping: one 64-lane wave per destination row b.  A feature row is F floats = F/4 float4 (172 -> 43 lanes
// active, one dwordx4 per lane, the row is one contiguous 688-byte burst); several rows are kept in flight per
// wave.  The neighbour slice of the CSR is read once, 64 entries per wave-instruction, and broadcast lane ->
// scalar with v_readlane, so row addresses are scalar and the loads are base + lane*16.
// HBM-bound: algorithmic bytes per row = 4F*(k + 1 + valid_v) + 4P*(k + 1) + index bytes.
#include "lstep_common.h"

namespace lstep {

struct GatherParams {
    lstep_csr_t csr;
    const float* node_raw;
    const float* edge_raw;
    const float* pe;
    int F, P, D;
    const float* time_w;
    const float* time_b;
    const float* edge_agg_w;
    const int64_t* node_ids;
    const double* times;
    int64_t batch;
    int K, G;
    float* out_edge;
    float* out_node;
    float* out_pe;
    float* out_self;
    int32_t* out_count;
    int ld_edge, ld_node, ld_pe, ld_self;  // row strides (floats) of the four outputs; padding columns up to the next multiple of 16 are zeroed
    // explicit neighbourhoods (kExplicit kernels: RNG-defined sampling strategies, utils/utils.py:175-198, drawn on the host): the K slots
    // of every row for the edge / PE branch and the time_gap slots for the node branch, exactly as get_historical_neighbors returned them
    const int64_t* ex_nbr;    // [batch, K]
    const int64_t* ex_eid;    // [batch, K]  (edge branch)
    const float* ex_nt;       // [batch, K]  float32 neighbour times
    const int64_t* ex_nbr_g;  // [batch, G]  (node branch)
    const float* ex_nt_g;     // [batch, G]  (node branch with weighted_sum)
    int weighted_sum;         // models/LSTEP.py:190-206: node rows weighted by exp(-(t - neighbour time)), normalised over the row's distinct times
    const uint8_t* skip_node; // optional [batch]: rows whose node channel another kernel computes (hub nodes: csrc/hub.hip); long-row instantiation only
    int rows_per_block;       // long-row instantiation: batch rows a workgroup owns (4 = one per wave; 1 or 2 for small batches, see coop_rows_per_block)
};

// Long-row instantiation, small batches: with one row per WAVE, 600 rows are 150 workgroups on 256 CUs and every workgroup walks four
// 2000-slot node channels one after the other (73 -> 61 us at the Enron shape, round 4: latency-bound per CU).  With fewer rows per
// workgroup the same four waves share ONE row's walk and the grid covers the chip; a wave without a row of its own only helps.  The split
// of a long row over the waves (every fourth 64-slot chunk) and the order its partial sums are added in do not change: same bits.
static int coop_rows_per_block(int64_t batch) {
    static const int forced = [] { const char* e = getenv("LSTEP_GATHER_ROWS_PER_BLOCK"); return e ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    return batch <= 1024 ? 1 : (batch <= 2048 ? 2 : kWavesPerBlock);
}

// zero the padding columns of one output row: [width, width rounded up to 16), clipped to the row stride.  A stride wider than that
// (the row is a block of a concatenated operand, lstep_tail_fwd) leaves the rest of the row to its owner.
__device__ __forceinline__ void zero_tail(float* row, int width, int ld, int lane) {
    const int end = min(ld, (width + 15) & ~15);
    const int c = width + lane * 4;
    if (c < end) st4(row + c, make_float4(0.f, 0.f, 0.f, 0.f));
}

constexpr int kRowsInFlight = 8;      // edge rows + PE rows per group (2 x 8 loads in flight per wave)
constexpr int kNodeRowsInFlight = 8;  // node rows per group
constexpr int kCoopRowsInFlight = 16; // rows in flight per wave on a long row (129 registers: three waves per SIMD in this instantiation; measured better on the Zipf workload than 12 in flight at four waves: 4.17 vs 4.39 ms per step)
constexpr int kCoopMin = 256;         // node-channel rows longer than this are summed by the whole workgroup (LSTEP_GATHER_COOP_MIN build knob)

// kCoop: the instantiation for graphs / slot lists that CAN hold long node-channel rows (the host decides from lstep_csr_t.max_degree or the
// explicit list's time_gap).  It is a template parameter, not a launch-uniform branch: the shared-row path costs registers (16 rows in flight),
// scalar spills and LDS whether it runs or not, and the plain instantiation is the roofline kernel of the uniform workloads (0.43 ms per
// 49 152 rows; with the path compiled in: 0.48).
template <bool kEdgeNode, bool kPe, bool kExplicit = false, bool kCoop = false>
__global__ __launch_bounds__(kBlock) void gather_aggregate_fwd_kernel(GatherParams p) {
    const int lane = lane_id();
    const int wv = wave_in_block();
    const int64_t row = !kCoop ? (int64_t)blockIdx.x * kWavesPerBlock + wv
                               : (wv < p.rows_per_block ? (int64_t)blockIdx.x * p.rows_per_block + wv : p.batch);      // (p.batch: a helper wave)
    if constexpr (!kCoop) {
        if (row >= p.batch) return;
    }
    // (kCoop: no early return -- the node channel of a LONG row is shared by the workgroup's four waves below, through barriers every wave
    // must reach; a wave past the end of the batch works on row 0's inputs and writes nothing)
    const bool active = !kCoop || row < p.batch;
    const int F = p.F, P = p.P, D = p.D, K = p.K;
    const bool fa = lane < (F >> 2);  // lane owns a float4 of a feature row
    const bool pa = lane < (P >> 2);
    const int64_t node = active ? p.node_ids[row] : -1;
    const double t = active ? p.times[row] : 0.0;
    int64_t lo = 0, cnt = 0;
    const bool in_range = node >= 0 && node < p.csr.num_rows;
    if (in_range && !kExplicit) {
        lo = p.csr.indptr[node];
        cnt = wave_count_before(p.csr.ts, lo, p.csr.indptr[node + 1], t, lane);
    }
    // explicit lists: every slot is given (padding slots carry neighbour id 0 and gather row 0 like any other id)
    const int k = !active ? 0 : (kExplicit ? K : (int)(cnt < K ? cnt : K));
    const int npad = active ? K - k : 0;
    const int64_t kfirst = lo + cnt - k;

    const float w0 = lane < D ? p.time_w[lane] : 0.0f, b0 = lane < D ? p.time_b[lane] : 0.0f;
    const float w1 = lane + kWave < D ? p.time_w[lane + kWave] : 0.0f, b1 = lane + kWave < D ? p.time_b[lane + kWave] : 0.0f;

    float4 accE = make_float4(0.f, 0.f, 0.f, 0.f), accP = accE;
    float xt0 = 0.f, xt1 = 0.f, pt0 = 0.f, pt1 = 0.f;

    for (int c0 = 0; c0 < k; c0 += kWave) {
        const int m = (k - c0) < kWave ? (k - c0) : kWave;
        int nb = 0, ed = 0;
        float dt = 0.f, aw = 0.f;
        if (lane < m) {
            if (kExplicit) {
                const int64_t e = row * K + c0 + lane;
                nb = (int)p.ex_nbr[e];
                if (kEdgeNode) ed = (int)p.ex_eid[e];
                dt = (float)(t - (double)p.ex_nt[e]);
            } else {
                const int64_t e = kfirst + c0 + lane;
                nb = p.csr.nbr[e];
                ed = p.csr.eid[e];
                dt = delta_t(t, p.csr.ts[e]);
            }
            nb = LSTEP_CHECKED(nb, p.csr.num_rows, kCheckGatherFwdNbr);       // (checked builds only: lstep_common.h)
            ed = LSTEP_CHECKED(ed, LSTEP_EDGE_ROWS(), kCheckGatherFwdEdge);
            if (kEdgeNode) aw = p.edge_agg_w[npad + c0 + lane];
        }
        // Settle the slot metadata BEFORE the row loop: otherwise hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front
        // of every v_readlane of these registers inside the loop and the row loads of a group serialise.
        settle(nb ^ ed ^ __float_as_int(dt) ^ __float_as_int(aw));
        for (int j = 0; j < m; j += kRowsInFlight) {
            int64_t ej[kRowsInFlight], nj[kRowsInFlight];
            float wj[kRowsInFlight], lj[kRowsInFlight];
#pragma unroll
            for (int u = 0; u < kRowsInFlight; ++u) {  // tail slots re-read the last live row with weight 0 (cache hit, no branch)
                const bool live = (j + u) < m;
                const int jj = live ? (j + u) : (m - 1);
                ej[u] = bcast_i32(ed, jj);
                nj[u] = bcast_i32(nb, jj);
                wj[u] = live ? bcast_f32(aw, jj) : 0.f;
                lj[u] = live ? 1.f : 0.f;
            }
            float4 re[kRowsInFlight], rp[kRowsInFlight];
            if (kEdgeNode && fa) {
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u) re[u] = ld4_stream(p.edge_raw + ej[u] * F + lane * 4);
            }
            if (kPe && pa) {
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u) rp[u] = ld4(p.pe + nj[u] * P + lane * 4);
            }
            // time features of this group's slots while its rows are in flight: cos(dt * w_d), zero where the
            // neighbour id is 0 (LSTEP.py:154,231)
            const int jend = (j + kRowsInFlight) < m ? (j + kRowsInFlight) : m;
            for (int jj = j; jj < jend; ++jj) {
                if (bcast_i32(nb, jj) == 0) continue;
                const float dj = bcast_f32(dt, jj);
                const float aj = bcast_f32(aw, jj);
                const float c0v = lane < D ? time_feat(dj, w0, b0) : 0.f;
                const float c1v = lane + kWave < D ? time_feat(dj, w1, b1) : 0.f;
                xt0 = fmaf(aj, c0v, xt0);
                xt1 = fmaf(aj, c1v, xt1);
                pt0 += c0v;
                pt1 += c1v;
            }
            if (kEdgeNode && fa) {
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u) fma4(accE, wj[u], re[u]);
            }
            if (kPe && pa) {
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u) fma4(accP, lj[u], rp[u]);
            }
        }
    }
    if (npad > 0) {  // padding slots gather row 0 of the edge table and of the PE table (pe[0] is live)
        if (kEdgeNode) {
            float s = 0.f;
            for (int i = lane; i < npad; i += kWave) s += p.edge_agg_w[i];
            s = wave_sum(s);
            if (fa) fma4(accE, s, ld4(p.edge_raw + lane * 4));
        }
        if (kPe && pa) fma4(accP, (float)npad, ld4(p.pe + lane * 4));
    }

    if (kEdgeNode) {
        if (active) {
            float* oe = p.out_edge + row * (int64_t)p.ld_edge;
            if (lane < D) oe[lane] = xt0;
            if (lane + kWave < D) oe[lane + kWave] = xt1;
            if (fa) st4(oe + D + lane * 4, accE);
            zero_tail(oe, D + F, p.ld_edge, lane);
        }

        // node channel: the last v = min(cnt, G) interactions; score 1/valid on ids > 0, then mean over G slots
        // (rows served by lstep_hub_node_sums: no node channel here, and the node output row is theirs)
        const bool skip = kCoop && p.skip_node != nullptr && active && p.skip_node[row] != 0;
        const int64_t v_all = (!active || skip) ? 0 : (kExplicit ? (int64_t)p.G : (cnt < p.G ? cnt : p.G));
        const int64_t vfirst = kExplicit ? row * (int64_t)p.G : lo + cnt - v_all;       // first slot: CSR position / position in the explicit list
        // LONG rows (more than kCoopMin slots: hub nodes of a power-law graph, every row of the reference's small dense datasets, every row of
        // an explicit time_gap-slot list) are summed by the whole workgroup: one wave walks a 2000-slot row in 250 dependent rounds of 8 row
        // loads while the other three idle -- 73 us for the 600 rows of an Enron-shaped batch, 2.2 ms of the Zipf-1.2 c4 step (round 4).  Each
        // wave takes every fourth 64-slot chunk; the four partial sums meet in LDS and are added in wave order (a fixed order: the result is a
        // function of the inputs).  Short rows stay with their own wave, as before.
        __shared__ long long sh_first[kCoop ? kWavesPerBlock : 1];
        __shared__ int sh_v[kCoop ? kWavesPerBlock : 1];
        __shared__ float4 sh_acc[kCoop ? kWavesPerBlock : 1][kCoop ? kMaxRowVec : 1];
        __shared__ int sh_valid[kCoop ? kWavesPerBlock : 1];
        constexpr bool coop_on = kCoop;
        const bool coop_row = coop_on && v_all > kCoopMin;
        if constexpr (coop_on) {
            if (lane == 0) {
                sh_v[wv] = coop_row ? (int)v_all : 0;
                sh_first[wv] = vfirst;
            }
            __syncthreads();
        }
        const int64_t v = coop_row ? 0 : v_all;          // what this wave sums alone
        // weighted_sum (models/LSTEP.py:190-206): every slot's node row is also scaled by w = clamp(e(time) / sum of e over the row's
        // DISTINCT non-zero neighbour times, 0, 1), e(x) = exp(-(t - x)) in float64; x is what scatter_mean makes of the slot's float32 time
        // (the sequential float32 sum of the c slots that share it, divided by c).  First pass: the denominator.
        double wden = 1.0;
        if (p.weighted_sum) {
            double part = 0.0;
            for (int64_t c0 = 0; c0 < v; c0 += kWave) {
                const int64_t s = c0 + lane;
                if (s < v) {
                    const float ts = kExplicit ? p.ex_nt_g[row * (int64_t)p.G + s] : (float)p.csr.ts[vfirst + s];
                    // one term per distinct time: counted at the FIRST slot of its run (slots are time-sorted in both sampling modes)
                    const float prev = s > 0 ? (kExplicit ? p.ex_nt_g[row * (int64_t)p.G + s - 1] : (float)p.csr.ts[vfirst + s - 1]) : 0.f;
                    if ((s == 0 || prev != ts) && ts != 0.f) {
                        int c = 1;
                        while (s + c < v && (kExplicit ? p.ex_nt_g[row * (int64_t)p.G + s + c] : (float)p.csr.ts[vfirst + s + c]) == ts) ++c;
                        float sum = 0.f;
                        for (int i = 0; i < c; ++i) sum += ts;
                        const float mean = sum / (float)c;
                        if (mean != 0.f) part += exp(-(t - (double)mean));
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, kWave);
            wden = part + (part == 0.0 ? 1.0 : 0.0);
        }
        float4 accN = make_float4(0.f, 0.f, 0.f, 0.f);
        int valid = 0;
        for (int64_t c0 = 0; c0 < v; c0 += kWave) {
            const int m = (int)((v - c0) < kWave ? (v - c0) : kWave);
            int idx = lane < m ? (kExplicit ? (int)p.ex_nbr_g[row * (int64_t)p.G + c0 + lane] : p.csr.nbr[vfirst + c0 + lane]) : 0;
            idx = LSTEP_CHECKED(idx, p.csr.num_rows, kCheckGatherFwdNbrGap);
            float wslot = 1.f;
            if (p.weighted_sum && lane < m) {
                const int64_t s = c0 + lane;
                const float ts = kExplicit ? p.ex_nt_g[row * (int64_t)p.G + s] : (float)p.csr.ts[vfirst + s];
                int64_t a = s, b = s;       // the run of slots that share this time
                while (a > 0 && (kExplicit ? p.ex_nt_g[row * (int64_t)p.G + a - 1] : (float)p.csr.ts[vfirst + a - 1]) == ts) --a;
                while (b + 1 < v && (kExplicit ? p.ex_nt_g[row * (int64_t)p.G + b + 1] : (float)p.csr.ts[vfirst + b + 1]) == ts) ++b;
                float sum = 0.f;
                for (int64_t i = a; i <= b; ++i) sum += ts;
                const float mean = sum / (float)(b - a + 1);
                double w = mean != 0.f ? exp(-(t - (double)mean)) / wden : 0.0;
                w = w < 0.0 ? 0.0 : (w > 1.0 ? 1.0 : w);
                wslot = (float)w;
            }
            valid += __popcll(__ballot(idx > 0));
            settle(idx ^ __float_as_int(wslot));
            for (int j = 0; j < m; j += kNodeRowsInFlight) {
                int64_t nj[kNodeRowsInFlight];
                float lj[kNodeRowsInFlight];
#pragma unroll
                for (int u = 0; u < kNodeRowsInFlight; ++u) {
                    const bool live = (j + u) < m;
                    const int r = bcast_i32(idx, live ? (j + u) : (m - 1));
                    nj[u] = r > 0 ? r : 0;
                    lj[u] = (live && r > 0) ? bcast_f32(wslot, live ? (j + u) : (m - 1)) : 0.f;
                }
                if (fa) {
                    float4 rn[kNodeRowsInFlight];
#pragma unroll
                    for (int u = 0; u < kNodeRowsInFlight; ++u) rn[u] = ld4(p.node_raw + nj[u] * F + lane * 4);
#pragma unroll
                    for (int u = 0; u < kNodeRowsInFlight; ++u) fma4(accN, lj[u], rn[u]);
                }
            }
        }
        if constexpr (coop_on)
        for (int r = 0; r < kWavesPerBlock; ++r) {
            const int vr = sh_v[r];
            if (vr == 0) continue;                       // (block-uniform)
            const int64_t first = sh_first[r];
            float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
            int pvalid = 0;
            for (int64_t c0 = (int64_t)wv * kWave; c0 < vr; c0 += (int64_t)kWave * kWavesPerBlock) {
                const int m = (int)((vr - c0) < kWave ? (vr - c0) : kWave);
                int idx = lane < m ? (kExplicit ? (int)p.ex_nbr_g[first + c0 + lane] : p.csr.nbr[first + c0 + lane]) : 0;
                idx = LSTEP_CHECKED(idx, p.csr.num_rows, kCheckGatherFwdNbrGap);
                pvalid += __popcll(__ballot(idx > 0));
                settle(idx);
                for (int j = 0; j < m; j += kCoopRowsInFlight) {
                    int64_t nj[kCoopRowsInFlight];
                    float lj[kCoopRowsInFlight];
#pragma unroll
                    for (int u = 0; u < kCoopRowsInFlight; ++u) {
                        const bool live = (j + u) < m;
                        const int rr = bcast_i32(idx, live ? (j + u) : (m - 1));
                        nj[u] = rr > 0 ? rr : 0;
                        lj[u] = (live && rr > 0) ? 1.f : 0.f;
                    }
                    if (fa) {
                        float4 rn[kCoopRowsInFlight];
#pragma unroll
                        for (int u = 0; u < kCoopRowsInFlight; ++u) rn[u] = ld4(p.node_raw + nj[u] * F + lane * 4);
#pragma unroll
                        for (int u = 0; u < kCoopRowsInFlight; ++u) fma4(part, lj[u], rn[u]);
                    }
                }
            }
            if (fa) sh_acc[wv][lane] = part;
            if (lane == 0) sh_valid[wv] = pvalid;
            __syncthreads();
            if (wv == r) {
                if (fa) {
#pragma unroll
                    for (int w2 = 0; w2 < kWavesPerBlock; ++w2) {
                        const float4 q = sh_acc[w2][lane];
                        accN.x += q.x; accN.y += q.y; accN.z += q.z; accN.w += q.w;
                    }
                }
#pragma unroll
                for (int w2 = 0; w2 < kWavesPerBlock; ++w2) valid += sh_valid[w2];
            }
            __syncthreads();
        }
        if (fa && active && !skip) {
            const float invG = 1.0f / (float)p.G;
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid > 0) {
                const float s = (1.0f / (float)valid) / (float)p.G;
                r0 = make_float4(accN.x * s, accN.y * s, accN.z * s, accN.w * s);
            } else if (!p.weighted_sum) {  // all slots padded: softmax is uniform 1/G over G copies of row 0, then /G again
                const float4 z = ld4(p.node_raw + lane * 4);
                r0 = make_float4(z.x * invG, z.y * invG, z.z * invG, z.w * invG);
            }    // (weighted_sum: every slot's time is 0 -> every weight is 0)
            float4 self = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in_range) self = ld4(p.node_raw + node * F + lane * 4);
            st4(p.out_node + row * (int64_t)p.ld_node + lane * 4, make_float4(r0.x + self.x, r0.y + self.y, r0.z + self.z, r0.w + self.w));
        }
        if (active && !skip) zero_tail(p.out_node + row * (int64_t)p.ld_node, F, p.ld_node, lane);
    }
    if (kPe && active) {
        float* op = p.out_pe + row * (int64_t)p.ld_pe;
        if (pa) st4(op + lane * 4, accP);
        if (lane < D) op[P + lane] = pt0;
        if (lane + kWave < D) op[P + lane + kWave] = pt1;
        if (pa) st4(p.out_self + row * (int64_t)p.ld_self + lane * 4, in_range ? ld4(p.pe + node * P + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
        zero_tail(op, P + D, p.ld_pe, lane);
        zero_tail(p.out_self + row * (int64_t)p.ld_self, P, p.ld_self, lane);
    }
    if (active && p.out_count != nullptr && lane == 0) p.out_count[row] = kExplicit ? K : (int32_t)cnt;
}

struct GatherBwdParams {
    lstep_csr_t csr;
    const float* edge_raw;
    int F, P, D;
    const float* time_w;
    const float* time_b;
    const int64_t* node_ids;
    const double* times;
    const int32_t* count;
    int64_t batch;
    int K;
    const float* grad_edge;
    const float* grad_pe_agg;
    const float* grad_self;
    const int32_t* slot_of;
    float* out_slot_dot;
    float* grad_pe_rows;
    int32_t* out_hits;  // [B, K]: spliced-row index of every slot's neighbour (or -1); when set, no atomics are issued
    int ld_edge, ld_pe, ld_self;  // row strides (floats) of grad_edge / grad_pe_agg / grad_self
    const int64_t* ex_nbr;        // explicit neighbourhoods (kExplicit): [batch, K] ids, edge ids, float32 times of the slots
    const int64_t* ex_eid;
    const float* ex_nt;
};

// atomically add a P-wide gradient row held as g[i] = elements lane + 64*i (contiguous dwords per wave-instruction)
__device__ __forceinline__ void atomic_add_row(float* dst, const float g[4], int P, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * kWave;
        if (c < P) atomicAdd(dst + c, g[i]);
    }
}

constexpr int kBwdRows = 8;      // edge rows in flight per wave in the backward kernel (power of two <= 32)

template <bool kExplicit = false>
__global__ __launch_bounds__(kBlock) void gather_aggregate_bwd_kernel(GatherBwdParams p) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (row >= p.batch) return;
    const int F = p.F, P = p.P, D = p.D, K = p.K;
    const bool fa = lane < (F >> 2);
    const int64_t node = p.node_ids[row];
    const bool in_range = node >= 0 && node < p.csr.num_rows;
    const double t = p.times[row];
    const int64_t cnt = (in_range && !kExplicit) ? p.count[row] : 0;
    const int64_t lo = (in_range && !kExplicit) ? p.csr.indptr[node] : 0;
    const int k = kExplicit ? K : (int)(cnt < K ? cnt : K);
    const int npad = K - k;
    const int64_t kfirst = lo + cnt - k;
    const bool do_edge = p.grad_edge != nullptr && p.out_slot_dot != nullptr;
    const bool do_pe = p.grad_pe_agg != nullptr && p.grad_pe_rows != nullptr && p.out_hits == nullptr;

    float gt0 = 0.f, gt1 = 0.f, w0 = 0.f, b0 = 0.f, w1 = 0.f, b1 = 0.f;
    float4 gE = make_float4(0.f, 0.f, 0.f, 0.f);
    if (do_edge) {
        const float* ge = p.grad_edge + row * (int64_t)p.ld_edge;
        if (lane < D) { gt0 = ge[lane]; w0 = p.time_w[lane]; b0 = p.time_b[lane]; }
        if (lane + kWave < D) { gt1 = ge[lane + kWave]; w1 = p.time_w[lane + kWave]; b1 = p.time_b[lane + kWave]; }
        if (fa) gE = ld4(ge + D + lane * 4);
    }
    float gp[4] = {0.f, 0.f, 0.f, 0.f};
    if (do_pe) {
        const float* g = p.grad_pe_agg + row * (int64_t)p.ld_pe;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + i * kWave < P) gp[i] = g[lane + i * kWave];
    }

    for (int c0 = 0; c0 < k; c0 += kWave) {
        const int m = (k - c0) < kWave ? (k - c0) : kWave;
        int nb = 0, ed = 0;
        float dt = 0.f;
        if (lane < m) {
            if (kExplicit) {
                const int64_t e = row * K + c0 + lane;
                nb = (int)p.ex_nbr[e];
                ed = p.ex_eid ? (int)p.ex_eid[e] : 0;
                dt = (float)(t - (double)p.ex_nt[e]);
            } else {
                const int64_t e = kfirst + c0 + lane;
                nb = p.csr.nbr[e];
                ed = p.csr.eid[e];
                dt = delta_t(t, p.csr.ts[e]);
            }
            nb = LSTEP_CHECKED(nb, p.csr.num_rows, kCheckGatherBwdNbr);       // (checked builds only: lstep_common.h)
            ed = LSTEP_CHECKED(ed, LSTEP_EDGE_ROWS(), kCheckGatherBwdEdge);
        }
        if (do_edge) {
            // Settled metadata, kBwdRows edge rows in flight, the slots' time-feature dots computed while they travel, and ONE butterfly
            // reduce-scatter per group: at offsets 32 / 16 / 8 a lane hands the half of the partial sums its partner keeps across and
            // adds what it receives (4 + 2 + 1 exchanges), offsets 4 / 2 / 1 finish the one sum left (3 more): 10 exchanges for 8 slots
            // instead of 48, and the first seven are independent of each other within a level.
            settle(nb ^ ed ^ __float_as_int(dt));
            float mine = 0.f;  // lane j keeps the dot product of slot c0 + j
            for (int j = 0; j < m; j += kBwdRows) {
                int64_t ej[kBwdRows];
                float4 re[kBwdRows];
#pragma unroll
                for (int u = 0; u < kBwdRows; ++u) {     // tail slots re-read the last live row (cache hit, no branch); their sums are dropped
                    ej[u] = bcast_i32(ed, (j + u) < m ? (j + u) : (m - 1));
                }
                if (fa) {
#pragma unroll
                    for (int u = 0; u < kBwdRows; ++u) re[u] = ld4_stream(p.edge_raw + ej[u] * F + lane * 4);
                }
                float part[kBwdRows];
#pragma unroll
                for (int u = 0; u < kBwdRows; ++u) {
                    part[u] = 0.f;
                    if ((j + u) < m && bcast_i32(nb, j + u) != 0) {     // wave-uniform
                        const float dj = bcast_f32(dt, j + u);
                        // (cos_hw: these dots only feed d(edge_agg.weight), and the kernel is bound by the cosines' issue slots)
                        part[u] = gt0 * time_feat_grad(dj, w0, b0);     // (lanes past D hold gt = 0)
                        part[u] = fmaf(gt1, time_feat_grad(dj, w1, b1), part[u]);
                    }
                }
                if (fa) {      // (same predicate as the loads: the rows need no merge value)
#pragma unroll
                    for (int u = 0; u < kBwdRows; ++u) {
                        // opaque to the optimiser up to here: otherwise it re-pairs the components of neighbouring rows for packed
                        // multiplies with copies placed right behind the loads, and waits for every row before the cosines start
                        asm volatile("" : "+v"(re[u].x), "+v"(re[u].y), "+v"(re[u].z), "+v"(re[u].w));
                        part[u] += dot4(gE, re[u]);
                    }
                }
#pragma unroll
                for (int half = kBwdRows / 2, off = 32; half >= 1; half >>= 1, off >>= 1) {
                    const bool upper = (lane & off) != 0;
#pragma unroll
                    for (int i = 0; i < half; ++i) {
                        const float send = upper ? part[i] : part[i + half];
                        const float keep = upper ? part[i + half] : part[i];
                        part[i] = keep + __shfl_xor(send, off, kWave);
                    }
                }
                float v = part[0];
#pragma unroll
                for (int off = 64 / kBwdRows / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
                // slot u of the group ended up in the lanes whose bits 5, 4, 3 spell u (bit 5 picked the upper half first); lane j + u
                // fetches it, and the chunk's dots leave with ONE store after the loop (a store per group would make the next group's
                // loads wait for it: stores and loads share the vmcnt counter)
                int src = 0;
#pragma unroll
                for (int half = kBwdRows / 2, off = 32; half >= 1; half >>= 1, off >>= 1) src += ((lane & (kBwdRows - 1)) & half) ? off : 0;
                const float got = __shfl(v, src, kWave);
                if (lane >= j && lane < j + kBwdRows) mine = got;
            }
            if (lane < m) p.out_slot_dot[row * (int64_t)K + npad + c0 + lane] = mine;
        }
        if (p.out_hits != nullptr && lane < m) p.out_hits[row * (int64_t)K + npad + c0 + lane] = p.slot_of[nb];
        if (do_pe) {
            for (int j = 0; j < m; ++j) {
                const int64_t nj = bcast_i32(nb, j);
                if (p.slot_of != nullptr) {
                    const int u = p.slot_of[nj];
                    if (u >= 0) atomic_add_row(p.grad_pe_rows + (int64_t)u * P, gp, P, lane);
                } else {
                    atomic_add_row(p.grad_pe_rows + nj * P, gp, P, lane);
                }
            }
        }
    }
    if (p.out_hits != nullptr)
        for (int s2 = lane; s2 < npad; s2 += kWave) p.out_hits[row * (int64_t)K + s2] = -1;
    if (do_edge && npad > 0) {  // padding slots: time features are zero, edge row is row 0
        float part = fa ? dot4(gE, ld4(p.edge_raw + lane * 4)) : 0.f;
        part = wave_sum(part);
        for (int s = lane; s < npad; s += kWave) p.out_slot_dot[row * (int64_t)K + s] = part;
    }
    if (p.grad_self != nullptr && p.grad_pe_rows != nullptr && p.out_hits == nullptr && in_range) {
        float gs[4] = {0.f, 0.f, 0.f, 0.f};
        const float* g = p.grad_self + row * (int64_t)p.ld_self;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + i * kWave < P) gs[i] = g[lane + i * kWave];
        if (p.slot_of != nullptr) {
            const int u = p.slot_of[node];
            if (u >= 0) atomic_add_row(p.grad_pe_rows + (int64_t)u * P, gs, P, lane);
        } else {
            atomic_add_row(p.grad_pe_rows + node * P, gs, P, lane);
        }
    }
}

static int check_dims(const char* who, int F, int P, int D) {
    if (F <= 0 || P <= 0 || D <= 0 || (F & 3) || (P & 3) || (D & 3) || F > 4 * kMaxRowVec || P > 4 * kMaxRowVec || D > kMaxTimeDim)
        return set_error(LSTEP_EINVAL, "%s: unsupported widths F=%d P=%d D=%d (need multiples of 4, F,P<=256, D<=128)", who, F, P, D);
    return LSTEP_OK;
}

static int check_ld(const char* who, int a, int wa, int b, int wb, int c, int wc, int d, int wd) {
    const int ld[4] = {a, b, c, d}, w[4] = {wa, wb, wc, wd};
    for (int i = 0; i < 4; ++i)
        if (ld[i] < w[i] || (ld[i] & 3))
            return set_error(LSTEP_EINVAL, "%s: row stride %d invalid for width %d (need a multiple of 4, >= width)", who, ld[i], w[i]);
    return LSTEP_OK;
}

}  // namespace lstep

using namespace lstep;

static int gather_aggregate_fwd_impl(const lstep_csr_t* csr, const float* node_raw, const float* edge_raw, const float* pe,
                                     int32_t feat_dim, int32_t pe_dim, const float* time_w, const float* time_b,
                                     int32_t time_dim, const float* edge_agg_w, const int64_t* node_ids,
                                     const double* times, int64_t batch, int32_t num_neighbors, int32_t time_gap,
                                     uint32_t branches, float* out_edge, float* out_node, float* out_pe, float* out_self,
                                     int32_t ld_edge, int32_t ld_node, int32_t ld_pe, int32_t ld_self, int32_t* out_count,
                                     const uint8_t* skip_node, void* stream) {
    if (num_neighbors <= 0 || time_gap <= 0)
        return set_error(LSTEP_EINVAL, "Number of sampled neighbors for each node should be greater than 0!");
    if (int rc = check_dims("lstep_gather_aggregate_fwd", feat_dim, pe_dim, time_dim)) return rc;
    const bool en = branches & LSTEP_BRANCH_EDGE_NODE, pb = branches & LSTEP_BRANCH_PE;
    if (!en && !pb) return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_fwd: no branch selected");
    if (batch < 0) return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_fwd: negative batch");
    if (batch == 0) return LSTEP_OK;
    if (!csr || !csr->indptr || !csr->nbr || !csr->eid || !csr->ts || !time_w || !time_b || !node_ids || !times)
        return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_fwd: NULL pointer");
    if (en && (!node_raw || !edge_raw || !edge_agg_w || !out_edge || !out_node))
        return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_fwd: edge/node branch needs node_raw, edge_raw, edge_agg_w, out_edge, out_node");
    if (pb && (!pe || !out_pe || !out_self))
        return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_fwd: PE branch needs pe, out_pe, out_self");
    if (ld_edge == 0) ld_edge = time_dim + feat_dim;
    if (ld_node == 0) ld_node = feat_dim;
    if (ld_pe == 0) ld_pe = pe_dim + time_dim;
    if (ld_self == 0) ld_self = pe_dim;
    if (int rc = check_ld("lstep_gather_aggregate_fwd", ld_edge, time_dim + feat_dim, ld_node, feat_dim, ld_pe, pe_dim + time_dim, ld_self, pe_dim)) return rc;
    GatherParams p{*csr, node_raw, edge_raw, pe, feat_dim, pe_dim, time_dim, time_w, time_b, edge_agg_w, node_ids, times,
                   batch, num_neighbors, time_gap, out_edge, out_node, out_pe, out_self, out_count, ld_edge, ld_node, ld_pe, ld_self,
                   nullptr, nullptr, nullptr, nullptr, nullptr, (branches & LSTEP_WEIGHTED_SUM) ? 1 : 0, skip_node, kWavesPerBlock};
    hipStream_t s = (hipStream_t)stream;
    // long node-channel rows possible?  (max_degree 0 = unknown: assume yes)
    const bool coop = en && !(branches & LSTEP_WEIGHTED_SUM) && (csr->max_degree == 0 || csr->max_degree > kCoopMin) && time_gap > kCoopMin;
    if (coop) p.rows_per_block = coop_rows_per_block(batch);
    const dim3 grid((unsigned)((batch + p.rows_per_block - 1) / p.rows_per_block)), block(kBlock);
    if (skip_node && !coop) return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_fwd_skip: skip_node needs the long-row form (max_degree and time_gap > 256, no weighted_sum)");
    if (en && pb) {
        if (coop) hipLaunchKernelGGL((gather_aggregate_fwd_kernel<true, true, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gather_aggregate_fwd_kernel<true, true>), grid, block, 0, s, p);
    } else if (en) {
        if (coop) hipLaunchKernelGGL((gather_aggregate_fwd_kernel<true, false, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gather_aggregate_fwd_kernel<true, false>), grid, block, 0, s, p);
    } else hipLaunchKernelGGL((gather_aggregate_fwd_kernel<false, true>), grid, block, 0, s, p);
    return check_launch("gather_aggregate_fwd_kernel");
}

extern "C" int lstep_gather_aggregate_fwd(const lstep_csr_t* csr, const float* node_raw, const float* edge_raw, const float* pe,
                                          int32_t feat_dim, int32_t pe_dim, const float* time_w, const float* time_b,
                                          int32_t time_dim, const float* edge_agg_w, const int64_t* node_ids,
                                          const double* times, int64_t batch, int32_t num_neighbors, int32_t time_gap,
                                          uint32_t branches, float* out_edge, float* out_node, float* out_pe, float* out_self,
                                          int32_t ld_edge, int32_t ld_node, int32_t ld_pe, int32_t ld_self, int32_t* out_count,
                                          void* stream) {
    return gather_aggregate_fwd_impl(csr, node_raw, edge_raw, pe, feat_dim, pe_dim, time_w, time_b, time_dim, edge_agg_w, node_ids, times, batch,
                                     num_neighbors, time_gap, branches, out_edge, out_node, out_pe, out_self, ld_edge, ld_node, ld_pe, ld_self,
                                     out_count, nullptr, stream);
}

// The same launch with rows whose node channel is computed elsewhere (hub nodes: lstep_hub_worklist / lstep_hub_node_sums, csrc/hub.hip):
// skip_node uint8 [batch], non-zero = the row's node channel is skipped and its out_node row is NOT written.
extern "C" int lstep_gather_aggregate_fwd_skip(const lstep_csr_t* csr, const float* node_raw, const float* edge_raw, const float* pe,
                                               int32_t feat_dim, int32_t pe_dim, const float* time_w, const float* time_b,
                                               int32_t time_dim, const float* edge_agg_w, const int64_t* node_ids,
                                               const double* times, int64_t batch, int32_t num_neighbors, int32_t time_gap,
                                               uint32_t branches, float* out_edge, float* out_node, float* out_pe, float* out_self,
                                               int32_t ld_edge, int32_t ld_node, int32_t ld_pe, int32_t ld_self, int32_t* out_count,
                                               const uint8_t* skip_node, void* stream) {
    return gather_aggregate_fwd_impl(csr, node_raw, edge_raw, pe, feat_dim, pe_dim, time_w, time_b, time_dim, edge_agg_w, node_ids, times, batch,
                                     num_neighbors, time_gap, branches, out_edge, out_node, out_pe, out_self, ld_edge, ld_node, ld_pe, ld_self,
                                     out_count, skip_node, stream);
}

// The gather stage on EXPLICIT neighbourhoods: one branch per call, the slots exactly as a sampler returned them (the RNG-defined
// strategies of utils/utils.py:175-198 draw three independent neighbourhoods per combining_pe_raw_feat call: K slots for the edge
// channel, time_gap slots for the node channel, K slots for the PE channel -- models/LSTEP.py:147,177,223).
extern "C" int lstep_gather_explicit_fwd(const float* node_raw, const float* edge_raw, const float* pe, int32_t feat_dim, int32_t pe_dim,
                                         const float* time_w, const float* time_b, int32_t time_dim, const float* edge_agg_w,
                                         const int64_t* node_ids, const double* times, int64_t batch, int32_t num_neighbors, int32_t time_gap,
                                         uint32_t branches, const int64_t* nbr, const int64_t* eid, const float* nt, const int64_t* nbr_gap,
                                         const float* nt_gap, int64_t num_rows, float* out_edge, float* out_node, float* out_pe,
                                         float* out_self, int32_t ld_edge, int32_t ld_node, int32_t ld_pe, int32_t ld_self, void* stream) {
    if (num_neighbors <= 0 || time_gap <= 0)
        return set_error(LSTEP_EINVAL, "Number of sampled neighbors for each node should be greater than 0!");
    if (int rc = check_dims("lstep_gather_explicit_fwd", feat_dim, pe_dim, time_dim)) return rc;
    const bool en = branches & LSTEP_BRANCH_EDGE_NODE, pb = branches & LSTEP_BRANCH_PE;
    if (en == pb) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_fwd: exactly one branch per call (their neighbourhoods differ)");
    if (batch < 0 || num_rows <= 0) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_fwd: bad sizes");
    if (batch == 0) return LSTEP_OK;
    if (!time_w || !time_b || !node_ids || !times || !nbr || !nt) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_fwd: NULL pointer");
    if (en && (!node_raw || !edge_raw || !edge_agg_w || !out_edge || !out_node || !eid || !nbr_gap || ((branches & LSTEP_WEIGHTED_SUM) && !nt_gap)))
        return set_error(LSTEP_EINVAL, "lstep_gather_explicit_fwd: edge/node branch needs tables, edge ids, the time_gap list and both outputs");
    if (pb && (!pe || !out_pe || !out_self)) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_fwd: PE branch needs pe, out_pe, out_self");
    if (ld_edge == 0) ld_edge = time_dim + feat_dim;
    if (ld_node == 0) ld_node = feat_dim;
    if (ld_pe == 0) ld_pe = pe_dim + time_dim;
    if (ld_self == 0) ld_self = pe_dim;
    if (int rc = check_ld("lstep_gather_explicit_fwd", ld_edge, time_dim + feat_dim, ld_node, feat_dim, ld_pe, pe_dim + time_dim, ld_self, pe_dim)) return rc;
    lstep_csr_t none{nullptr, nullptr, nullptr, nullptr, num_rows, 0, 0};   // (only num_rows is read: the bound of the self-row lookups)
    GatherParams p{none, node_raw, edge_raw, pe, feat_dim, pe_dim, time_dim, time_w, time_b, edge_agg_w, node_ids, times,
                   batch, num_neighbors, time_gap, out_edge, out_node, out_pe, out_self, nullptr, ld_edge, ld_node, ld_pe, ld_self,
                   nbr, eid, nt, nbr_gap, nt_gap, (branches & LSTEP_WEIGHTED_SUM) ? 1 : 0, nullptr, kWavesPerBlock};
    const bool coop = en && time_gap > kCoopMin && !(branches & LSTEP_WEIGHTED_SUM);
    if (coop) p.rows_per_block = coop_rows_per_block(batch);
    const dim3 grid((unsigned)((batch + p.rows_per_block - 1) / p.rows_per_block)), block(kBlock);
    if (coop)
        hipLaunchKernelGGL((gather_aggregate_fwd_kernel<true, false, true, true>), grid, block, 0, (hipStream_t)stream, p);
    else if (en) hipLaunchKernelGGL((gather_aggregate_fwd_kernel<true, false, true>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gather_aggregate_fwd_kernel<false, true, true>), grid, block, 0, (hipStream_t)stream, p);
    return check_launch("gather_aggregate_fwd_kernel<explicit>");
}

extern "C" int lstep_gather_aggregate_bwd(const lstep_csr_t* csr, const float* edge_raw, int32_t feat_dim, int32_t pe_dim,
                                          const float* time_w, const float* time_b, int32_t time_dim, const int64_t* node_ids,
                                          const double* times, const int32_t* count, int64_t batch, int32_t num_neighbors,
                                          const float* grad_edge, const float* grad_pe_agg, const float* grad_self,
                                          int32_t ld_edge, int32_t ld_pe, int32_t ld_self, const int32_t* slot_of,
                                          float* out_slot_dot, float* grad_pe_rows, int32_t* out_hits, void* stream) {
    if (num_neighbors <= 0) return set_error(LSTEP_EINVAL, "Number of sampled neighbors for each node should be greater than 0!");
    if (int rc = check_dims("lstep_gather_aggregate_bwd", feat_dim, pe_dim, time_dim)) return rc;
    if (batch < 0) return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_bwd: negative batch");
    if (batch == 0) return LSTEP_OK;
    if (!csr || !csr->indptr || !csr->nbr || !csr->eid || !csr->ts || !time_w || !time_b || !node_ids || !times || !count)
        return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_bwd: NULL pointer");
    if (grad_edge && (!edge_raw || !out_slot_dot)) return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_bwd: grad_edge needs edge_raw and out_slot_dot");
    if ((grad_pe_agg || grad_self) && !grad_pe_rows && !out_hits) return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_bwd: PE gradients need grad_pe_rows or out_hits");
    if (out_hits && !slot_of) return set_error(LSTEP_EINVAL, "lstep_gather_aggregate_bwd: out_hits needs slot_of");
    if (ld_edge == 0) ld_edge = time_dim + feat_dim;
    if (ld_pe == 0) ld_pe = pe_dim + time_dim;
    if (ld_self == 0) ld_self = pe_dim;
    if (int rc = check_ld("lstep_gather_aggregate_bwd", ld_edge, time_dim + feat_dim, feat_dim, feat_dim, ld_pe, pe_dim + time_dim, ld_self, pe_dim)) return rc;
    GatherBwdParams p{*csr, edge_raw, feat_dim, pe_dim, time_dim, time_w, time_b, node_ids, times, count, batch, num_neighbors,
                      grad_edge, grad_pe_agg, grad_self, slot_of, out_slot_dot, grad_pe_rows, out_hits, ld_edge, ld_pe, ld_self,
                      nullptr, nullptr, nullptr};
    const dim3 grid((unsigned)((batch + kWavesPerBlock - 1) / kWavesPerBlock)), block(kBlock);
    hipLaunchKernelGGL(gather_aggregate_bwd_kernel<false>, grid, block, 0, (hipStream_t)stream, p);
    return check_launch("gather_aggregate_bwd_kernel");
}

// Backward of lstep_gather_explicit_fwd on the same slot lists: call it once with the edge channel's list (grad_edge -> out_slot_dot) and once
// with the PE channel's list (grad_pe_agg / grad_self -> grad_pe_rows or out_hits).
extern "C" int lstep_gather_explicit_bwd(const float* edge_raw, int32_t feat_dim, int32_t pe_dim, const float* time_w, const float* time_b,
                                         int32_t time_dim, const int64_t* node_ids, const double* times, int64_t batch, int32_t num_neighbors,
                                         const int64_t* nbr, const int64_t* eid, const float* nt, int64_t num_rows, const float* grad_edge,
                                         const float* grad_pe_agg, const float* grad_self, int32_t ld_edge, int32_t ld_pe, int32_t ld_self,
                                         const int32_t* slot_of, float* out_slot_dot, float* grad_pe_rows, int32_t* out_hits, void* stream) {
    if (num_neighbors <= 0) return set_error(LSTEP_EINVAL, "Number of sampled neighbors for each node should be greater than 0!");
    if (int rc = check_dims("lstep_gather_explicit_bwd", feat_dim, pe_dim, time_dim)) return rc;
    if (batch < 0 || num_rows <= 0) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_bwd: bad sizes");
    if (batch == 0) return LSTEP_OK;
    if (!time_w || !time_b || !node_ids || !times || !nbr || !nt) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_bwd: NULL pointer");
    if (grad_edge && (!edge_raw || !out_slot_dot || !eid)) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_bwd: grad_edge needs edge_raw, edge ids and out_slot_dot");
    if ((grad_pe_agg || grad_self) && !grad_pe_rows && !out_hits) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_bwd: PE gradients need grad_pe_rows or out_hits");
    if (out_hits && !slot_of) return set_error(LSTEP_EINVAL, "lstep_gather_explicit_bwd: out_hits needs slot_of");
    if (ld_edge == 0) ld_edge = time_dim + feat_dim;
    if (ld_pe == 0) ld_pe = pe_dim + time_dim;
    if (ld_self == 0) ld_self = pe_dim;
    if (int rc = check_ld("lstep_gather_explicit_bwd", ld_edge, time_dim + feat_dim, feat_dim, feat_dim, ld_pe, pe_dim + time_dim, ld_self, pe_dim)) return rc;
    lstep_csr_t none{nullptr, nullptr, nullptr, nullptr, num_rows, 0, 0};
    GatherBwdParams p{none, edge_raw, feat_dim, pe_dim, time_dim, time_w, time_b, node_ids, times, nullptr, batch, num_neighbors,
                      grad_edge, grad_pe_agg, grad_self, slot_of, out_slot_dot, grad_pe_rows, out_hits, ld_edge, ld_pe, ld_self, nbr, eid, nt};
    const dim3 grid((unsigned)((batch + kWavesPerBlock - 1) / kWavesPerBlock)), block(kBlock);
    hipLaunchKernelGGL(gather_aggregate_bwd_kernel<true>, grid, block, 0, (hipStream_t)stream, p);
    return check_launch("gather_aggregate_bwd_kernel<explicit>");
}
