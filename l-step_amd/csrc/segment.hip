// Segmented row sums: U1 / U2 message accumulation of update_pe without the dense [N+1, P+D] scatter target
//   (reference models/LSTEP.py:282-290: per batch edge, both directions; :319-322: per sampled neighbour), and the
//   sort-based PE-gradient reduction of the gather backward.  Entries are pre-grouped by destination segment; the
//   entry array is cut into fixed 64-entry chunks (one wave each) so hub segments cannot serialise the kernel.
#include "lstep_common.h"

namespace lstep {

constexpr int kSegInFlight = 8;
constexpr int kShort = 64;   // segments of up to 64 entries are summed by one wave; longer ones (hub nodes) are cut at the chunk boundaries
// Entries per wave (`chunk`, a launch parameter): 64 for long lists, 16 for short ones.  A wave keeps 8 rows in flight, so a 64-entry chunk
// is 8 dependent rounds of memory latency whatever the list length: 32 768 entries = 512 waves took 38 us (22 MB: latency, not bandwidth).
__host__ __device__ inline int segment_chunk(int64_t num_entries) { return num_entries <= 65536 ? 16 : 64; }

// flush one run of a segment: plain store when this chunk holds the whole segment.  A segment split over several chunks (long segments =
// hub nodes): with a scratch buffer (`part`, this run's slot) the partial sum is parked there and segment_join_split_rows_kernel adds the
// partials of a segment in chunk order -- deterministic whatever the scheduling; without one, float atomics (contiguous dwords per
// wave-instruction) into the zeroed row, in arrival order.
__device__ __forceinline__ void flush_run(float* __restrict__ o, const float4& acc, float t0, float t1, int W, int D, bool whole,
                                          bool accumulate, int lane, float* __restrict__ part) {
    const bool wa = lane < (W >> 2);
    if (whole && accumulate) {   // this wave is the only writer of the row: plain read-modify-write
        if (wa) {
            const float4 old = ld4(o + lane * 4);
            st4(o + lane * 4, make_float4(old.x + acc.x, old.y + acc.y, old.z + acc.z, old.w + acc.w));
        }
        if (lane < D) o[W + lane] += t0;
        if (lane + kWave < D) o[W + lane + kWave] += t1;
    } else if (whole) {
        if (wa) st4(o + lane * 4, acc);
        if (lane < D) o[W + lane] = t0;
        if (lane + kWave < D) o[W + lane + kWave] = t1;
    } else if (part) {
        if (wa) st4(part + lane * 4, acc);
        if (lane < D) part[W + lane] = t0;
        if (lane + kWave < D) part[W + lane + kWave] = t1;
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

// How many of the entries ent_seg[at .. at + 64) (forward) / ent_seg[at - 64 .. at) (backward) next to position `at` carry segment `sg`
// without a gap: 0 .. 64, where 64 means "at least 64".  One wave-wide load.
__device__ __forceinline__ int run_forward(const int32_t* __restrict__ ent_seg, int64_t at, int64_t n, int sg, int lane) {
    const bool same = (at + lane < n) && ent_seg[at + lane] == sg;
    const unsigned long long diff = ~__ballot(same);
    return diff ? __builtin_ctzll(diff) : kWave;
}
__device__ __forceinline__ int run_backward(const int32_t* __restrict__ ent_seg, int64_t at, int sg, int lane) {
    const bool same = (at - 1 - lane >= 0) && ent_seg[at - 1 - lane] == sg;
    const unsigned long long diff = ~__ballot(same);
    return diff ? __builtin_ctzll(diff) : kWave;
}

// Entries are sorted by segment; wave c is responsible for the 64-entry chunk [c * kChunk, (c + 1) * kChunk).  A segment of at most 64
// entries that straddles a chunk boundary belongs WHOLLY to the chunk it starts in: that wave reads on past its chunk (up to 63 more
// entries), the next wave skips them -- so ordinary segments are summed by one wave, in entry order, and stored with plain stores.  Only
// segments of more than 64 entries (hub nodes) are cut at the chunk boundaries; their per-chunk partial sums are parked in `parts`
// (deterministic form) or added with float atomics (`parts` == NULL).  Rows are fetched 8 at a time regardless of segment boundaries
// (memory-level parallelism does not depend on the run lengths); the accumulation walks the entries in order and flushes a run whenever
// the segment id changes.  chunk_flags[c] (with `parts`): bit 0 = slot 0 holds a partial that continues a cut segment, bit 1 = slot 1
// holds the partial that STARTS a cut segment, bit 2 = the slot-0 run fills the chunk and the segment goes on.
__global__ __launch_bounds__(kBlock) void segment_rows_sum_kernel(const float* __restrict__ table, int W, int ld_table,
                                                                   const float* __restrict__ tw, const float* __restrict__ tb, int D,
                                                                   const int32_t* __restrict__ ent_seg, const int32_t* __restrict__ ent_row,
                                                                   const float* __restrict__ ent_dt, int64_t num_entries,
                                                                   float* __restrict__ out, int ld_out, bool accumulate,
                                                                   const int32_t* __restrict__ num_live, float* __restrict__ parts,
                                                                   int32_t* __restrict__ chunk_flags, int row_div, int kChunk) {
    const int lane = lane_id();
    const int64_t chunk = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t e0 = chunk * kChunk;
    // partial sums of cut segments: two slots of W + D floats per chunk
    float* const part0 = parts ? parts + (chunk * 2) * (int64_t)(W + D) : nullptr;
    if (num_live) {      // the entry list is padded: only its first min(*num_live, num_entries) entries count (lstep_sort_live_bounded)
        const int64_t live = *num_live;
        if (live < num_entries) num_entries = live;
    }
    if (e0 >= num_entries) {
        if (chunk_flags && lane == 0) chunk_flags[chunk] = 0;
        return;
    }
    const int64_t e1 = (e0 + kChunk < num_entries) ? e0 + kChunk : num_entries;
    const int seg_first = ent_seg[e0], seg_last = ent_seg[e1 - 1];
    // head: a segment that began in an earlier chunk.  Short (<= 64 entries in all): the earlier wave sums it, skip it here; long: cut.
    int64_t b0 = e0;
    bool head_cut = false;
    if (e0 > 0 && ent_seg[e0 - 1] == seg_first) {
        const int back = run_backward(ent_seg, e0, seg_first, lane);
        const int lead = run_forward(ent_seg, e0, num_entries, seg_first, lane);
        if (back + lead <= kShort) b0 = e0 + lead; else head_cut = true;
    }
    // tail: a segment that goes on into the next chunk.  Short and begun here: read on to its end; long: cut.
    int64_t b1 = e1;
    bool tail_cut = false;
    if (e1 < num_entries && ent_seg[e1] == seg_last) {
        const int tail = run_backward(ent_seg, e1, seg_last, lane);
        const int fwd = run_forward(ent_seg, e1, num_entries, seg_last, lane);
        if (tail + fwd <= kShort) b1 = e1 + fwd; else tail_cut = true;
    }
    if (chunk_flags && lane == 0)
        chunk_flags[chunk] = (head_cut ? 1 : 0) | ((tail_cut && !(head_cut && seg_first == seg_last)) ? 2 : 0) |
                             ((head_cut && tail_cut && seg_first == seg_last) ? 4 : 0);
    if (b0 >= b1) return;      // (every entry of the chunk belongs to a short segment of the previous chunk)
    const bool wa = lane < (W >> 2);
    const float w0 = lane < D ? tw[lane] : 0.f, b0f = lane < D ? tb[lane] : 0.f;
    const float w1 = lane + kWave < D ? tw[lane + kWave] : 0.f, b1f = lane + kWave < D ? tb[lane + kWave] : 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float t0 = 0.f, t1 = 0.f;
    int cur = -1;  // segment of the open run
    for (int64_t c0 = b0; c0 < b1; c0 += kWave) {
        const int m = (int)((b1 - c0) < kWave ? (b1 - c0) : kWave);
        int sg = 0, r = 0;
        float dt = 0.f;
        if (lane < m) {
            sg = ent_seg[c0 + lane];
            r = ent_row[c0 + lane];
            if (row_div > 1) r /= row_div;      // (ent_row holds slot indices row * K + j: the gradient hits of the gather backward)
            if (D > 0) dt = ent_dt[c0 + lane];
        }
        settle(sg ^ r ^ __float_as_int(dt));
        for (int j = 0; j < m; j += kSegInFlight) {
            float4 x[kSegInFlight];
            if (wa) {
#pragma unroll
                for (int u = 0; u < kSegInFlight; ++u) {
                    const int64_t rj = bcast_i32(r, (j + u) < m ? (j + u) : (m - 1));  // tail slots re-read the last row (unused)
                    x[u] = ld4(table + rj * ld_table + lane * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < kSegInFlight; ++u) {
                if ((j + u) >= m) break;
                const int sj = bcast_i32(sg, j + u);
                if (sj != cur) {
                    if (cur >= 0) {
                        const bool cut_head = head_cut && cur == seg_first, cut = cut_head || (tail_cut && cur == seg_last);
                        flush_run(out + (int64_t)cur * ld_out, acc, t0, t1, W, D, !cut, accumulate, lane,
                                  part0 ? part0 + (cut_head ? 0 : W + D) : nullptr);
                    }
                    cur = sj;
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    t0 = t1 = 0.f;
                }
                if (wa) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
                if (D > 0) {
                    const float dj = bcast_f32(dt, j + u);
                    if (lane < D) t0 += time_feat(dj, w0, b0f);
                    if (lane + kWave < D) t1 += time_feat(dj, w1, b1f);
                }
            }
        }
    }
    if (cur >= 0) {
        const bool cut_head = head_cut && cur == seg_first, cut = cut_head || (tail_cut && cur == seg_last);
        flush_run(out + (int64_t)cur * ld_out, acc, t0, t1, W, D, !cut, accumulate, lane, part0 ? part0 + (cut_head ? 0 : W + D) : nullptr);
    }
}

// Hub segments (thousands of entries: a power-law graph's top nodes collect a fifth of a batch's messages) span hundreds of chunks; adding
// their parked partials one after another is a chain of dependent loads (434 us per launch on the Zipf-1.2 c4 step, round 4).  The join is a
// FIXED-SHAPE two-level tree instead: a first pass sums every aligned group of kJoinGroup consecutive MIDDLE chunks (chunks that lie wholly
// inside one cut segment, flags == 5) into one group partial -- all its loads in flight at once, added in chunk order; the second pass walks a
// cut segment's chunks as before but takes a whole group in one step wherever a group partial exists, eight group partials in flight.  The
// grouping is a function of the chunk layout, i.e. of the inputs alone: results stay deterministic (replicas on different GPUs bit-identical).
constexpr int kJoinGroup = 16;
constexpr int kJoinMinChunks = 4 * kJoinGroup;     // shorter lists: the serial walk is at most that many steps, skip the first pass

__global__ __launch_bounds__(kBlock) void segment_join_groups_kernel(int64_t chunks, int W, int D, const float* __restrict__ parts,
                                                                     const int32_t* __restrict__ chunk_flags, float* __restrict__ gparts,
                                                                     int32_t* __restrict__ gflags) {
    const int lane = lane_id();
    const int64_t g = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t groups = chunks / kJoinGroup;
    if (g >= groups) return;
    const int64_t c0 = g * kJoinGroup;
    const int fl = lane < kJoinGroup ? chunk_flags[c0 + lane] : 5;
    const bool all = __ballot(fl == 5) == ~0ull;
    if (lane == 0) gflags[g] = all ? 1 : 0;
    if (!all) return;
    const int ldp = W + D;
    const bool wa = lane < (W >> 2);
    const float* p = parts + (c0 * 2) * (int64_t)ldp;      // slot 0 of chunk c0; the next chunk's slot 0 is 2 * ldp floats on
    float4 x[kJoinGroup];
    float a0[kJoinGroup], a1[kJoinGroup];
#pragma unroll
    for (int u = 0; u < kJoinGroup; ++u) {
        x[u] = wa ? ld4(p + (int64_t)u * 2 * ldp + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        a0[u] = lane < D ? p[(int64_t)u * 2 * ldp + W + lane] : 0.f;
        a1[u] = lane + kWave < D ? p[(int64_t)u * 2 * ldp + W + lane + kWave] : 0.f;
    }
    float4 acc = x[0];
    float t0 = a0[0], t1 = a1[0];
#pragma unroll
    for (int u = 1; u < kJoinGroup; ++u) {
        acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w;
        t0 += a0[u];
        t1 += a1[u];
    }
    float* o = gparts + g * (int64_t)ldp;
    if (wa) st4(o + lane * 4, acc);
    if (lane < D) o[W + lane] = t0;
    if (lane + kWave < D) o[W + lane + kWave] = t1;
}

// Second pass of the deterministic form: one LANE per chunk looks at the chunk's flags; the rare chunk in which a cut (> 64 entries) segment
// starts has its wave add that segment's parked partials in chunk order (whole groups of middle chunks through their group partial, see
// above) and write (or, `accumulate`, add to) its row -- the only writer of the row.  A launch over a list without hub segments is
// ~chunks / 64 waves that read one flag each.
__global__ __launch_bounds__(kBlock) void segment_join_split_rows_kernel(const int32_t* __restrict__ ent_seg, int64_t chunks, int W, int D,
                                                                         const float* __restrict__ parts, const int32_t* __restrict__ chunk_flags,
                                                                         float* __restrict__ out, int ld_out, bool accumulate, int kChunk,
                                                                         const float* __restrict__ gparts, const int32_t* __restrict__ gflags) {
    const int lane = lane_id();
    const int64_t c_lane = ((int64_t)blockIdx.x * kWavesPerBlock + wave_in_block()) * kWave + lane;
    const int fl = c_lane < chunks ? chunk_flags[c_lane] : 0;
    unsigned long long todo = __ballot((fl & 2) != 0);
    const int ldp = W + D;
    const bool wa = lane < (W >> 2);
    const int64_t groups = gflags ? chunks / kJoinGroup : 0;
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t c = c_lane - lane + src;                                      // the chunk in which a cut segment starts
        const int sg = ent_seg[(c + 1) * kChunk];                                   // (it goes on into the next chunk: that entry exists)
        const float* p = parts + (c * 2 + 1) * (int64_t)ldp;
        float4 acc = wa ? ld4(p + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float t0 = lane < D ? p[W + lane] : 0.f, t1 = lane + kWave < D ? p[W + lane + kWave] : 0.f;
        int64_t j = c + 1;
        while (j < chunks) {
            if (groups && (j & (kJoinGroup - 1)) == 0) {
                // how many consecutive whole groups of middle chunks start here (they all belong to this segment: chunk j - 1 goes on into j)
                const int64_t g0 = j / kJoinGroup;
                const bool full = (g0 + lane < groups) && gflags[g0 + lane] != 0;
                const unsigned long long nf = ~__ballot(full);
                const int run = nf ? __builtin_ctzll(nf) : kWave;
                if (run > 0) {
                    for (int q = 0; q < run; q += kSegInFlight) {
                        float4 x[kSegInFlight];
                        float a0[kSegInFlight], a1[kSegInFlight];
#pragma unroll
                        for (int u = 0; u < kSegInFlight; ++u) {
                            const float* gp = gparts + (g0 + ((q + u) < run ? (q + u) : (run - 1))) * (int64_t)ldp;
                            x[u] = wa ? ld4(gp + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                            a0[u] = lane < D ? gp[W + lane] : 0.f;
                            a1[u] = lane + kWave < D ? gp[W + lane + kWave] : 0.f;
                        }
#pragma unroll
                        for (int u = 0; u < kSegInFlight; ++u) {
                            if ((q + u) >= run) break;
                            acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w;
                            t0 += a0[u];
                            t1 += a1[u];
                        }
                    }
                    j += (int64_t)run * kJoinGroup;
                    continue;
                }
            }
            p = parts + (j * 2) * (int64_t)ldp;                                     // chunk j's first run continues the segment
            if (wa) { const float4 v = ld4(p + lane * 4); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            if (lane < D) t0 += p[W + lane];
            if (lane + kWave < D) t1 += p[W + lane + kWave];
            if (!(chunk_flags[j] & 4)) break;
            ++j;
        }
        float* o = out + (int64_t)sg * ld_out;
        if (accumulate) {
            if (wa) { const float4 old = ld4(o + lane * 4); st4(o + lane * 4, make_float4(old.x + acc.x, old.y + acc.y, old.z + acc.z, old.w + acc.w)); }
            if (lane < D) o[W + lane] += t0;
            if (lane + kWave < D) o[W + lane + kWave] += t1;
        } else {
            if (wa) st4(o + lane * 4, acc);
            if (lane < D) o[W + lane] = t0;
            if (lane + kWave < D) o[W + lane + kWave] = t1;
        }
    }
}

// Pre-pass for an UNINITIALISED output in the atomic form: only the rows of CUT segments (more than 64 entries, see above) are accumulated
// with atomics and need zeros; every other row that owns entries is written whole by one wave.  One wave per chunk boundary.
__global__ __launch_bounds__(kBlock) void segment_zero_split_rows_kernel(const int32_t* __restrict__ ent_seg, int64_t num_entries,
                                                                         float* __restrict__ out, int ld_out, int width,
                                                                         const int32_t* __restrict__ num_live, int kChunk) {
    const int lane = lane_id();
    const int64_t c = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block() + 1;   // boundary between chunk c - 1 and chunk c: one wave each
    const int64_t e = c * kChunk;
    if (num_live) {
        const int64_t live = *num_live;
        if (live < num_entries) num_entries = live;
    }
    if (e >= num_entries) return;
    const int sg = ent_seg[e];
    if (ent_seg[e - 1] != sg) return;
    if (run_backward(ent_seg, e, sg, lane) + run_forward(ent_seg, e, num_entries, sg, lane) <= kShort) return;   // short: one wave sums it
    float* o = out + (int64_t)sg * ld_out;
    for (int k = lane * 4; k < width; k += kWave * 4) st4(o + k, make_float4(0.f, 0.f, 0.f, 0.f));
}

__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(float* __restrict__ table, int W, const int64_t* __restrict__ ids,
                                                               int64_t num_ids, const float* __restrict__ rows) {
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (i >= num_ids) return;
    const int64_t r = LSTEP_CHECKED(ids[i], LSTEP_NODE_ROWS(), kCheckScatterRowsId);      // (checked builds only: lstep_common.h)
    for (int c = lane; c < (W >> 2); c += kWave) st4(table + r * W + c * 4, ld4(rows + i * W + c * 4));
}

// table[ids[i], :] += tanh(z[i, :])   (residual PE update of models/LSTEP.py:299-303 and :335-339, fused with the write)
__global__ __launch_bounds__(kBlock) void residual_tanh_rows_kernel(float* __restrict__ table, int W, const int64_t* __restrict__ ids,
                                                                     int64_t num_ids, const float* __restrict__ z, int ld_z) {
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (i >= num_ids) return;
    const int64_t r = LSTEP_CHECKED(ids[i], LSTEP_NODE_ROWS(), kCheckScatterRowsId);
    for (int c = lane; c < (W >> 2); c += kWave) {
        const float4 a = ld4(table + r * W + c * 4);
        const float4 b = ld4(z + i * (int64_t)ld_z + c * 4);
        st4(table + r * W + c * 4, make_float4(a.x + tanhf(b.x), a.y + tanhf(b.y), a.z + tanhf(b.z), a.w + tanhf(b.w)));
    }
}

// update_pe phase 2, row 0: every PADDED slot of the sampled neighbourhoods scatters cat[pe[source], 0] into row 0 (models/LSTEP.py:317-322:
// the reference lets row 0 collect them like any other neighbour).  partial[blk, :] = sum over the block's source rows r of
// (number of zero slots of nbr[r, :]) * table[ids[r], :W]; the caller adds the partials.  Rows without padding cost one ballot.
constexpr int kPadRowsPerWave = 16;
__global__ __launch_bounds__(kBlock) void padding_rows_sum_kernel(const int64_t* __restrict__ nbr, int K, const int64_t* __restrict__ ids, int64_t n,
                                                                   const float* __restrict__ table, int W, int ld, float* __restrict__ partial) {
    __shared__ float4 sh[kWavesPerBlock][kMaxRowVec];
    const int lane = lane_id(), wave = wave_in_block();
    const bool wa = lane < (W >> 2);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t r0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * kPadRowsPerWave;
    // All 16 rows of the wave at once: their slot lists in one batch of loads, then their table rows in one batch (rows without padding
    // re-read the first row with weight 0).  Row by row it is 16 x 2 dependent memory latencies per wave: 52 us at 32 768 rows.
    int64_t id = 0;
    if (lane < kPadRowsPerWave) {
        const int64_t r = r0 + lane < n ? r0 + lane : n - 1;
        id = ids ? ids[r] : r;
    }
    int zeros[kPadRowsPerWave];
#pragma unroll
    for (int u = 0; u < kPadRowsPerWave; ++u) zeros[u] = 0;
    for (int j0 = 0; j0 < K; j0 += kWave) {
        bool z[kPadRowsPerWave];
#pragma unroll
        for (int u = 0; u < kPadRowsPerWave; ++u) z[u] = (r0 + u < n) && (j0 + lane < K) && nbr[(r0 + u) * K + j0 + lane] == 0;
#pragma unroll
        for (int u = 0; u < kPadRowsPerWave; ++u) zeros[u] += __popcll(__ballot(z[u]));
    }
    const int64_t id0 = bcast_i64(id, 0);
    float4 v[kPadRowsPerWave];
    if (wa) {
#pragma unroll
        for (int u = 0; u < kPadRowsPerWave; ++u) v[u] = ld4(table + (zeros[u] ? bcast_i64(id, u) : id0) * (int64_t)ld + lane * 4);
#pragma unroll
        for (int u = 0; u < kPadRowsPerWave; ++u) {      // (row order: the same sum as the row-by-row loop)
            if (zeros[u]) {
                const float w = (float)zeros[u];
                acc.x = fmaf(w, v[u].x, acc.x); acc.y = fmaf(w, v[u].y, acc.y); acc.z = fmaf(w, v[u].z, acc.z); acc.w = fmaf(w, v[u].w, acc.w);
            }
        }
    }
    if (wa) sh[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && wa) {
#pragma unroll
        for (int w2 = 1; w2 < kWavesPerBlock; ++w2) {
            const float4 v = sh[w2][lane];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        st4(partial + (int64_t)blockIdx.x * W + lane * 4, acc);
    }
}

// out[slot[i], :W] += rows[i, :W] for the entries with slot[i] >= 0 (float atomics: for the few stragglers that are not worth a sort)
__global__ __launch_bounds__(kBlock) void scatter_add_rows_kernel(float* __restrict__ out, int W, int ld_out, const int32_t* __restrict__ slot,
                                                                   int64_t n, const float* __restrict__ rows, int ld_rows) {
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (i >= n) return;
    const int sl = slot[i];
    if (sl < 0) return;
    for (int c = lane; c < W; c += kWave) atomicAdd(out + (int64_t)sl * ld_out + c, rows[i * (int64_t)ld_rows + c]);
}

}  // namespace lstep

using namespace lstep;

namespace {
// scratch of the deterministic form: [chunks * 2 partial rows | chunk flags | chunks / kJoinGroup group partial rows | group flags], 16-byte aligned pieces
struct JoinLayout {
    int64_t flags_off, gparts_off, gflags_off, total, groups;
};
inline JoinLayout join_layout(int64_t chunks, int64_t ldp) {
    auto up16 = [](int64_t b) { return (b + 15) / 16 * 16; };
    JoinLayout l;
    l.groups = chunks >= kJoinMinChunks ? chunks / kJoinGroup : 0;
    l.flags_off = up16(chunks * 2 * ldp * (int64_t)sizeof(float));
    l.gparts_off = l.flags_off + up16(chunks * (int64_t)sizeof(int32_t));
    l.gflags_off = l.gparts_off + up16(l.groups * ldp * (int64_t)sizeof(float));
    l.total = l.gflags_off + up16(l.groups * (int64_t)sizeof(int32_t));
    return l;
}

// the two passes that join the parked partials of cut segments (deterministic form)
inline void launch_join(const int32_t* ent_seg, int64_t chunks, int W, int D, void* workspace, float* out, int ld_out, bool accumulate, int kChunk,
                        hipStream_t stream) {
    const JoinLayout l = join_layout(chunks, W + D);
    float* parts = (float*)workspace;
    int32_t* flags = (int32_t*)((char*)workspace + l.flags_off);
    float* gparts = l.groups ? (float*)((char*)workspace + l.gparts_off) : nullptr;
    int32_t* gflags = l.groups ? (int32_t*)((char*)workspace + l.gflags_off) : nullptr;
    if (l.groups)
        hipLaunchKernelGGL(segment_join_groups_kernel, dim3((unsigned)((l.groups + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, stream,
                           chunks, W, D, parts, flags, gparts, gflags);
    hipLaunchKernelGGL(segment_join_split_rows_kernel, dim3((unsigned)((chunks + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, ent_seg, chunks, W,
                       D, parts, flags, out, ld_out, accumulate, kChunk, gparts, gflags);
}
}  // namespace

extern "C" int64_t lstep_segment_rows_sum_workspace(int64_t num_entries, int32_t width, int32_t time_dim) {
    const int kChunk = segment_chunk(num_entries);
    if (num_entries <= kChunk) return 0;       // one chunk: nothing can be cut
    const int64_t chunks = (num_entries + kChunk - 1) / kChunk;
    return join_layout(chunks, width + time_dim).total;
}

extern "C" int lstep_segment_rows_sum(const float* table, int32_t width, int32_t ld_table, const float* time_w, const float* time_b,
                                      int32_t time_dim, const int32_t* ent_seg, const int32_t* ent_row, const float* ent_dt,
                                      int64_t num_entries, float* out, int32_t ld_out, int32_t accumulate, const int32_t* num_live,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    if (num_entries < 0) return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum: negative count");
    if (num_entries == 0) return LSTEP_OK;
    if (ld_table == 0) ld_table = width;
    if (ld_out == 0) ld_out = width + time_dim;
    if (width <= 0 || (width & 3) || width > 4 * kMaxRowVec || time_dim < 0 || (time_dim & 3) || time_dim > kMaxTimeDim || ld_table < width ||
        (ld_table & 3) || ld_out < width + time_dim || (ld_out & 3))
        return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum: unsupported widths W=%d D=%d ld_table=%d ld_out=%d", width, time_dim, ld_table, ld_out);
    if (!table || !ent_seg || !ent_row || !out || (time_dim > 0 && (!time_w || !time_b || !ent_dt)))
        return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum: NULL pointer");
    const int kChunk = segment_chunk(num_entries);
    const int64_t chunks = (num_entries + kChunk - 1) / kChunk;
    const unsigned grid = (unsigned)((chunks + kWavesPerBlock - 1) / kWavesPerBlock);
    const unsigned bgrid = (unsigned)((chunks - 1 + kWavesPerBlock - 1) / kWavesPerBlock);
    float* parts = nullptr;
    int32_t* flags = nullptr;
    if (workspace && chunks > 1) {       // deterministic form: cut segments are joined in chunk order by a second pass
        if (((uintptr_t)workspace & 15) || workspace_bytes < lstep_segment_rows_sum_workspace(num_entries, width, time_dim))
            return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum: workspace too small or misaligned (need %lld bytes, 16-byte aligned)",
                             (long long)lstep_segment_rows_sum_workspace(num_entries, width, time_dim));
        parts = (float*)workspace;
        flags = (int32_t*)((char*)workspace + join_layout(chunks, width + time_dim).flags_off);
    }
    if (!parts && accumulate == 2 && chunks > 1)   // uninitialised output, atomic form: zero just the rows the atomics will add to
        hipLaunchKernelGGL(segment_zero_split_rows_kernel, dim3(bgrid), dim3(kBlock), 0, (hipStream_t)stream, ent_seg, num_entries, out,
                           (int)ld_out, (int)(width + time_dim), num_live, kChunk);
    hipLaunchKernelGGL(segment_rows_sum_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, table, (int)width, (int)ld_table, time_w,
                       time_b, (int)time_dim, ent_seg, ent_row, ent_dt, num_entries, out, (int)ld_out, accumulate == 1, num_live, parts, flags, 1, kChunk);
    if (parts) launch_join(ent_seg, chunks, (int)width, (int)time_dim, workspace, out, (int)ld_out, accumulate == 1, kChunk, (hipStream_t)stream);
    return check_launch("segment_rows_sum_kernel");
}

// lstep_segment_rows_sum (no time part, accumulate 0 / 1) over a PADDED entry list: only the first min(*num_live, num_entries) entries count
extern "C" int lstep_segment_rows_sum_live(const float* table, int32_t width, int32_t ld_table, const int32_t* ent_seg, const int32_t* ent_row,
                                           int32_t row_div, int64_t num_entries, const int32_t* num_live, float* out, int32_t ld_out,
                                           int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (num_entries < 0) return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum_live: negative count");
    if (num_entries == 0) return LSTEP_OK;
    if (ld_table == 0) ld_table = width;
    if (ld_out == 0) ld_out = width;
    if (width <= 0 || (width & 3) || width > 4 * kMaxRowVec || ld_table < width || (ld_table & 3) || ld_out < width || (ld_out & 3) ||
        (accumulate != 0 && accumulate != 1) || row_div < 1)
        return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum_live: unsupported arguments");
    if (!table || !ent_seg || !ent_row || !out || !num_live) return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum_live: NULL pointer");
    const int kChunk = segment_chunk(num_entries);
    const int64_t chunks = (num_entries + kChunk - 1) / kChunk;
    const unsigned grid = (unsigned)((chunks + kWavesPerBlock - 1) / kWavesPerBlock);
    float* parts = nullptr;
    int32_t* flags = nullptr;
    if (workspace && chunks > 1) {
        if (((uintptr_t)workspace & 15) || workspace_bytes < lstep_segment_rows_sum_workspace(num_entries, width, 0))
            return set_error(LSTEP_EINVAL, "lstep_segment_rows_sum_live: workspace too small or misaligned");
        parts = (float*)workspace;
        flags = (int32_t*)((char*)workspace + join_layout(chunks, width).flags_off);
    }
    hipLaunchKernelGGL(segment_rows_sum_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, table, (int)width, (int)ld_table,
                       (const float*)nullptr, (const float*)nullptr, 0, ent_seg, ent_row, (const float*)nullptr, num_entries, out, (int)ld_out,
                       accumulate == 1, num_live, parts, flags, (int)row_div, kChunk);
    if (parts) launch_join(ent_seg, chunks, (int)width, 0, workspace, out, (int)ld_out, accumulate == 1, kChunk, (hipStream_t)stream);
    return check_launch("segment_rows_sum_kernel<live>");
}

// ---- Small batches: the gradient of the spliced rows WITHOUT a sort.  out[u, :W] = sum over the hits (b, j) with hits[b * K + j] == u of
// g_hit[b, :W], in ascending (b, j), then sum over the rows b whose own node is spliced row u (slot_of[ids[b]] == u) of g_self[b, :W], in
// ascending b.  One workgroup per spliced row: each of its four waves scans a quarter of the list (64 entries per step: one coalesced read
// and a ballot, sixteen steps in flight), parks the
// matching row numbers in a per-wave LDS queue and fetches them sixteen at a time.  O(U * n / 64) scan steps: a few microseconds for the
// reference's own batch sizes (B = 200 / 600: U <= 1 200 rows, n <= 36 000 hits), where the general path -- compaction, a
// single-workgroup sort, two segment sums with their joins, a scatter with atomics: nine dependent launches -- is the critical chain of the
// whole captured step (round 4).  Fixed order of summation, no atomics.
constexpr int kSpliceQueue = 16 * 64 + 64;      // queued row numbers per wave: a round of kSpliceAhead scan steps adds at most 64 each
constexpr int kSpliceInFlight = 16;
constexpr int kSpliceAhead = 16;       // scan steps (64 entries each) whose loads are in flight together
__device__ __forceinline__ void splice_take(const int* __restrict__ q, int first, int count, const float* __restrict__ table, int ld, bool wa, int lane,
                                            float4& acc) {
    for (int g = 0; g < count; g += kSpliceInFlight) {
        float4 x[kSpliceInFlight];
        if (wa) {
#pragma unroll
            for (int i = 0; i < kSpliceInFlight; ++i) {
                const int b = q[first + ((g + i) < count ? (g + i) : (count - 1))];      // (tail slots re-read the last row: unused)
                x[i] = ld4(table + (int64_t)b * ld + lane * 4);
            }
#pragma unroll
            for (int i = 0; i < kSpliceInFlight; ++i) {
                if ((g + i) < count) { acc.x += x[i].x; acc.y += x[i].y; acc.z += x[i].z; acc.w += x[i].w; }      // (no early exit: x[] stays in registers)
            }
        }
    }
}

template <int kSpliceWaves>
__global__ __launch_bounds__(kSpliceWaves * kWave) void spliced_grad_small_kernel(const int32_t* __restrict__ hits, int64_t num_hits, int K,
                                                                     const float* __restrict__ g_hit, int ld_hit, const int32_t* __restrict__ slot_of,
                                                                     const int64_t* __restrict__ ids, int64_t num_self, const float* __restrict__ g_self,
                                                                     int ld_self, int W, float* __restrict__ out, int ld_out, int64_t num_rows) {
    __shared__ int queue[kSpliceWaves][kSpliceQueue];
    __shared__ float4 part[kSpliceWaves][kWave];
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
    const int64_t u = blockIdx.x;                 // one workgroup per spliced row: each of its waves scans an equal share of each list
    int* q = queue[wv];
    const bool wa = lane < (W >> 2);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t n_all = pass == 0 ? num_hits : num_self;
        const float* table = pass == 0 ? g_hit : g_self;
        const int ld = pass == 0 ? ld_hit : ld_self;
        if (table == nullptr) continue;
        const int64_t quarter = ((n_all + kSpliceWaves - 1) / kSpliceWaves + kWave - 1) / kWave * kWave;      // whole scan steps
        const int64_t first = wv * quarter;
        const int64_t n = (first + quarter < n_all) ? first + quarter : n_all;       // this wave scans [first, n)
        int qn = 0;
        // kSpliceAhead scan steps are fetched together: a step is one load (two dependent ones for the own-row pass) followed by a ballot,
        // and a wave that waited for each in turn spent ~0.5 us per 64 entries (wikipedia-shaped batch: 590 steps per wave)
        for (int64_t e0 = first; e0 < n; e0 += (int64_t)kWave * kSpliceAhead) {
            int key[kSpliceAhead];
#pragma unroll
            for (int a = 0; a < kSpliceAhead; ++a) {
                const int64_t e = e0 + (int64_t)a * kWave + lane;
                key[a] = -2;
                if (e < n) key[a] = pass == 0 ? hits[e] : (int)ids[e];
            }
            if (pass == 1) {
#pragma unroll
                for (int a = 0; a < kSpliceAhead; ++a) key[a] = key[a] >= 0 ? slot_of[LSTEP_CHECKED(key[a], LSTEP_NODE_ROWS(), kCheckSplicedGradKey)] : -2;
            }
#pragma unroll
            for (int a = 0; a < kSpliceAhead; ++a) {       // (only the cheap part is unrolled: the keys stay in registers)
                const bool hit = key[a] == (int32_t)u;
                const unsigned long long m = __ballot(hit);
                const int64_t e = e0 + (int64_t)a * kWave + lane;
                if (hit) q[qn + __popcll(m & ((1ull << lane) - 1ull))] = (int)(pass == 0 ? e / K : e);
                qn += __popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
            if (qn >= kWave) {            // the queue holds a round's matches at most (kSpliceAhead x 64) plus < 64 left over
                splice_take(q, 0, qn, table, ld, wa, lane, acc);
                __builtin_amdgcn_wave_barrier();
                qn = 0;
            }
        }
        if (qn > 0) splice_take(q, 0, qn, table, ld, wa, lane, acc);
        __builtin_amdgcn_wave_barrier();
    }
    // the four partial sums meet in wave order (a fixed association: the result is a function of the inputs)
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && wa) {
        float4 t = part[0][lane];
#pragma unroll
        for (int w = 1; w < kSpliceWaves; ++w) {
            const float4 v = part[w][lane];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        st4(out + u * ld_out + lane * 4, t);
    }
}

// out[keys[e], :W] += table[e / div, :W] for the live entries e = live_index[i], i in [capacity, *count): the ones a bounded sort left out
__global__ __launch_bounds__(kBlock) void scatter_add_overflow_kernel(float* __restrict__ out, int W, int ld_out, const int32_t* __restrict__ keys,
                                                                       const int32_t* __restrict__ live_index, const int32_t* __restrict__ count,
                                                                       int64_t capacity, int div, const float* __restrict__ table, int ld_table) {
    const int lane = lane_id();
    const int64_t n = *count;
    const int64_t waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t i = capacity + (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block(); i < n; i += waves) {
        const int e = live_index[i];
        const float* src = table + (int64_t)(e / div) * ld_table;
        float* dst = out + (int64_t)keys[e] * ld_out;
        for (int c = lane; c < W; c += kWave) atomicAdd(dst + c, src[c]);
    }
}

extern "C" int lstep_scatter_add_overflow(float* out, int32_t width, int32_t ld_out, const int32_t* keys, const int32_t* live_index,
                                          const int32_t* count, int64_t capacity, int32_t div, const float* table, int32_t ld_table, void* stream) {
    if (width <= 0 || ld_out < width || ld_table < width || capacity < 0 || div <= 0) return set_error(LSTEP_EINVAL, "lstep_scatter_add_overflow: bad sizes");
    if (!out || !keys || !live_index || !count || !table) return set_error(LSTEP_EINVAL, "lstep_scatter_add_overflow: NULL pointer");
    hipLaunchKernelGGL(scatter_add_overflow_kernel, dim3(256), dim3(kBlock), 0, (hipStream_t)stream, out, (int)width, (int)ld_out, keys, live_index,
                       count, capacity, (int)div, table, (int)ld_table);
    return check_launch("scatter_add_overflow_kernel");
}

extern "C" int64_t lstep_padding_rows_sum_blocks(int64_t n) {
    return n <= 0 ? 0 : (n + kWavesPerBlock * kPadRowsPerWave - 1) / (kWavesPerBlock * kPadRowsPerWave);
}

extern "C" int lstep_padding_rows_sum(const int64_t* nbr, int32_t num_neighbors, const int64_t* ids, int64_t n, const float* table, int32_t width,
                                      int32_t ld_table, float* partial, void* stream) {
    if (n < 0 || num_neighbors <= 0 || width <= 0 || (width & 3) || width > 4 * kMaxRowVec || ld_table < width || (ld_table & 3))
        return set_error(LSTEP_EINVAL, "lstep_padding_rows_sum: bad sizes");
    if (n == 0) return LSTEP_OK;
    if (!nbr || !table || !partial) return set_error(LSTEP_EINVAL, "lstep_padding_rows_sum: NULL pointer");
    hipLaunchKernelGGL(padding_rows_sum_kernel, dim3((unsigned)lstep_padding_rows_sum_blocks(n)), dim3(kBlock), 0, (hipStream_t)stream, nbr,
                       (int)num_neighbors, ids, n, table, (int)width, (int)ld_table, partial);
    return check_launch("padding_rows_sum_kernel");
}

extern "C" int lstep_scatter_add_rows(float* out, int32_t width, int32_t ld_out, const int32_t* slot, int64_t n, const float* rows, int32_t ld_rows,
                                      void* stream) {
    if (n < 0 || width <= 0 || ld_out < width || ld_rows < width) return set_error(LSTEP_EINVAL, "lstep_scatter_add_rows: bad sizes");
    if (n == 0) return LSTEP_OK;
    if (!out || !slot || !rows) return set_error(LSTEP_EINVAL, "lstep_scatter_add_rows: NULL pointer");
    const unsigned grid = (unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, out, (int)width, (int)ld_out, slot, n, rows,
                       (int)ld_rows);
    return check_launch("scatter_add_rows_kernel");
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


extern "C" int lstep_spliced_grad_small(const int32_t* hits, int64_t num_hits, int32_t num_neighbors, const float* g_hit, int32_t ld_hit,
                                        const int32_t* slot_of, const int64_t* ids, int64_t num_self, const float* g_self, int32_t ld_self,
                                        int32_t width, float* out, int32_t ld_out, int64_t num_rows, void* stream) {
    if (num_rows < 0 || num_hits < 0 || num_self < 0 || num_neighbors <= 0) return set_error(LSTEP_EINVAL, "lstep_spliced_grad_small: bad sizes");
    if (num_rows == 0) return LSTEP_OK;
    if (width <= 0 || (width & 3) || width > 4 * kWave || ld_out < width || (ld_out & 3)) return set_error(LSTEP_EINVAL, "lstep_spliced_grad_small: unsupported width");
    if (!out || (g_hit && (!hits || ld_hit < width || (ld_hit & 3))) || (g_self && (!slot_of || !ids || ld_self < width || (ld_self & 3))))
        return set_error(LSTEP_EINVAL, "lstep_spliced_grad_small: NULL pointer or bad row stride");
    if ((((uintptr_t)out | (uintptr_t)g_hit | (uintptr_t)g_self) & 15) != 0) return set_error(LSTEP_EINVAL, "lstep_spliced_grad_small: rows must be 16-byte aligned");
    // four waves per row for short lists, eight beyond 16 k entries (the scan is a chain of dependent rounds: more waves = fewer rounds each)
    if (num_hits + num_self > 16384)
        hipLaunchKernelGGL(spliced_grad_small_kernel<8>, dim3((unsigned)num_rows), dim3(8 * kWave), 0, (hipStream_t)stream,
                           hits, num_hits, (int)num_neighbors, g_hit, (int)ld_hit, slot_of, ids, num_self, g_self, (int)ld_self, (int)width, out, (int)ld_out,
                           num_rows);
    else
    hipLaunchKernelGGL(spliced_grad_small_kernel<4>, dim3((unsigned)num_rows), dim3(kBlock), 0, (hipStream_t)stream,
                       hits, num_hits, (int)num_neighbors, g_hit, (int)ld_hit, slot_of, ids, num_self, g_self, (int)ld_self, (int)width, out, (int)ld_out,
                       num_rows);
    return check_launch("spliced_grad_small_kernel");
}
