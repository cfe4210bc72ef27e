// Host-side replay of the RNG-defined neighbour sampling strategies ('uniform', 'time_interval_aware': reference utils/utils.py:175-198).
// The reference draws with numpy's LEGACY generator -- `self.random_state.choice(a=cnt, size=K, p=p)` once per row, in row order -- so the
// sampled neighbourhoods are defined by MT19937's output stream and by the exact arithmetic of RandomState.choice:
//   p is None : randint(0, cnt) = masked rejection on 32-bit outputs (numpy/random/src/distributions: random_bounded_uint64_fill with
//               use_masked = 1 -> buffered_bounded_masked_uint32, which the legacy path does NOT buffer); cnt == 1 consumes nothing;
//   p given   : cdf = cumsum(float64(p)) / last, u = random_sample() = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two outputs,
//               idx = searchsorted(cdf, u, side='right').
// This file restates those two paths and the row loop around them (strictly-before count by binary search, gather of the picked slots);
// the generator state (624 words + position) is the caller's -- lstep_amd.sampler hands over numpy's own RandomState.get_state() and
// stores the advanced state back, so Python-side draws and native draws interleave on ONE stream.  Pure host code (no kernel): it replaces
// an O(rows) interpreter loop (49 152 choice() calls per c4 step), not a device op.  The time-sorted re-ordering of the
// sampled slots (utils/utils.py:192-196: an UNSTABLE numpy argsort whose tie order is numpy's business) stays with numpy, batched.
#include <stdint.h>
#include <string.h>

#include "lstep_common.h"

namespace {

struct Mt19937 {
    uint32_t* key;   // [624]
    int pos;
    void refill() {
        const uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrix = 0x9908b0dfu;
        int i = 0;
        uint32_t y;
        for (; i < 624 - 397; ++i) {
            y = (key[i] & kUpper) | (key[i + 1] & kLower);
            key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & kMatrix);
        }
        for (; i < 623; ++i) {
            y = (key[i] & kUpper) | (key[i + 1] & kLower);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & kMatrix);
        }
        y = (key[623] & kUpper) | (key[0] & kLower);
        key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & kMatrix);
        pos = 0;
    }
    uint32_t next() {
        if (pos == 624) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double next_double() {
        const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

inline int64_t count_before(const double* ts, int64_t lo, int64_t hi, double t) {      // np.searchsorted(ts[lo:hi], t), side 'left'
    int64_t a = lo, b = hi;
    while (a < b) {
        const int64_t mid = a + ((b - a) >> 1);
        if (ts[mid] < t) a = mid + 1; else b = mid;
    }
    return a - lo;
}

}  // namespace

extern "C" int lstep_count_before_host(const int64_t* indptr, const double* ts, int64_t num_rows, const int64_t* node_ids, const double* times,
                                       int64_t m, int64_t* out_count) {
    if (m < 0 || num_rows <= 0) return lstep::set_error(LSTEP_EINVAL, "lstep_count_before_host: bad sizes");
    if (m == 0) return LSTEP_OK;
    if (!indptr || !ts || !node_ids || !times || !out_count) return lstep::set_error(LSTEP_EINVAL, "lstep_count_before_host: NULL pointer");
    for (int64_t r = 0; r < m; ++r) {
        const int64_t node = node_ids[r];
        if (node < 0 || node >= num_rows) return lstep::set_error(LSTEP_EINVAL, "lstep_count_before_host: node id out of range");
        out_count[r] = count_before(ts, indptr[node], indptr[node + 1], times[r]);
    }
    return LSTEP_OK;
}

extern "C" int lstep_sample_random_host(const int64_t* indptr, const int64_t* nbr, const int64_t* eid, const double* ts, int64_t num_rows,
                                        const int64_t* node_ids, const double* times, int64_t m, int32_t num_neighbors, const float* p_values,
                                        const int64_t* p_offsets, uint32_t* mt_key, int32_t* mt_pos, int64_t* out_nbr, int64_t* out_eid,
                                        float* out_t) {
    if (m < 0 || num_rows <= 0) return lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_host: bad sizes");
    if (num_neighbors <= 0) return lstep::set_error(LSTEP_EINVAL, "Number of sampled neighbors for each node should be greater than 0!");
    if (m == 0) return LSTEP_OK;
    if (!indptr || !nbr || !eid || !ts || !node_ids || !times || !mt_key || !mt_pos || !out_nbr || !out_eid || !out_t || (p_values && !p_offsets))
        return lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_host: NULL pointer");
    if (*mt_pos < 0 || *mt_pos > 624) return lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_host: bad generator position");
    Mt19937 rng{mt_key, *mt_pos};
    const int K = num_neighbors;
    double* cdf = nullptr;
    int64_t cdf_cap = 0;
    int rc = LSTEP_OK;
    for (int64_t r = 0; r < m && rc == LSTEP_OK; ++r) {
        const int64_t node = node_ids[r];
        if (node < 0 || node >= num_rows) { rc = lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_host: node id out of range"); break; }
        const int64_t lo = indptr[node];
        const int64_t cnt = count_before(ts, lo, indptr[node + 1], times[r]);
        if (cnt == 0) continue;                     // (utils/utils.py:174: no draw for a node without history)
        int64_t* on = out_nbr + r * K;
        int64_t* oe = out_eid + r * K;
        float* ot = out_t + r * K;
        if (!p_values) {
            // RandomState.randint(0, cnt, size=K): rng = cnt - 1; rng == 0 fills zeros WITHOUT consuming the stream
            const uint64_t range = (uint64_t)(cnt - 1);
            if (range > 0xFFFFFFFEull) { rc = lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_host: history longer than 2^32 - 1"); break; }
            uint32_t mask = (uint32_t)range;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            for (int j = 0; j < K; ++j) {
                uint32_t v = 0;
                if (range != 0) { while ((v = (rng.next() & mask)) > (uint32_t)range) {} }
                on[j] = nbr[lo + v]; oe[j] = eid[lo + v]; ot[j] = (float)ts[lo + v];
            }
        } else {
            const int64_t p0 = p_offsets[r], p1 = p_offsets[r + 1];
            if (p1 - p0 != cnt) { rc = lstep::set_error(LSTEP_EINVAL, "'a' and 'p' must have same size"); break; }
            if (cnt > cdf_cap) {
                delete[] cdf;
                cdf_cap = cnt * 2;
                cdf = new double[cdf_cap];
            }
            double s = 0.0;
            for (int64_t i = 0; i < cnt; ++i) { s += (double)p_values[p0 + i]; cdf[i] = s; }      // p.cumsum() in float64, sequential
            const double last = cdf[cnt - 1];
            for (int64_t i = 0; i < cnt; ++i) cdf[i] /= last;
            for (int j = 0; j < K; ++j) {
                const double u = rng.next_double();
                int64_t a = 0, b = cnt;                 // searchsorted(cdf, u, side='right'): first index with cdf[i] > u
                while (a < b) {
                    const int64_t mid = a + ((b - a) >> 1);
                    if (cdf[mid] <= u) a = mid + 1; else b = mid;
                }
                if (a >= cnt) a = cnt - 1;              // (cannot happen: u < 1 = cdf[cnt - 1]; NaN probabilities are rejected by the caller)
                on[j] = nbr[lo + a]; oe[j] = eid[lo + a]; ot[j] = (float)ts[lo + a];
            }
        }
    }
    delete[] cdf;
    *mt_pos = rng.pos;
    return rc;
}
