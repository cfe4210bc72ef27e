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

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

#include "lstep_common.h"

namespace {

// One state block: the 624 state words advance in place, their tempered outputs go to `out`.  Plain loops the compiler vectorises (the
// recurrences reach 1 word ahead and 227 words back: no dependence inside a vector); the second copy is the same body compiled for AVX2,
// picked at run time -- the generator is the sequential part of the RNG-defined strategies, ~1.4 raw words per accepted draw.
#define LSTEP_MT_BLOCK_BODY                                                                           \
    const uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrix = 0x9908b0dfu;                \
    for (int i = 0; i < 624 - 397; ++i) {                                                             \
        const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);                                 \
        key[i] = key[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix);                               \
    }                                                                                                 \
    for (int i = 624 - 397; i < 623; ++i) {                                                           \
        const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);                                 \
        key[i] = key[i - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix);                               \
    }                                                                                                 \
    {                                                                                                 \
        const uint32_t y = (key[623] & kUpper) | (key[0] & kLower);                                   \
        key[623] = key[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix);                                 \
    }                                                                                                 \
    for (int i = 0; i < 624; ++i) {                                                                   \
        uint32_t y = key[i];                                                                          \
        y ^= (y >> 11);                                                                               \
        y ^= (y << 7) & 0x9d2c5680u;                                                                  \
        y ^= (y << 15) & 0xefc60000u;                                                                 \
        y ^= (y >> 18);                                                                               \
        out[i] = y;                                                                                   \
    }

void mt_next_block_generic(uint32_t* __restrict key, uint32_t* __restrict out) { LSTEP_MT_BLOCK_BODY }
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) void mt_next_block_avx2(uint32_t* __restrict key, uint32_t* __restrict out) { LSTEP_MT_BLOCK_BODY }
const bool g_have_avx2 = __builtin_cpu_supports("avx2") && getenv("LSTEP_NO_AVX2") == nullptr;
const bool g_have_avx512 = __builtin_cpu_supports("avx512f") && getenv("LSTEP_NO_AVX2") == nullptr && getenv("LSTEP_NO_AVX512") == nullptr;
// w raw words -> the accepted ones ((word & mask) <= range), packed from dst on; returns how many.  Sixteen at a time: compare, compress in
// the register, store all 16 lanes (so up to 15 words behind the accepted ones are scribbled on: the caller's rows are filled front to back
// and the scratch ends in 64 spare bytes).
__attribute__((target("avx512f"))) int masked_compact_avx512(const uint32_t* src, int w, uint32_t mask, uint32_t range, uint32_t* dst) {
    const __m512i vm = _mm512_set1_epi32((int)mask), vr = _mm512_set1_epi32((int)range);
    int j = 0, i = 0;
    for (; i + 16 <= w; i += 16) {
        const __m512i v = _mm512_and_si512(_mm512_loadu_si512((const void*)(src + i)), vm);
        const __mmask16 k = _mm512_cmple_epu32_mask(v, vr);
        _mm512_storeu_si512((void*)(dst + j), _mm512_maskz_compress_epi32(k, v));
        j += __builtin_popcount((unsigned)k);
    }
    for (; i < w; ++i) {
        const uint32_t v = src[i] & mask;
        dst[j] = v;
        j += (v <= range) ? 1 : 0;
    }
    return j;
}
#else
const bool g_have_avx512 = false;
int masked_compact_avx512(const uint32_t*, int, uint32_t, uint32_t, uint32_t*) { return 0; }
void mt_next_block_avx2(uint32_t* key, uint32_t* out) { mt_next_block_generic(key, out); }
const bool g_have_avx2 = false;
#endif

struct Mt19937 {
    uint32_t* key;   // [624]: numpy's state words (untempered)
    int pos;         // next word to hand out; 624 = regenerate first
    uint32_t out[624];   // the tempered outputs of the current state block
    Mt19937(uint32_t* k, int p) : key(k), pos(p) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out[i] = y;
        }
    }
    void refill() {
        if (g_have_avx2) mt_next_block_avx2(key, out); else mt_next_block_generic(key, out);      // (a 512-bit copy measured slower)
        pos = 0;
    }
    uint32_t next() {
        if (pos == 624) refill();
        return out[pos++];
    }
    double next_double() {
        const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // n accepted values of `next() & mask` that are <= range (randint's masked rejection), block-wise and branch-free: a block of w raw words
    // never accepts more than w <= n - j values, so exactly the words the one-by-one loop would have consumed are consumed.  dst needs 15
    // words of slack behind its n (the AVX-512 compaction stores whole vectors).
    void masked_fill(uint32_t mask, uint32_t range, uint32_t* dst, int n) {
        int j = 0;
        while (j < n) {
            if (pos == 624) refill();
            const int w = (624 - pos) < (n - j) ? (624 - pos) : (n - j);
            const uint32_t* src = out + pos;
            if (g_have_avx512) {
                j += masked_compact_avx512(src, w, mask, range, dst + j);
            } else {
                for (int i = 0; i < w; ++i) {
                    const uint32_t v = src[i] & mask;
                    dst[j] = v;
                    j += (v <= range) ? 1 : 0;
                }
            }
            pos += w;
        }
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
    Mt19937 rng(mt_key, *mt_pos);
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


// ---- the same draws with the per-row re-sort done here (round 5) ---------------------------------------------------------------------------
// utils/utils.py:192-196 re-orders the K sampled slots of a row by their float32 time with numpy's unstable argsort.  The node's history is
// time-sorted (utils/utils.py:120-127), float32 rounding is monotone, so ordering the slots by time IS ordering the picked positions
// v_0 .. v_{K-1} (integers below cnt) -- except among DISTINCT positions whose float32 times are equal, where the order is whatever numpy's sort
// does (introsort or a SIMD sort, by build and CPU).  Equal positions are the same interaction: any order writes the same triple.  So:
//   phase A (one thread: the generator's stream is sequential, and the masked rejection makes its consumption data dependent) only draws --
//           positions (uniform) or uniform doubles (weighted) -- into a scratch array;
//   phase B (num_threads workers over blocks of rows) builds the weighted rows' cdf and searches it, sorts the positions (a counting sort over
//           [0, cnt) when the history is short against K, std::sort otherwise), gathers the triples in sorted order and checks neighbouring
//           distinct positions for equal float32 times.  A row with such a tie is written in DRAW order instead and flagged in ambiguous[r]:
//           the caller sorts exactly those rows with numpy, as before.
// Rows without history are left untouched and flagged 0.
namespace {

std::mutex g_scratch_mutex;
void* g_scratch = nullptr;
size_t g_scratch_bytes = 0;
const bool g_timing = getenv("LSTEP_RNG_TIMING") != nullptr;

struct RowJob {
    const int64_t* nbr; const int64_t* eid; const double* ts;
    int K;
    int64_t* out_nbr; int64_t* out_eid; float* out_t; uint8_t* ambiguous;
};

struct SortScratch {
    std::vector<uint32_t> hist, sorted;
    std::vector<double> cdf;
};

inline void write_draw_order(const RowJob& j, int64_t r, int64_t lo, const uint32_t* pick) {
    int64_t* on = j.out_nbr + r * j.K; int64_t* oe = j.out_eid + r * j.K; float* ot = j.out_t + r * j.K;
    for (int i = 0; i < j.K; ++i) { const int64_t s = lo + pick[i]; on[i] = j.nbr[s]; oe[i] = j.eid[s]; ot[i] = (float)j.ts[s]; }
}

// positions -> the row's K output slots in time order; false = a tie among distinct positions (nothing useful written)
inline bool write_sorted(const RowJob& j, int64_t r, int64_t lo, int64_t cnt, const uint32_t* pick, SortScratch& sc) {
    const int K = j.K;
    int64_t* on = j.out_nbr + r * K; int64_t* oe = j.out_eid + r * K; float* ot = j.out_t + r * K;
    if (cnt <= (int64_t)2 * K + 4096) {
        if ((int64_t)sc.hist.size() < cnt) sc.hist.resize((size_t)cnt * 2);
        uint32_t* h = sc.hist.data();
        memset(h, 0, (size_t)cnt * sizeof(uint32_t));
        for (int i = 0; i < K; ++i) ++h[pick[i]];
        int o = 0;
        bool have_prev = false;
        float prev = 0.f;
        for (int64_t v = 0; v < cnt; ++v) {
            const uint32_t c = h[v];
            if (!c) continue;
            const int64_t s = lo + v;
            const int64_t n = j.nbr[s], e = j.eid[s];
            const float t = (float)j.ts[s];
            if (have_prev && t == prev) return false;
            have_prev = true; prev = t;
            for (uint32_t q = 0; q < c; ++q, ++o) { on[o] = n; oe[o] = e; ot[o] = t; }
        }
        return true;
    }
    if ((int)sc.sorted.size() < K) sc.sorted.resize(K);
    uint32_t* sv = sc.sorted.data();
    memcpy(sv, pick, (size_t)K * sizeof(uint32_t));
    std::sort(sv, sv + K);
    for (int i = 0; i < K; ++i) {
        const int64_t s = lo + sv[i];
        const float t = (float)j.ts[s];
        if (i > 0 && sv[i] != sv[i - 1] && t == ot[i - 1]) return false;
        on[i] = j.nbr[s]; oe[i] = j.eid[s]; ot[i] = t;
    }
    return true;
}

}  // namespace

extern "C" int lstep_sample_random_sorted_host(const int64_t* indptr, const int64_t* nbr, const int64_t* eid, const double* ts, int64_t num_rows,
                                               const int64_t* node_ids, const double* times, int64_t m, int32_t num_neighbors,
                                               const float* p_values, const int64_t* p_offsets, uint32_t* mt_key, int32_t* mt_pos,
                                               int64_t* out_nbr, int64_t* out_eid, float* out_t, uint8_t* ambiguous, int32_t num_threads) {
    if (m < 0 || num_rows <= 0) return lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_sorted_host: bad sizes");
    if (num_neighbors <= 0) return lstep::set_error(LSTEP_EINVAL, "Number of sampled neighbors for each node should be greater than 0!");
    if (m == 0) return LSTEP_OK;
    if (!indptr || !nbr || !eid || !ts || !node_ids || !times || !mt_key || !mt_pos || !out_nbr || !out_eid || !out_t || !ambiguous ||
        (p_values && !p_offsets))
        return lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_sorted_host: NULL pointer");
    if (*mt_pos < 0 || *mt_pos > 624) return lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_sorted_host: bad generator position");
    const int K = num_neighbors;
    const bool weighted = p_values != nullptr;
    // the draw scratch: positions (uniform, uint32 [m, K]) or uniforms (weighted, double [m, K]); one process-wide block is kept between calls
    // (up to 256 MB: a fresh 100-200 MB block would be page-faulted in by the drawing thread on every call), taken by whoever gets the lock
    std::vector<int64_t> cnts;
    const size_t need = (size_t)m * K * (weighted ? sizeof(double) : sizeof(uint32_t)) + 64;      // (+64: masked_fill's vector stores)
    std::unique_lock<std::mutex> keep(g_scratch_mutex, std::try_to_lock);
    void* owned = nullptr;
    void* scratch = nullptr;
    try {
        cnts.assign((size_t)m, 0);
    } catch (const std::bad_alloc&) {
        return lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_sorted_host: out of host memory");
    }
    if (keep.owns_lock() && need <= ((size_t)256 << 20)) {
        if (g_scratch_bytes < need) {
            free(g_scratch);
            g_scratch = malloc(need);
            g_scratch_bytes = g_scratch ? need : 0;
        }
        scratch = g_scratch;
    } else {
        scratch = owned = malloc(need);
    }
    if (!scratch) return lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_sorted_host: out of host memory for the draw scratch");
    struct FreeOwned { void* p; ~FreeOwned() { free(p); } } free_owned{owned};
    uint32_t* const picks = (uint32_t*)scratch;
    double* const us = (double*)scratch;
    const auto t_start = std::chrono::steady_clock::now();
    // ---- phase B: search / sort / gather, rows in parallel; the workers start BEFORE phase A and take each block of rows as soon as its
    //      draws are published (rows_ready), so the two phases overlap and the call costs about the longer of the two
    const RowJob job{nbr, eid, ts, K, out_nbr, out_eid, out_t, ambiguous};
    std::atomic<int64_t> next_block{0}, rows_ready{0};
    std::atomic<int> failed{0};
    std::mutex ready_mutex;
    std::condition_variable ready_cv;
    const int64_t kBlock = 32;
    auto publish = [&](int64_t rows_done) {          // rows [0, rows_done) are drawn; wake the workers once per block of rows
        rows_ready.store(rows_done, std::memory_order_release);
        if (rows_done % kBlock == 0 || rows_done == m) {
            { std::lock_guard<std::mutex> lk(ready_mutex); }
            ready_cv.notify_all();
        }
    };
    auto worker = [&]() {
        SortScratch sc;
        std::vector<uint32_t> local;
        if (weighted) local.resize(K);
        for (;;) {
            const int64_t b = next_block.fetch_add(1, std::memory_order_relaxed);
            const int64_t r0 = b * kBlock;
            if (r0 >= m) break;
            const int64_t r1 = std::min(m, r0 + kBlock);
            if (rows_ready.load(std::memory_order_acquire) < r1 && !failed.load(std::memory_order_relaxed)) {
                std::unique_lock<std::mutex> lk(ready_mutex);           // (blocks: a spinning worker would eat the drawing thread's CPU share)
                ready_cv.wait(lk, [&] { return rows_ready.load(std::memory_order_acquire) >= r1 || failed.load(std::memory_order_relaxed); });
            }
            if (failed.load(std::memory_order_relaxed)) break;
            for (int64_t r = r0; r < r1; ++r) {
                const int64_t cnt = cnts[r];
                ambiguous[r] = 0;
                if (cnt == 0) continue;
                const int64_t lo = indptr[node_ids[r]];
                const uint32_t* pk;
                if (weighted) {
                    if ((int64_t)sc.cdf.size() < cnt) sc.cdf.resize((size_t)cnt * 2);
                    double* cdf = sc.cdf.data();
                    const float* p = p_values + p_offsets[r];
                    double s = 0.0;
                    for (int64_t i = 0; i < cnt; ++i) { s += (double)p[i]; cdf[i] = s; }          // p.cumsum() in float64, sequential
                    const double last = cdf[cnt - 1];
                    for (int64_t i = 0; i < cnt; ++i) cdf[i] /= last;
                    const double* u = us + (size_t)r * K;
                    for (int j = 0; j < K; ++j) {
                        int64_t a = 0, bb = cnt;             // searchsorted(cdf, u, side='right')
                        while (a < bb) {
                            const int64_t mid = a + ((bb - a) >> 1);
                            if (cdf[mid] <= u[j]) a = mid + 1; else bb = mid;
                        }
                        if (a >= cnt) a = cnt - 1;
                        local[j] = (uint32_t)a;
                    }
                    pk = local.data();
                } else {
                    pk = picks + (size_t)r * K;
                }
                if (!write_sorted(job, r, lo, cnt, pk, sc)) {
                    write_draw_order(job, r, lo, pk);
                    ambiguous[r] = 1;
                }
            }
        }
    };
    int T = num_threads > 0 ? num_threads : (int)std::thread::hardware_concurrency();
    T = std::max(1, std::min(T, 64));
    T = (int)std::min<int64_t>(T, (m + kBlock - 1) / kBlock);
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < T; ++t) pool.emplace_back(worker);
    } catch (...) {}                              // (no more threads to be had: the ones that started, and this one, finish the rows)
    // ---- phase A: the draws, in row order, on the caller's generator
    Mt19937 rng(mt_key, *mt_pos);
    int rc = LSTEP_OK;
    for (int64_t r = 0; r < m; ++r) {
        const int64_t node = node_ids[r];
        if (node < 0 || node >= num_rows) { rc = lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_sorted_host: node id out of range"); break; }
        const int64_t lo = indptr[node];
        const int64_t cnt = count_before(ts, lo, indptr[node + 1], times[r]);
        cnts[r] = cnt;
        if (cnt == 0) { publish(r + 1); continue; }
        if (!weighted) {
            const uint64_t range = (uint64_t)(cnt - 1);
            if (range > 0xFFFFFFFEull) { rc = lstep::set_error(LSTEP_EINVAL, "lstep_sample_random_sorted_host: history longer than 2^32 - 1"); break; }
            uint32_t mask = (uint32_t)range;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            uint32_t* pk = picks + (size_t)r * K;
            if (range == 0) memset(pk, 0, (size_t)K * sizeof(uint32_t));       // (nothing consumed)
            else rng.masked_fill(mask, (uint32_t)range, pk, K);
        } else {
            if (p_offsets[r + 1] - p_offsets[r] != cnt) { rc = lstep::set_error(LSTEP_EINVAL, "'a' and 'p' must have same size"); break; }
            double* u = us + (size_t)r * K;
            for (int j = 0; j < K; ++j) u[j] = rng.next_double();
        }
        publish(r + 1);
    }
    *mt_pos = rng.pos;
    const auto t_drawn = std::chrono::steady_clock::now();
    if (rc != LSTEP_OK) {
        failed.store(1);
        { std::lock_guard<std::mutex> lk(ready_mutex); }
        ready_cv.notify_all();
    } else worker();                                // (this thread joins in on what is left)
    for (auto& th : pool) th.join();
    if (g_timing) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[lstep rng] rows %lld x %d%s: draws %.2f ms, tail of sort/gather %.2f ms (%d threads)\n", (long long)m, K,
                weighted ? " weighted" : "", std::chrono::duration<double, std::milli>(t_drawn - t_start).count(),
                std::chrono::duration<double, std::milli>(t_end - t_drawn).count(), (int)pool.size() + 1);
    }
    return rc;
}
