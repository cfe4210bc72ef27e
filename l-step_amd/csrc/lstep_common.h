// Shared device helpers for the L-STEP gfx950 kernels.  Wave = 64 lanes everywhere (CDNA4).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lstep_hip.h"

namespace lstep {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;  // 256-thread workgroups: one wave per SIMD
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kMaxTimeDim = 128;   // two time-encoding dims per lane
constexpr int kMaxRowVec = 64;     // one float4 per lane => rows up to 256 floats

int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);

// ---- checked build (-DLSTEP_BOUNDS_CHECK=1, `python tools/build_checked.py`): every id-indexed load of the kernels compares its index with
// the table's row count first.  An index out of range does NOT fault the GPU (it is replaced by row 0, the padding row): the first
// offender is recorded in a sticky device record -- which load (tag), the index, the limit -- that lstep_debug_device_error returns and
// clears, so that ONE ordinary run of the test suite names the kernel (tests/conftest.py checks it behind every GPU test when the
// loaded library is a checked build).  Product builds compile the checks away.
enum CheckTag : int {
    kCheckGatherFwdNode = 1, kCheckGatherFwdNbr = 2, kCheckGatherFwdEdge = 3, kCheckGatherFwdNbrGap = 4,
    kCheckGatherBwdNode = 5, kCheckGatherBwdNbr = 6, kCheckGatherBwdEdge = 7,
    kCheckSampleNode = 8, kCheckFilterNode = 9, kCheckUpdateRowsId = 10, kCheckLossNode = 11, kCheckScatterRowsId = 12,
    kCheckSegmentRow = 13, kCheckSegmentSeg = 14, kCheckRowsById = 15, kCheckOwnerRowsId = 16, kCheckSplicedGradKey = 17,
};
#ifdef LSTEP_BOUNDS_CHECK
// words: [0] tag of the first offender (0 = none), [1] its index, [2] its limit, [3] number of offenders, [4] node-table rows, [5] edge-table
// rows (lstep_debug_set_limits; 0 = unknown: loads without a row count of their own are then not checked).  One pointer per translation
// unit (no relocatable device code in this build), all set to the same buffer by lstep_debug_* in api.hip.
static __device__ unsigned long long* g_check_words = nullptr;
void register_check_setter(void (*fn)(unsigned long long*));
namespace {
struct CheckTu {
    CheckTu() {
        register_check_setter([](unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_check_words), &p, sizeof(p)); });
    }
} g_check_tu;
}  // namespace
__device__ __forceinline__ bool check_index(long long idx, long long limit, int tag) {
    if (limit <= 0 || (idx >= 0 && idx < limit)) return true;
    unsigned long long* w = g_check_words;
    if (w != nullptr) {
        if (atomicCAS(w, 0ull, (unsigned long long)tag) == 0ull) { w[1] = (unsigned long long)idx; w[2] = (unsigned long long)limit; }
        atomicAdd(w + 3, 1ull);
    }
    return false;
}
__device__ __forceinline__ long long check_node_rows() { return g_check_words ? (long long)g_check_words[4] : 0; }
__device__ __forceinline__ long long check_edge_rows() { return g_check_words ? (long long)g_check_words[5] : 0; }
#define LSTEP_CHECKED(idx, limit, tag) (lstep::check_index((long long)(idx), (long long)(limit), (tag)) ? (idx) : 0)
#define LSTEP_NODE_ROWS() lstep::check_node_rows()
#define LSTEP_EDGE_ROWS() lstep::check_edge_rows()
#else
#define LSTEP_CHECKED(idx, limit, tag) (idx)
#define LSTEP_NODE_ROWS() 0
#define LSTEP_EDGE_ROWS() 0
#endif

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// wave index inside the workgroup, as a scalar (the compiler cannot prove threadIdx.x >> 6 is wave-uniform)
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ int bcast_i32(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ int64_t bcast_i64(int64_t v, int src_lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffff), src_lane);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src_lane);
    return ((int64_t)hi << 32) | (int64_t)lo;
}
__device__ __forceinline__ float bcast_f32(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// Make the compiler wait for (and only then forget about) the loads that produced v: an empty asm that consumes it.
__device__ __forceinline__ void settle(int v) { asm volatile("" ::"v"(v)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Number of entries of the non-decreasing array ts[lo, hi) that are strictly smaller than t
// (= np.searchsorted(ts[lo:hi], t), side 'left': reference utils/utils.py:140).  Whole wave cooperates:
// 64 probes per round shrink the range 64x, the last <=64 candidates are compared in one ballot.
__device__ __forceinline__ int64_t wave_count_before(const double* __restrict__ ts, int64_t lo, int64_t hi, double t, int lane) {
    int64_t base = lo;
    int64_t len = hi - lo;
    while (len > kWave) {
        const int64_t stride = (len + kWave - 1) / kWave;
        const int64_t idx = base + (int64_t)(lane + 1) * stride - 1;
        const bool less = (idx < base + len) ? (ts[idx] < t) : false;
        const int c = __popcll(__ballot(less));  // probes are sorted: the 'less' lanes form a prefix
        const int64_t nb = base + (int64_t)c * stride;
        int64_t nl = base + len - nb;
        if (nl > stride - 1) nl = stride - 1;  // probe c itself is >= t (or past the end)
        if (nl < 0) nl = 0;
        base = nb;
        len = nl;
    }
    const bool less = (lane < len) ? (ts[base + lane] < t) : false;
    return (base - lo) + __popcll(__ballot(less));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// read-once streams (history rows, edge-feature rows): non-temporal load, so they do not evict the re-used tables
// (CSR, node features, PE table) from L2 / Infinity Cache.  LSTEP_NO_NT builds (tuning A/B only) fall back to plain loads.
__device__ __forceinline__ float4 ld4_stream(const float* p) {
#ifdef LSTEP_NO_NT
    return ld4(p);
#else
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void fma4(float4& acc, float s, const float4& v) {
    acc.x = fmaf(s, v.x, acc.x);
    acc.y = fmaf(s, v.y, acc.y);
    acc.z = fmaf(s, v.z, acc.z);
    acc.w = fmaf(s, v.w, acc.w);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// float32 time delta exactly as the reference forms it: float64(t) - float64(float32(neighbour time)), then .float()
// (utils/utils.py:166 stores neighbour times as float32; models/LSTEP.py:153,228-230 subtract in float64).
__device__ __forceinline__ float delta_t(double t, double nbr_ts) { return (float)(t - (double)(float)nbr_ts); }

// cos(x) on the hardware's transcendental unit (round 5): v_cos_f32 takes its argument in REVOLUTIONS and is only defined on [-256, 256], so
// the argument is divided by 2 pi and reduced to [-0.5, 0.5] in float64 (exact enough up to 2e9: 22 fraction bits left at the top of the
// range) -- 4 float64 instructions and one 8-cycle transcendental instead of the ~27 instructions of cos_full_range.  Measured over 4 M
// arguments, log-uniform in [1e-6, 2e9] (tools/microbench/vcos_accuracy.hip, profiles/r05_cos_ab.txt): max |error| 2.7e-7, rms 4.3e-8
// (cos_full_range: 9.1e-8 / 1.9e-8).  Used where a kernel is bound by the cosines' issue slots and the result feeds a parameter
// GRADIENT (gather_aggregate_bwd_kernel: 212 -> 177 us alone at 49 152 rows); the forward kernels, whose outputs are held to the
// reference's values, keep cos_full_range.
__device__ __forceinline__ float cos_hw(float x) {
    if (!(fabsf(x) <= 2.0e9f)) return cosf(x);      // (also NaN / inf)
    const double rev = (double)x * 0.15915494309189533577;
    return __builtin_amdgcn_cosf((float)(rev - __builtin_rint(rev)));
}

// cos(x) for any float32 x.  The time encoder's arguments span 1e-6 .. 1e9 inside ONE wave (w_d = 10^(-9 d / (D - 1)), models/modules.py:30),
// so the library cosf runs its small-argument path AND its Payne-Hanek path (integer multi-word multiplies, ~150 instructions, divergent) for
// every wave.  Up to 2e9 a float64 reduction by pi/2 (two-word constant: the reduced argument is exact to 1e-16) is enough; the rest is the
// classic pair of float32 minimax polynomials on [-pi/4, pi/4] (Cephes sinf / cosf coefficients).  ~30 issue slots; measured against
// cosl over 4e7 arguments up to 2e9: max error 9.3e-8 absolute, 1.6 ulp (the library: 2 ulp by specification).
__device__ __forceinline__ float cos_full_range(float x) {
#ifdef LSTEP_LIBRARY_COS   // tuning A/B only
    return cosf(x);
#elif defined(LSTEP_HW_COS)  // tuning A/B only (round 5): every cosine of the library through cos_hw below
    return cos_hw(x);
#else
    if (!(fabsf(x) <= 2.0e9f)) return cosf(x);      // (also NaN / inf)
    const double xd = (double)x;
    const double k = __builtin_rint(xd * 0.63661977236758134308);
    double r = __builtin_fma(k, -1.57079632679489655800e+00, xd);
    r = __builtin_fma(k, -6.12323399573676603587e-17, r);
    const int q = (int)k;
    const float rf = (float)r;
    const float z = rf * rf;
    const float c = fmaf(fmaf(fmaf(2.443315711809948e-05f, z, -1.388731625493765e-03f), z, 4.166664568298827e-02f), z * z, fmaf(-0.5f, z, 1.0f));
    const float s = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, rf, rf);
    const float res = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -res : res;
#endif
}

// (the same element for consumers that feed a gradient sum, not a reference-checked output: see cos_hw)
#ifdef LSTEP_NO_HW_COS   // A/B switch back (build flag)
#define LSTEP_TIME_FEAT_GRAD_COS cos_full_range
#else
#define LSTEP_TIME_FEAT_GRAD_COS cos_hw
#endif

// TimeEncoder element: cos(dt * w + b) in float32 (models/modules.py:37).  Full-range: arguments reach 1e9.
#ifdef LSTEP_ABLATE_COS  // tuning experiment only: price of the cosine (never defined in product builds)
__device__ __forceinline__ float time_feat(float dt, float w, float b) { return fmaf(dt, w, b); }
#else
__device__ __forceinline__ float time_feat(float dt, float w, float b) { return cos_full_range(fmaf(dt, w, b)); }
#endif
__device__ __forceinline__ float time_feat_grad(float dt, float w, float b) { return LSTEP_TIME_FEAT_GRAD_COS(fmaf(dt, w, b)); }

}  // namespace lstep
