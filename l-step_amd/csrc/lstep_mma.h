// fp32 matrix-core building blocks shared by the dense kernels (tail.hip, head.hip): every product is formed TRANSPOSED,
// Y^T = W X^T with v_mfma_f32_16x16x4_f32, a wave owning S slabs of 16 activation rows.
//   * operands straight from row-major memory: lane l = (i = l & 15, g = l >> 4) loads the float4 W[n0 + i][k0 + 4g .. +3] and
//     X[r0 + i][k0 + 4g .. +3]; component v of both is the (A, B) pair of the MFMA that contracts k in {k0 + 4g' + v}; four MFMAs
//     cover the 16-wide k chunk, in a permuted but consistent k order.  No LDS, no transposes.
//   * an accumulator tile IS the next product's B operand: the C/D layout puts feature 16t + 4g + v of row i in register v of
//     lane (i, g) -- exactly the B element the MFMA "v" above wants for k = 16t + 4g + v.
#pragma once

#include "lstep_common.h"

namespace lstep {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// tanh on the hardware exp2 / rcp: 1 - 2 / (1 + e^{2|x|}) for |x| >= 0.25 (both 1 ulp: absolute error <= 1.5e-7, +-1 at the ends without
// special cases), the odd Taylor polynomial through x^7 below (truncation <= 8e-8 at 0.25).  Branch-free, 16 instructions; tanhf is ~45 with
// two divergent branches, and update_pe evaluates 50 M of them per step at 1 M nodes -- on waves that hold a whole SIMD to themselves.
// LSTEP_EXACT_TANH=1 (an A/B BUILD, `lstep_amd._native.build_library(defines=["LSTEP_EXACT_TANH=1"], lib_path=...)`, loaded through LSTEP_LIB):
// libm's tanhf in every place the fast form is used -- tools/tanh_drift.py runs long training traces on both builds and reports how far the
// PE tables drift apart (the fast form's 1.5e-7 per evaluation compounds through `pe += tanh(...)` across steps).
__device__ __forceinline__ float tanh_fast(float x) {
#ifdef LSTEP_EXACT_TANH
    return tanhf(x);
#endif
    const float ax = fabsf(x), x2 = x * x;
    const float t = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);            // e^{2|x|}
    const float big = copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + t), x);
    const float small = x + x * x2 * fmaf(x2, fmaf(x2, -0.053968253968f, 0.133333333333f), -0.333333333333f);
    return ax < 0.25f ? small : big;
}

// One 16-wide k chunk of operands: A tiles (weights) and B slabs (activation rows), one float4 per lane each.
template <int T, int S>
struct Chunk {
    f32x4 a[T];
    f32x4 b[S];
};

// acc[t][s] += sum_k W[16 t + i][k] * X_s[row][k]  over `chunks` 16-wide k chunks.
//   wl = W + this lane's (i * ldw + 4 g) (already offset to the first tile / first k);  xl[s] = X + row_s * ldx + 4 g + first k.
// Double-buffered by hand: the loads of chunk c + 1 are issued before the MFMAs of chunk c (the sched barriers keep them
// there); one chunk is T * S * 4 MFMAs = 128 T S cycles, enough to cover an L2 hit and most HBM misses.
template <int T, int S>
__device__ __forceinline__ void mma_wx(f32x4 (*acc)[S], const float* wl, int ldw, int chunks, const float* const (&xl)[S]) {
    auto load = [&](Chunk<T, S>& o, int c) {
#pragma unroll
        for (int s = 0; s < S; ++s) o.b[s] = ldv4(xl[s] + 16 * c);
#pragma unroll
        for (int t = 0; t < T; ++t) o.a[t] = ldv4(wl + (size_t)16 * t * ldw + 16 * c);
    };
    auto run = [&](const Chunk<T, S>& o) {
        // v outermost: back-to-back MFMAs on one accumulator would pay the 40-cycle dependent latency instead of the 32-cycle issue
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) acc[t][s] = mfma4(o.a[t][v], o.b[s][v], acc[t][s]);
            }
        }
    };
    Chunk<T, S> c0, c1;
    load(c0, 0);
    for (int c = 0; c < chunks; c += 2) {
        const bool two = c + 1 < chunks;   // uniform
        if (two) load(c1, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        run(c0);
        __builtin_amdgcn_sched_barrier(0);
        if (two) {
            if (c + 2 < chunks) load(c0, c + 2);
            __builtin_amdgcn_sched_barrier(0);
            run(c1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// mma_wx for an operand X whose rows hold only `kvalid` (a multiple of 4) meaningful columns from xl on: the lanes of the last chunk that
// would read past them contribute zeros instead (the matching weight columns are zero padding; what lies behind X's columns may be anything,
// NaN included).  g = lane >> 4.
template <int T, int S>
__device__ __forceinline__ void mma_wx_masked(f32x4 (*acc)[S], const float* wl, int ldw, int chunks, const float* const (&xl)[S], int kvalid, int g) {
    auto load = [&](Chunk<T, S>& o, int c) {
        const bool ok = 16 * c + 4 * g + 4 <= kvalid;
#pragma unroll
        for (int s = 0; s < S; ++s) o.b[s] = ok ? ldv4(xl[s] + 16 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < T; ++t) o.a[t] = ldv4(wl + (size_t)16 * t * ldw + 16 * c);
    };
    auto run = [&](const Chunk<T, S>& o) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) acc[t][s] = mfma4(o.a[t][v], o.b[s][v], acc[t][s]);
            }
        }
    };
    Chunk<T, S> c0, c1;
    load(c0, 0);
    for (int c = 0; c < chunks; c += 2) {
        const bool two = c + 1 < chunks;
        if (two) load(c1, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        run(c0);
        __builtin_amdgcn_sched_barrier(0);
        if (two) {
            if (c + 2 < chunks) load(c0, c + 2);
            __builtin_amdgcn_sched_barrier(0);
            run(c1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// acc[t][s] += sum over the TK register tiles r[tk][s] (16 features each) of W[16 t + i][16 tk + k] * r
template <int T, int S, int TK>
__device__ __forceinline__ void mma_wr(f32x4 (*acc)[S], const float* wl, int ldw, const f32x4 (*r)[S]) {
    f32x4 a[2][T];
#pragma unroll
    for (int t = 0; t < T; ++t) a[0][t] = ldv4(wl + (size_t)16 * t * ldw);
#pragma unroll
    for (int tk = 0; tk < TK; ++tk) {
        if (tk + 1 < TK) {
#pragma unroll
            for (int t = 0; t < T; ++t) a[(tk + 1) & 1][t] = ldv4(wl + (size_t)16 * t * ldw + 16 * (tk + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) acc[t][s] = mfma4(a[tk & 1][t][v], r[tk][s][v], acc[t][s]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace lstep
