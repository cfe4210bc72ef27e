// The dense tail after the gather stage as ONE kernel per direction:
//
//   h1  = relu(W1 x_edge + b1)                         edge_mlp_1 with edge_agg folded in        (models/LSTEP.py:161-166)
//   p1  = relu(Wn1 x_pe + bn1)                         pe_neighbor_mlp_1                          (:240-242)
//   q   = own + tanh(Wq [own ; p1] + bq)               self_update_neighbor_pe + pe_neighbor_mlp_2, tanh, residual (:243-247)
//   out = Wall [x_node ; h1 ; q] + ball                edge_mlp_2 -> node_mlp -> out_node_emb, pre-multiplied (:170,219,264)
//
// (the composed, 16-padded weights come from the host layer, see model.py:_TailWeights).  Everything is computed TRANSPOSED on the
// fp32 matrix cores: a wave owns S slabs of 16 rows and forms Y^T = W X^T with v_mfma_f32_16x16x4_f32, A = a 16 x 4 block of W,
// B = a 4 x 16 block of X^T.  Two properties of that instruction make the chain register-only:
//   * operands straight from row-major memory: lane l = (i = l & 15, g = l >> 4) loads the float4 W[n0 + i][k0 + 4g .. +3] and
//     X[r0 + i][k0 + 4g .. +3]; component v of both is the (A, B) pair of the MFMA that contracts k in {k0 + 4g' + v};
//     four MFMAs cover the 16-wide k chunk, in a permuted but consistent k order.  No LDS, no transposes.
//   * an accumulator tile IS the next product's B operand: the C/D layout puts feature 16t + 4g + v of row i in register v of
//     lane (i, g) -- exactly the B element the MFMA "v" above wants for k = 16t + 4g + v.  So h1, p1 and q never leave the
//     registers between the layers (they are also written out once, for the backward pass).
// The weights stream from L2 (1.2 MB, shared by every wave); the activations are read once from HBM.
#include <stdlib.h>
#include <type_traits>

#include "lstep_mma.h"

namespace lstep {

constexpr int kCe = 272, kPp = 176, kFn = 176;        // 16-aligned widths: edge / PE-aggregate channel (172 + 100), PE, node / out
constexpr int kTe = kCe / 16, kTp = kPp / 16, kTn = kFn / 16;
constexpr int kC1 = kFn + kCe + kPp;                  // [x_node | h1 | q]   624
constexpr int kC2 = 2 * kPp;                          // [own | p1]          352

struct TailFwdParams {
    const float* xe;    // [m, ld_e]  x_edge  (gather output, kCe used)
    const float* xp;    // [m, ld_p]  x_pe    (gather output, kCe used)
    float* c1;          // [m, kC1]   [x_node (in) | h1 (out) | q (out)]
    float* c2;          // [m, kC2]   [own (in) | p1 (out)]
    float* out;         // [m, kFn]
    const float *w1, *b1, *wn1, *bn1, *wq, *bq, *wall, *ball;
    int64_t m;
    int32_t ld_e, ld_p;
};

template <int S>
__global__ __launch_bounds__(kBlock, 1) void tail_fwd_kernel(const TailFwdParams p) {
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const int64_t task = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t r0 = task * (16 * S);
    if (r0 >= p.m) return;   // no barriers in this kernel

    // per-slab row of this lane (clamped: rows past m compute on a valid row and are not stored)
    int64_t row[S];
    bool live[S];
    const float *xe_l[S], *xp_l[S], *c1_l[S], *c2_l[S];   // this lane's B-operand position in each activation matrix
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int64_t r = r0 + 16 * s + i;
        live[s] = r < p.m;
        if (!live[s]) r = p.m - 1;
        row[s] = r;
        xe_l[s] = p.xe + r * p.ld_e + 4 * g;
        xp_l[s] = p.xp + r * p.ld_p + 4 * g;
        c1_l[s] = p.c1 + r * kC1 + 4 * g;
        c2_l[s] = p.c2 + r * kC2 + 4 * g;
    }
    // this lane's position in a weight matrix with row stride ldw: row i of a 16-row tile, k = 4 g
    auto wlane = [&](const float* w, int ldw) { return w + i * ldw + 4 * g; };
    // bias in C layout: features 16 t + 4 g .. + 3
    auto bias_tile = [&](const float* b, int t) { return *reinterpret_cast<const f32x4*>(b + 16 * t + 4 * g); };

    // ---- p1 = relu(Wn1 x_pe + bn1) ------------------------------------------------------------------------------------------
    f32x4 p1[kTp][S];
#pragma unroll
    for (int t = 0; t < kTp; ++t) {
        const f32x4 bv = bias_tile(p.bn1, t);
#pragma unroll
        for (int s = 0; s < S; ++s) p1[t][s] = bv;
    }
    mma_wx<kTp, S>(p1, wlane(p.wn1, kCe), kCe, kTe, xp_l);
#pragma unroll
    for (int t = 0; t < kTp; ++t) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int v = 0; v < 4; ++v) p1[t][s][v] = fmaxf(p1[t][s][v], 0.f);
            if (live[s]) *reinterpret_cast<f32x4*>(p.c2 + row[s] * kC2 + kPp + 16 * t + 4 * g) = p1[t][s];
        }
    }

    // ---- q = own + tanh(Wq [own ; p1] + bq) ---------------------------------------------------------------------------------
    f32x4 q[kTp][S];
#pragma unroll
    for (int t = 0; t < kTp; ++t) {
        const f32x4 bv = bias_tile(p.bq, t);
#pragma unroll
        for (int s = 0; s < S; ++s) q[t][s] = bv;
    }
    mma_wx<kTp, S>(q, wlane(p.wq, kC2), kC2, kTp, c2_l);
    mma_wr<kTp, S, kTp>(q, wlane(p.wq + kPp, kC2), kC2, p1);
    {   // the own rows as ONE batch of loads (p1 is dead: its registers take them).  Loaded tile by tile, each load waits out a full memory
        // latency behind the previous tile's store -- the compiler may not hoist a load over a store that might alias, and the wave has its
        // SIMD to itself: nothing covers the wait.
        f32x4 own[kTp][S];
#pragma unroll
        for (int t = 0; t < kTp; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s) own[t][s] = ldv4(p.c2 + row[s] * kC2 + 16 * t + 4 * g);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kTp; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int v = 0; v < 4; ++v) q[t][s][v] = own[t][s][v] + tanh_fast(q[t][s][v]);
                if (live[s]) *reinterpret_cast<f32x4*>(p.c1 + row[s] * kC1 + kFn + kCe + 16 * t + 4 * g) = q[t][s];
            }
        }
    }

    // ---- out = Wall [x_node ; h1 ; q] + ball, h1 = relu(W1 x_edge + b1) produced 6 (5) tiles at a time -------------------------
    f32x4 o[kTn][S];
#pragma unroll
    for (int t = 0; t < kTn; ++t) {
        const f32x4 bv = bias_tile(p.ball, t);
#pragma unroll
        for (int s = 0; s < S; ++s) o[t][s] = bv;
    }
    mma_wx<kTn, S>(o, wlane(p.wall, kC1), kC1, kTn, c1_l);
    mma_wr<kTn, S, kTp>(o, wlane(p.wall + kFn + kCe, kC1), kC1, q);
    auto h1_group = [&](auto group_size, int t0) {   // h1 tiles [t0, t0 + G): produce, store, fold into out
        constexpr int G = decltype(group_size)::value;
        f32x4 h[G][S];
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const f32x4 bv = bias_tile(p.b1, t0 + t);
#pragma unroll
            for (int s = 0; s < S; ++s) h[t][s] = bv;
        }
        mma_wx<G, S>(h, wlane(p.w1 + (size_t)16 * t0 * kCe, kCe), kCe, kTe, xe_l);
#pragma unroll
        for (int t = 0; t < G; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int v = 0; v < 4; ++v) h[t][s][v] = fmaxf(h[t][s][v], 0.f);
                if (live[s]) *reinterpret_cast<f32x4*>(p.c1 + row[s] * kC1 + kFn + 16 * (t0 + t) + 4 * g) = h[t][s];
            }
        }
        mma_wr<kTn, S, G>(o, wlane(p.wall + kFn + 16 * t0, kC1), kC1, h);
    };
    static_assert(kTe == 17, "h1 tile groups 6 + 6 + 5");
    h1_group(std::integral_constant<int, 6>{}, 0);
    h1_group(std::integral_constant<int, 6>{}, 6);
    h1_group(std::integral_constant<int, 5>{}, 12);
#pragma unroll
    for (int t = 0; t < kTn; ++t) {
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (live[s]) *reinterpret_cast<f32x4*>(p.out + row[s] * kFn + 16 * t + 4 * g) = o[t][s];
    }
}

// ---- backward: the same chain run through the transposed weights ---------------------------------------------------------------
//   d_q  = WallT[q rows] g            d_z = d_q * (1 - tanh^2)          d_own = d_q + WqT[own rows] d_z
//   d_p1 = WqT[p1 rows] d_z * [p1 > 0]                                   d_xpe = Wn1T d_p1
//   d_h1 = WallT[h1 rows] g * [h1 > 0]                                   d_xedge = W1T d_h1
// (xT = the [in, out] row-major transpose of a forward weight, so every product has the forward's operand layout).  d_z, d_p1, d_h1
// are also written out: they are the dY operands of the four weight-gradient products (lstep_linear_wgrad).
struct TailBwdParams {
    const float* g;      // [m, kFn]  gradient of out
    const float* c1;     // [m, kC1]  [x_node | h1 | q]
    const float* c2;     // [m, kC2]  [own | p1]
    const float *w1t, *wn1t, *wqt, *wallt;   // [272,272] [272,176] [352,176] [624,176]
    float* dxe;          // [m, kCe]
    float* dxp;          // [m, kCe]
    float* down;         // [m, ld_down] (kPp used)
    float* dh1;          // [m, kCe]
    float* dp1;          // [m, kPp]
    float* dz;           // [m, kPp]
    int64_t m;
    int32_t ld_down;
};

template <int S>
__global__ __launch_bounds__(kBlock, 1) void tail_bwd_kernel(const TailBwdParams p) {
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const int64_t task = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t r0 = task * (16 * S);
    if (r0 >= p.m) return;   // no barriers in this kernel

    int64_t row[S];
    bool live[S];
    const float* g_l[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int64_t r = r0 + 16 * s + i;
        live[s] = r < p.m;
        if (!live[s]) r = p.m - 1;
        row[s] = r;
        g_l[s] = p.g + r * kFn + 4 * g;
    }
    auto wlane = [&](const float* w, int ldw) { return w + i * ldw + 4 * g; };
    auto zero = [&](f32x4 (*a)[S], int tiles) {
#pragma unroll
        for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s) a[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    {   // ---- PE channel ---------------------------------------------------------------------------------------------------------
        f32x4 dq[kTp][S], dz[kTp][S];
        zero(dq, kTp);
        mma_wx<kTp, S>(dq, wlane(p.wallt + (size_t)(kFn + kCe) * kFn, kFn), kFn, kTn, g_l);
        {   // q and own as ONE batch of loads (q lands in dz's registers), then the arithmetic and the stores: see tail_fwd_kernel
            f32x4 ov[kTp][S];
#pragma unroll
            for (int t = 0; t < kTp; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    dz[t][s] = ldv4(p.c1 + row[s] * kC1 + kFn + kCe + 16 * t + 4 * g);
                    ov[t][s] = ldv4(p.c2 + row[s] * kC2 + 16 * t + 4 * g);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < kTp; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float th = dz[t][s][v] - ov[t][s][v];   // tanh(z), from q = own + tanh(z)
                        dz[t][s][v] = dq[t][s][v] * (1.f - th * th);
                    }
                    if (live[s]) *reinterpret_cast<f32x4*>(p.dz + row[s] * kPp + 16 * t + 4 * g) = dz[t][s];
                }
            }
        }
        mma_wr<kTp, S, kTp>(dq, wlane(p.wqt, kPp), kPp, dz);   // d_own = d_q + WqT[own rows] d_z
#pragma unroll
        for (int t = 0; t < kTp; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (live[s]) *reinterpret_cast<f32x4*>(p.down + row[s] * p.ld_down + 16 * t + 4 * g) = dq[t][s];
        }
        zero(dq, kTp);                                          // now d_p1
        mma_wr<kTp, S, kTp>(dq, wlane(p.wqt + (size_t)kPp * kPp, kPp), kPp, dz);
        {   // p1 as one batch of loads (dz is dead)
            f32x4 pv[kTp][S];
#pragma unroll
            for (int t = 0; t < kTp; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) pv[t][s] = ldv4(p.c2 + row[s] * kC2 + kPp + 16 * t + 4 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < kTp; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) dq[t][s][v] = pv[t][s][v] > 0.f ? dq[t][s][v] : 0.f;
                    if (live[s]) *reinterpret_cast<f32x4*>(p.dp1 + row[s] * kPp + 16 * t + 4 * g) = dq[t][s];
                }
            }
        }
        // d_xpe = Wn1T d_p1, 17 output tiles as 9 + 8 (register budget)
        auto dxp_part = [&](auto tiles, int t0) {
            constexpr int TT = decltype(tiles)::value;
            f32x4 dx[TT][S];
            zero(dx, TT);
            mma_wr<TT, S, kTp>(dx, wlane(p.wn1t + (size_t)16 * t0 * kPp, kPp), kPp, dq);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (live[s]) *reinterpret_cast<f32x4*>(p.dxp + row[s] * kCe + 16 * (t0 + t) + 4 * g) = dx[t][s];
            }
        };
        dxp_part(std::integral_constant<int, 9>{}, 0);
        dxp_part(std::integral_constant<int, 8>{}, 9);
    }

    {   // ---- edge channel: d_h1 six (five) tiles at a time, each group folded into d_xedge right away -----------------------------
        f32x4 dx[kTe][S];
        zero(dx, kTe);
        auto dh1_group = [&](auto group_size, int t0) {
            constexpr int G = decltype(group_size)::value;
            f32x4 dh[G][S];
            zero(dh, G);
            mma_wx<G, S>(dh, wlane(p.wallt + (size_t)(kFn + 16 * t0) * kFn, kFn), kFn, kTn, g_l);
            f32x4 hv[G][S];     // h1 as one batch of loads
#pragma unroll
            for (int t = 0; t < G; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) hv[t][s] = ldv4(p.c1 + row[s] * kC1 + kFn + 16 * (t0 + t) + 4 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < G; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) dh[t][s][v] = hv[t][s][v] > 0.f ? dh[t][s][v] : 0.f;
                    if (live[s]) *reinterpret_cast<f32x4*>(p.dh1 + row[s] * kCe + 16 * (t0 + t) + 4 * g) = dh[t][s];
                }
            }
            // d_xedge[all 17 tiles] += W1T[:, this group's columns] d_h1 group, as 9 + 8 output tiles (A-operand registers)
            mma_wr<9, S, G>(dx, wlane(p.w1t + 16 * t0, kCe), kCe, dh);
            mma_wr<kTe - 9, S, G>(dx + 9, wlane(p.w1t + (size_t)(16 * 9) * kCe + 16 * t0, kCe), kCe, dh);
        };
        static_assert(kTe == 17, "d_h1 tile groups 6 + 6 + 5");
        dh1_group(std::integral_constant<int, 6>{}, 0);
        dh1_group(std::integral_constant<int, 6>{}, 6);
        dh1_group(std::integral_constant<int, 5>{}, 12);
#pragma unroll
        for (int t = 0; t < kTe; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (live[s]) *reinterpret_cast<f32x4*>(p.dxe + row[s] * kCe + 16 * t + 4 * g) = dx[t][s];
        }
    }
}

// ---- few rows: ONE slab per workgroup, the output tiles of every layer dealt out to its four waves --------------------------------------
// At the reference's own batch sizes (B = 200 .. 4096: 38 .. 768 slabs) the kernels above leave most SIMDs idle while one wave walks a slab
// through the whole layer chain (4 588 dependent-issue MFMAs, ~105 us whatever the row count).  Here wave w of a workgroup owns the output
// tiles w, w + 4, w + 8, ... of each layer; a finished layer goes to LDS in its accumulator layout -- which IS the B-operand layout of the
// next product (lstep_mma.h) -- and one barrier later every wave reads all of it.  Same sums per output element, a quarter of the chain per wave.
constexpr int kSplitMaxSlabs = 800;     // measured: 12 288 rows 107 / 88 us split vs 116 / 99 us whole (fwd / bwd); 16 384 rows 133 / 116 vs 112 / 101

// tile j of a wave: t = w + 4 j; waves whose last tile does not exist recompute their first one and drop it
template <int T>
struct WaveTiles {
    int t[T];
    bool valid[T];
    __device__ __forceinline__ WaveTiles(int w, int tiles) {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            valid[j] = w + 4 * j < tiles;
            t[j] = valid[j] ? w + 4 * j : w;
        }
    }
};

// acc[j] += sum over `chunks` 16-wide k chunks of (weight tile at wt[j])[i][k] * X[row i][k], X from global memory (xl = this lane's position)
// kMasked: X's rows hold only `kvalid` (a multiple of 4) meaningful columns from xl on; lanes that would read past them contribute zeros
// (as mma_wx_masked, lstep_mma.h).
template <int T, bool kMasked = false>
__device__ __forceinline__ void mma1_wx(f32x4 (&acc)[T], const float* const (&wt)[T], int chunks, const float* xl, int kvalid = 0, int g = 0) {
    struct Ch { f32x4 a[T]; f32x4 b; };
    auto load = [&](Ch& o, int c) {
        if constexpr (kMasked) o.b = 16 * c + 4 * g + 4 <= kvalid ? ldv4(xl + 16 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        else o.b = ldv4(xl + 16 * c);
#pragma unroll
        for (int j = 0; j < T; ++j) o.a[j] = ldv4(wt[j] + 16 * c);
    };
    auto run = [&](const Ch& o) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int j = 0; j < T; ++j) acc[j] = mfma4(o.a[j][v], o.b[v], acc[j]);
        }
    };
    Ch c0, c1;
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

// acc[j] += sum over the TK tiles of a layer held in LDS (tile tk of lane l at lds[tk * 64 + l]) of (weight tile at wt[j])[i][16 tk + k] * tile
template <int T, int TK>
__device__ __forceinline__ void mma1_wl(f32x4 (&acc)[T], const float* const (&wt)[T], const f32x4* lds, int lane) {
    f32x4 a[2][T], b[2];
#pragma unroll
    for (int j = 0; j < T; ++j) a[0][j] = ldv4(wt[j]);
    b[0] = lds[lane];
#pragma unroll
    for (int tk = 0; tk < TK; ++tk) {
        if (tk + 1 < TK) {
#pragma unroll
            for (int j = 0; j < T; ++j) a[(tk + 1) & 1][j] = ldv4(wt[j] + 16 * (tk + 1));
            b[(tk + 1) & 1] = lds[(tk + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int j = 0; j < T; ++j) acc[j] = mfma4(a[tk & 1][j][v], b[tk & 1][v], acc[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr int kSplitTp = (kTp + 3) / 4, kSplitTe = (kTe + 3) / 4, kSplitTn = (kTn + 3) / 4;   // tiles per wave: 3, 5, 3

__global__ __launch_bounds__(kBlock) void tail_fwd_split_kernel(const TailFwdParams p) {
    __shared__ f32x4 s_p1[kTp * 64], s_q[kTp * 64], s_h1[kTe * 64];
    const int lane = lane_id(), w = wave_in_block();
    const int i = lane & 15, g = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    const bool live = r0 + i < p.m;
    const int64_t row = live ? r0 + i : p.m - 1;
    const float* xe_l = p.xe + row * p.ld_e + 4 * g;
    const float* xp_l = p.xp + row * p.ld_p + 4 * g;
    const float* c1_l = p.c1 + row * kC1 + 4 * g;
    const float* c2_l = p.c2 + row * kC2 + 4 * g;
    const WaveTiles<kSplitTp> tp(w, kTp);
    const WaveTiles<kSplitTe> te(w, kTe);
    const WaveTiles<kSplitTn> tn(w, kTn);

    {   // ---- p1 = relu(Wn1 x_pe + bn1)
        f32x4 acc[kSplitTp];
        const float* wt[kSplitTp];
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
            acc[j] = ldv4(p.bn1 + 16 * tp.t[j] + 4 * g);
            wt[j] = p.wn1 + (size_t)(16 * tp.t[j] + i) * kCe + 4 * g;
        }
        mma1_wx<kSplitTp>(acc, wt, kTe, xp_l);
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[j][v] = fmaxf(acc[j][v], 0.f);
            if (tp.valid[j]) {
                s_p1[tp.t[j] * 64 + lane] = acc[j];
                if (live) *reinterpret_cast<f32x4*>(p.c2 + row * kC2 + kPp + 16 * tp.t[j] + 4 * g) = acc[j];
            }
        }
    }
    {   // ---- h1 = relu(W1 x_edge + b1)
        f32x4 acc[kSplitTe];
        const float* wt[kSplitTe];
#pragma unroll
        for (int j = 0; j < kSplitTe; ++j) {
            acc[j] = ldv4(p.b1 + 16 * te.t[j] + 4 * g);
            wt[j] = p.w1 + (size_t)(16 * te.t[j] + i) * kCe + 4 * g;
        }
        mma1_wx<kSplitTe>(acc, wt, kTe, xe_l);
#pragma unroll
        for (int j = 0; j < kSplitTe; ++j) {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[j][v] = fmaxf(acc[j][v], 0.f);
            if (te.valid[j]) {
                s_h1[te.t[j] * 64 + lane] = acc[j];
                if (live) *reinterpret_cast<f32x4*>(p.c1 + row * kC1 + kFn + 16 * te.t[j] + 4 * g) = acc[j];
            }
        }
    }
    __syncthreads();
    {   // ---- q = own + tanh(Wq [own ; p1] + bq)
        f32x4 acc[kSplitTp];
        const float *wt[kSplitTp], *wt2[kSplitTp];
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
            acc[j] = ldv4(p.bq + 16 * tp.t[j] + 4 * g);
            wt[j] = p.wq + (size_t)(16 * tp.t[j] + i) * kC2 + 4 * g;
            wt2[j] = wt[j] + kPp;
        }
        mma1_wx<kSplitTp>(acc, wt, kTp, c2_l);
        mma1_wl<kSplitTp, kTp>(acc, wt2, s_p1, lane);
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
            const f32x4 own = ldv4(p.c2 + row * kC2 + 16 * tp.t[j] + 4 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[j][v] = own[v] + tanh_fast(acc[j][v]);
            if (tp.valid[j]) {
                s_q[tp.t[j] * 64 + lane] = acc[j];
                if (live) *reinterpret_cast<f32x4*>(p.c1 + row * kC1 + kFn + kCe + 16 * tp.t[j] + 4 * g) = acc[j];
            }
        }
    }
    __syncthreads();
    {   // ---- out = Wall [x_node ; h1 ; q] + ball
        f32x4 acc[kSplitTn];
        const float *wt[kSplitTn], *wt_h[kSplitTn], *wt_q[kSplitTn];
#pragma unroll
        for (int j = 0; j < kSplitTn; ++j) {
            acc[j] = ldv4(p.ball + 16 * tn.t[j] + 4 * g);
            wt[j] = p.wall + (size_t)(16 * tn.t[j] + i) * kC1 + 4 * g;
            wt_h[j] = wt[j] + kFn;
            wt_q[j] = wt[j] + kFn + kCe;
        }
        mma1_wx<kSplitTn>(acc, wt, kTn, c1_l);
        mma1_wl<kSplitTn, kTp>(acc, wt_q, s_q, lane);
        mma1_wl<kSplitTn, kTe>(acc, wt_h, s_h1, lane);
#pragma unroll
        for (int j = 0; j < kSplitTn; ++j)
            if (tn.valid[j] && live) *reinterpret_cast<f32x4*>(p.out + row * kFn + 16 * tn.t[j] + 4 * g) = acc[j];
    }
}

__global__ __launch_bounds__(kBlock) void tail_bwd_split_kernel(const TailBwdParams p) {
    __shared__ f32x4 s_dz[kTp * 64], s_dp1[kTp * 64], s_dh1[kTe * 64];
    const int lane = lane_id(), w = wave_in_block();
    const int i = lane & 15, g = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    const bool live = r0 + i < p.m;
    const int64_t row = live ? r0 + i : p.m - 1;
    const float* g_l = p.g + row * kFn + 4 * g;
    const WaveTiles<kSplitTp> tp(w, kTp);
    const WaveTiles<kSplitTe> te(w, kTe);
    auto zero = [](auto& a) {
        for (auto& x : a) x = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    f32x4 dq[kSplitTp];
    {   // ---- d_q = WallT[q rows] g;  d_z = d_q (1 - tanh^2)
        const float* wt[kSplitTp];
        zero(dq);
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) wt[j] = p.wallt + (size_t)(kFn + kCe + 16 * tp.t[j] + i) * kFn + 4 * g;
        mma1_wx<kSplitTp>(dq, wt, kTn, g_l);
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
            const f32x4 qv = ldv4(p.c1 + row * kC1 + kFn + kCe + 16 * tp.t[j] + 4 * g);
            const f32x4 ov = ldv4(p.c2 + row * kC2 + 16 * tp.t[j] + 4 * g);
            f32x4 dz;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float th = qv[v] - ov[v];
                dz[v] = dq[j][v] * (1.f - th * th);
            }
            if (tp.valid[j]) {
                s_dz[tp.t[j] * 64 + lane] = dz;
                if (live) *reinterpret_cast<f32x4*>(p.dz + row * kPp + 16 * tp.t[j] + 4 * g) = dz;
            }
        }
    }
    {   // ---- d_h1 = WallT[h1 rows] g * [h1 > 0]
        f32x4 dh[kSplitTe];
        const float* wt[kSplitTe];
        zero(dh);
#pragma unroll
        for (int j = 0; j < kSplitTe; ++j) wt[j] = p.wallt + (size_t)(kFn + 16 * te.t[j] + i) * kFn + 4 * g;
        mma1_wx<kSplitTe>(dh, wt, kTn, g_l);
#pragma unroll
        for (int j = 0; j < kSplitTe; ++j) {
            const f32x4 hv = ldv4(p.c1 + row * kC1 + kFn + 16 * te.t[j] + 4 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) dh[j][v] = hv[v] > 0.f ? dh[j][v] : 0.f;
            if (te.valid[j]) {
                s_dh1[te.t[j] * 64 + lane] = dh[j];
                if (live) *reinterpret_cast<f32x4*>(p.dh1 + row * kCe + 16 * te.t[j] + 4 * g) = dh[j];
            }
        }
    }
    __syncthreads();
    {   // ---- d_own = d_q + WqT[own rows] d_z;  d_p1 = WqT[p1 rows] d_z * [p1 > 0]
        const float *wt[kSplitTp], *wt2[kSplitTp];
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
            wt[j] = p.wqt + (size_t)(16 * tp.t[j] + i) * kPp + 4 * g;
            wt2[j] = wt[j] + (size_t)kPp * kPp;
        }
        mma1_wl<kSplitTp, kTp>(dq, wt, s_dz, lane);
        f32x4 dp[kSplitTp];
        zero(dp);
        mma1_wl<kSplitTp, kTp>(dp, wt2, s_dz, lane);
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
            const f32x4 pv = ldv4(p.c2 + row * kC2 + kPp + 16 * tp.t[j] + 4 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) dp[j][v] = pv[v] > 0.f ? dp[j][v] : 0.f;
            if (tp.valid[j]) {
                s_dp1[tp.t[j] * 64 + lane] = dp[j];
                if (live) {
                    *reinterpret_cast<f32x4*>(p.down + row * p.ld_down + 16 * tp.t[j] + 4 * g) = dq[j];
                    *reinterpret_cast<f32x4*>(p.dp1 + row * kPp + 16 * tp.t[j] + 4 * g) = dp[j];
                }
            }
        }
    }
    {   // ---- d_xedge = W1T d_h1
        f32x4 dx[kSplitTe];
        const float* wt[kSplitTe];
        zero(dx);
#pragma unroll
        for (int j = 0; j < kSplitTe; ++j) wt[j] = p.w1t + (size_t)(16 * te.t[j] + i) * kCe + 4 * g;
        mma1_wl<kSplitTe, kTe>(dx, wt, s_dh1, lane);
#pragma unroll
        for (int j = 0; j < kSplitTe; ++j)
            if (te.valid[j] && live) *reinterpret_cast<f32x4*>(p.dxe + row * kCe + 16 * te.t[j] + 4 * g) = dx[j];
    }
    __syncthreads();
    {   // ---- d_xpe = Wn1T d_p1
        f32x4 dx[kSplitTe];
        const float* wt[kSplitTe];
        zero(dx);
#pragma unroll
        for (int j = 0; j < kSplitTe; ++j) wt[j] = p.wn1t + (size_t)(16 * te.t[j] + i) * kPp + 4 * g;
        mma1_wl<kSplitTe, kTp>(dx, wt, s_dp1, lane);
#pragma unroll
        for (int j = 0; j < kSplitTe; ++j)
            if (te.valid[j] && live) *reinterpret_cast<f32x4*>(p.dxp + row * kCe + 16 * te.t[j] + 4 * g) = dx[j];
    }
}

static bool tail_split(int64_t m) {
    const char* off = getenv("LSTEP_TAIL_NO_SPLIT");   // A/B and the split-vs-whole parity test: read per call
    return !(off && off[0] == '1') && (m + 15) / 16 <= kSplitMaxSlabs;
}

// ---- update_pe: z = pe_mlp_2(relu(pe_mlp_1(agg))) [+ self_update_pe(own)], table[id] += tanh(z)  (models/LSTEP.py:292-303, 327-339) ----
struct UpdateParams {
    const float* agg;      // [>= n, ld_agg]  aggregated messages cat[pe, time] (kCe used)
    const int64_t* ids;    // [n] rows of the table to update (unique)
    const float *w1, *b1, *w2, *b2;   // [176, 272], [176], [176, 176], [176]   (zero-padded)
    const float *ws, *bs;  // [176, 176], [176] self_update_pe, or NULL (phase 2: the term is dead code in the reference)
    float* table;          // [N + 1, pe_dim]
    float* mirror;         // optional second table of the same shape that receives the new rows too (the batch's history slot)
    const int32_t* live;   // optional device count: only the first min(*live, n) rows are updated (n is then the capacity the grid covers)
    const int32_t* ring_start;   // optional lstep_ring_ref_t: `mirror` is the base of the history ring, the slot index lives on the device
    int32_t ring_add, ring_slots;
    int64_t ring_stride;
    int64_t n;
    int32_t ld_agg, pe_dim;
    int32_t time_dim;      // kPre kernels: meaningful columns of the time part
    int32_t mirror_world, mirror_rank;   // world > 1: `mirror` is an owner-sharded slot -- row id goes to row id / world, only if id % world == rank
};

constexpr int kTd = 112, kTt = kTd / 16;     // padded time width of the pre-multiplied form

// kPre: the first layer arrives half done.  pe_mlp_1 is linear in the message sum, W1 [sum pe rows ; sum time features] = sum (W1a pe row) +
// W1b sum time features, and every message's pe row is one of the U batch-node rows: the caller multiplies those U rows by W1a once
// (a [U, 172] x [172, 172] product instead of 172 x 172 multiply-adds for each of the ~9 U touched rows), the segment sums run over the
// products, and agg = [sum W1a pe (176) | sum time features (time_dim)].  Here: h = relu(agg[:176] + W1b agg[176:] + b1), w1 = W1b [176, 112].
// One task = S slabs of 16 rows, start to finish (see update_rows_kernel).  `w2` / `ldw2`: the second layer's weights -- global memory
// [176, 176], or the workgroup's LDS copy with a padded row stride (update_rows_lds_kernel).  `p.n` is already clamped to the live count and
// `p.mirror` already points at the slot.
template <int S, bool kPre, bool kSelf>
__device__ __forceinline__ void update_rows_task(const UpdateParams& p, int64_t task, const float* w2, int ldw2) {
    static_assert(!(kPre && kSelf), "the pre-multiplied form is phase 2: no self term");
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const int64_t r0 = task * (16 * S);
    bool live[S];
    const float* agg_l[S];
    int64_t id[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int64_t r = r0 + 16 * s + i;
        live[s] = r < p.n;
        if (!live[s]) r = p.n - 1;
        agg_l[s] = p.agg + r * p.ld_agg + 4 * g;
        id[s] = LSTEP_CHECKED(p.ids[r], LSTEP_NODE_ROWS(), kCheckUpdateRowsId);      // (checked builds only: lstep_common.h)
    }
    auto wlane = [&](const float* w, int ldw) { return w + i * ldw + 4 * g; };
    // A wave has its SIMD to itself (all 512 registers): whatever it waits for, nothing else on the SIMD covers.  So the row loads are batched
    // -- straight-line code, every load of a batch issued before the first one is used: the partial sums of the pre-multiplied form up
    // front, the old table rows (into the registers of the dead hidden layer) after the second layer.  Tile-by-tile loops cost one full
    // memory latency PER TILE: the compiler may not hoist a load over the previous tile's store (they might alias), and a uniform branch
    // inside a tile loop splits it into blocks that each end in vmcnt(0).  Measured at 290 k rows: 460 -> 367 us.
    // (Issuing the batches behind the operand loads of the MFMA phase before them does not help: loads return in order, so the next
    // weight tiles queue behind the batch, and the batch is a bandwidth-bound burst -- every wave of the launch issues it at the same time.)
    f32x4 h[kTp][S], z[kTp][S];
    float *own_row[S], *mir_row[S];     // mir_row: where this row goes in the mirror (NULL: nowhere)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        own_row[s] = p.table + id[s] * p.pe_dim;
        mir_row[s] = nullptr;
        if (p.mirror) {
            if (p.mirror_world > 1) {
                if (id[s] % p.mirror_world == p.mirror_rank) mir_row[s] = p.mirror + (id[s] / p.mirror_world) * p.pe_dim;
            } else {
                mir_row[s] = p.mirror + id[s] * p.pe_dim;
            }
        }
    }
    if constexpr (kPre) {
#pragma unroll
        for (int t = 0; t < kTp; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s) h[t][s] = ldv4(agg_l[s] + 16 * t);      // accumulator layout = row-major float4 at feature 16 t + 4 g
        }
    }
#pragma unroll
    for (int t = 0; t < kTp; ++t) {
        const f32x4 b1v = ldv4(p.b1 + 16 * t + 4 * g);
        f32x4 b2v = ldv4(p.b2 + 16 * t + 4 * g);
        if constexpr (kSelf) b2v += ldv4(p.bs + 16 * t + 4 * g);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if constexpr (kPre) h[t][s] += b1v;
            else h[t][s] = b1v;
            z[t][s] = b2v;
        }
    }
    if constexpr (kPre) {
        const float* tf_l[S];
#pragma unroll
        for (int s = 0; s < S; ++s) tf_l[s] = agg_l[s] + kPp;
        mma_wx_masked<kTp, S>(h, wlane(p.w1, kTd), kTd, kTt, tf_l, p.time_dim, g);
    } else {
        mma_wx<kTp, S>(h, wlane(p.w1, kCe), kCe, kTe, agg_l);
    }
#pragma unroll
    for (int t = 0; t < kTp; ++t) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int v = 0; v < 4; ++v) h[t][s][v] = fmaxf(h[t][s][v], 0.f);
        }
    }
    mma_wr<kTp, S, kTp>(z, wlane(w2, ldw2), ldw2, h);
    if constexpr (kSelf) {
        // own rows straight from the table: pe_dim = 172 columns, so the last lane group of the last chunk would read past the row: masked
        // (the padded weight columns it would meet are zero)
        const float* own_l[S];
#pragma unroll
        for (int s = 0; s < S; ++s) own_l[s] = own_row[s] + 4 * g;
        mma_wx_masked<kTp, S>(z, wlane(p.ws, kPp), kPp, (p.pe_dim + 15) / 16, own_l, p.pe_dim, g);
    }
    f32x4 rows[kTp][S];
#pragma unroll
    for (int t = 0; t < kTp; ++t) {
        const int f = 16 * t + 4 * g;
#pragma unroll
        for (int s = 0; s < S; ++s) rows[t][s] = ldv4(f + 4 <= p.pe_dim ? own_row[s] + f : own_row[s]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kTp; ++t) {
        const int f = 16 * t + 4 * g;
        if (f + 4 <= p.pe_dim) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (live[s]) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) rows[t][s][v] += tanh_fast(z[t][s][v]);
                    *reinterpret_cast<f32x4*>(own_row[s] + f) = rows[t][s];
                    if (mir_row[s]) *reinterpret_cast<f32x4*>(mir_row[s] + f) = rows[t][s];
                }
            }
        }
    }
}

template <int S, bool kPre = false, bool kSelf = false>
__global__ __launch_bounds__(kBlock, 1) void update_rows_kernel(UpdateParams p) {
    const int64_t task = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (p.live) {
        const int64_t live = *p.live;
        if (live < p.n) p.n = live;
    }
    if (task * (16 * S) >= p.n) return;   // no barriers in this kernel
    if (p.ring_start && p.mirror) p.mirror += (int64_t)((*p.ring_start + p.ring_add) % p.ring_slots) * p.ring_stride;
    update_rows_task<S, kPre, kSelf>(p, task, p.w2, kPp);
}

// The pre-multiplied form for MANY rows (update_pe phase 2: 290 k rows per c4 step) as a persistent kernel with the second layer's weights
// resident in LDS (round 4).  update_rows_kernel<3, true> spends 187 us in MFMA phases and 185 us moving rows, one
// after the other: its waves own all 512 registers of their SIMD, so nothing runs beside them, and every wave of the launch is in the same
// phase at the same time.  Here: ONE slab per wave (<= 256 registers), EIGHT waves per workgroup = two per SIMD, one
// workgroup per CU; W2 [176, 176] (124 KB of the 203 KB of weights) is copied into LDS once per workgroup (row stride 196 floats = 4 mod
// 64 banks: the 16 lanes of a quarter-wave read 16 consecutive rows at the same column without a bank conflict) and read from there by
// every task; only W1b [176, 112] (79 KB) still streams from L2 per slab -- 2.6x less L2 -> CU traffic than one slab per wave used to cost
// (that form saturated the L2 -> CU path: 510 vs 367 us, appendix A).  Waves take slabs round-robin and never meet at a barrier after the
// copy, so they drift apart and one wave's row loads / stores run under its SIMD partner's MFMAs.
// The task body of that kernel: update_rows_task<1, true, false> with both layers' output tiles processed in two halves (6 + 5 tiles), so
// that accumulators, operand buffers and the old rows of ONE half are alive at a time: <= 256 registers without spills (the whole-width body
// needs 400).  Same products in the same order per output element: bit-identical to update_rows_kernel<1, true>.
template <int T>
__device__ __forceinline__ void update_pre_layer1_half(const UpdateParams& p, const float* agg_l, int t0, int i, int g, f32x4 (*h)[1]) {
    f32x4 acc[T][1];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t][0] = ldv4(agg_l + 16 * (t0 + t)) + ldv4(p.b1 + 16 * (t0 + t) + 4 * g);
    const float* tf_l[1] = {agg_l + kPp};
    mma_wx_masked<T, 1>(acc, p.w1 + (size_t)(16 * t0 + i) * kTd + 4 * g, kTd, kTt, tf_l, p.time_dim, g);
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int v = 0; v < 4; ++v) h[t0 + t][0][v] = fmaxf(acc[t][0][v], 0.f);
    }
}

template <int T>
__device__ __forceinline__ void update_pre_layer2_half(const UpdateParams& p, const float* w2, int ldw2, int t0, int i, int g, const f32x4 (*h)[1],
                                                       float* own_row, float* mir_row, bool live) {
    f32x4 z[T][1];
#pragma unroll
    for (int t = 0; t < T; ++t) z[t][0] = ldv4(p.b2 + 16 * (t0 + t) + 4 * g);
    mma_wr<T, 1, kTp>(z, w2 + (size_t)(16 * t0 + i) * ldw2 + 4 * g, ldw2, h);
    f32x4 rows[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int f = 16 * (t0 + t) + 4 * g;
        rows[t] = ldv4(f + 4 <= p.pe_dim ? own_row + f : own_row);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int f = 16 * (t0 + t) + 4 * g;
        if (f + 4 <= p.pe_dim && live) {
#pragma unroll
            for (int v = 0; v < 4; ++v) rows[t][v] += tanh_fast(z[t][0][v]);
            *reinterpret_cast<f32x4*>(own_row + f) = rows[t];
            if (mir_row) *reinterpret_cast<f32x4*>(mir_row + f) = rows[t];
        }
    }
}

__device__ __forceinline__ void update_rows_pre_task_halves(const UpdateParams& p, int64_t task, const float* w2, int ldw2) {
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    int64_t r = task * 16 + i;
    const bool live = r < p.n;
    if (!live) r = p.n - 1;
    const float* agg_l = p.agg + r * p.ld_agg + 4 * g;
    const int64_t id = LSTEP_CHECKED(p.ids[r], LSTEP_NODE_ROWS(), kCheckUpdateRowsId);
    float* own_row = p.table + id * p.pe_dim;
    float* mir_row = nullptr;
    if (p.mirror) {
        if (p.mirror_world > 1) {
            if (id % p.mirror_world == p.mirror_rank) mir_row = p.mirror + (id / p.mirror_world) * p.pe_dim;
        } else {
            mir_row = p.mirror + id * p.pe_dim;
        }
    }
    constexpr int kHalf = 6;
    f32x4 h[kTp][1];
    // (the weight bases are made opaque per task: otherwise the ~100 tile addresses are hoisted out of the persistent kernel's task loop as
    // loop invariants and the body spills)
    UpdateParams q = p;
    int64_t zero = 0;
    asm volatile("" : "+s"(zero));       // (an opaque offset, not an opaque pointer: the loads stay global_load, not flat_load)
    q.w1 = p.w1 + zero;
    q.b1 = p.b1 + zero;
    q.b2 = p.b2 + zero;
    update_pre_layer1_half<kHalf>(q, agg_l, 0, i, g, h);
    update_pre_layer1_half<kTp - kHalf>(q, agg_l, kHalf, i, g, h);
    update_pre_layer2_half<kHalf>(q, w2, ldw2, 0, i, g, h, own_row, mir_row, live);
    update_pre_layer2_half<kTp - kHalf>(q, w2, ldw2, kHalf, i, g, h, own_row, mir_row, live);
}

constexpr int kUpdLdsLd = 196;                       // LDS row stride of W2 (floats)
constexpr int kUpdLdsBytes = kPp * kUpdLdsLd * 4;    // 137 984 B
template <int kUpdLdsWaves>
__global__ __launch_bounds__(kUpdLdsWaves * kWave, 1) void update_rows_lds_kernel(UpdateParams p) {
    extern __shared__ __attribute__((aligned(16))) float s_w2[];
    for (int e = threadIdx.x; e < kPp * (kPp / 4); e += kUpdLdsWaves * kWave) {
        const int r = e / (kPp / 4), c4 = e % (kPp / 4);
        *reinterpret_cast<f32x4*>(&s_w2[r * kUpdLdsLd + 4 * c4]) = ldv4(p.w2 + (size_t)r * kPp + 4 * c4);
    }
    __syncthreads();
    if (p.live) {
        const int64_t live = *p.live;
        if (live < p.n) p.n = live;
    }
    if (p.ring_start && p.mirror) p.mirror += (int64_t)((*p.ring_start + p.ring_add) % p.ring_slots) * p.ring_stride;
    const int64_t slabs = (p.n + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * kUpdLdsWaves;
    for (int64_t task = (int64_t)blockIdx.x * kUpdLdsWaves + wave_in_block(); task < slabs; task += stride)
        update_rows_pre_task_halves(p, task, s_w2, kUpdLdsLd);
}

// update_pe for few rows: one slab per workgroup, the 11 output tiles of both layers dealt out to its four waves (as tail_fwd_split_kernel
// above).  At B = 200 .. 4096 the slab-chain kernel's ~45 us are one wave's walk through 2 x 11 tiles whatever the row count, and
// update_pe runs it twice per step on the chain of dependent kernels that bounds the small-batch step.  Same products in the same order per
// output element as update_rows_kernel<1, kPre>.
template <bool kPre>
__global__ __launch_bounds__(kBlock) void update_rows_split_kernel(UpdateParams p) {
    __shared__ f32x4 s_h[kTp * 64];
    const int lane = lane_id(), w = wave_in_block();
    const int i = lane & 15, g = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    if (p.live) {
        const int64_t live = *p.live;
        if (live < p.n) p.n = live;
    }
    if (r0 >= p.n) return;   // the whole workgroup: no wave is left waiting at a barrier
    if (p.ring_start && p.mirror) p.mirror += (int64_t)((*p.ring_start + p.ring_add) % p.ring_slots) * p.ring_stride;
    const bool live = r0 + i < p.n;
    const int64_t r = live ? r0 + i : p.n - 1;
    const float* agg_l = p.agg + r * p.ld_agg + 4 * g;
    float* own_row = p.table + LSTEP_CHECKED(p.ids[r], LSTEP_NODE_ROWS(), kCheckUpdateRowsId) * p.pe_dim;
    float* mir_row = nullptr;
    if (p.mirror) {
        const int64_t id = LSTEP_CHECKED(p.ids[r], LSTEP_NODE_ROWS(), kCheckUpdateRowsId);
        if (p.mirror_world > 1) {
            if (id % p.mirror_world == p.mirror_rank) mir_row = p.mirror + (id / p.mirror_world) * p.pe_dim;
        } else {
            mir_row = p.mirror + id * p.pe_dim;
        }
    }
    const WaveTiles<kSplitTp> tp(w, kTp);
    {   // ---- h = relu(pe_mlp_1(agg))
        f32x4 acc[kSplitTp];
        const float* wt[kSplitTp];
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
            acc[j] = ldv4(p.b1 + 16 * tp.t[j] + 4 * g);
            if constexpr (kPre) {
                acc[j] += ldv4(agg_l + 16 * tp.t[j]);
                wt[j] = p.w1 + (size_t)(16 * tp.t[j] + i) * kTd + 4 * g;
            } else {
                wt[j] = p.w1 + (size_t)(16 * tp.t[j] + i) * kCe + 4 * g;
            }
        }
        if constexpr (kPre) mma1_wx<kSplitTp, true>(acc, wt, kTt, agg_l + kPp, p.time_dim, g);
        else mma1_wx<kSplitTp>(acc, wt, kTe, agg_l);
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[j][v] = fmaxf(acc[j][v], 0.f);
            if (tp.valid[j]) s_h[tp.t[j] * 64 + lane] = acc[j];
        }
    }
    __syncthreads();
    f32x4 z[kSplitTp];
    {   // ---- z = pe_mlp_2(h) [+ self_update_pe(own)]
        const float* wt[kSplitTp];
#pragma unroll
        for (int j = 0; j < kSplitTp; ++j) {
            z[j] = ldv4(p.b2 + 16 * tp.t[j] + 4 * g);
            if (p.ws != nullptr) z[j] += ldv4(p.bs + 16 * tp.t[j] + 4 * g);
            wt[j] = p.w2 + (size_t)(16 * tp.t[j] + i) * kPp + 4 * g;
        }
        mma1_wl<kSplitTp, kTp>(z, wt, s_h, lane);
        if (!kPre && p.ws != nullptr) {
            // own rows straight from the table, pe_dim = 172 columns: the lanes that would read past the row re-read its start instead
            // (the padded weight columns they meet are zero)
            const int last = (p.pe_dim + 15) / 16 - 1;
            for (int c = 0; c <= last; ++c) {
                const bool ok = 16 * c + 4 * g + 4 <= p.pe_dim;
                const f32x4 b = ldv4(ok ? own_row + 4 * g + 16 * c : own_row);
                f32x4 a[kSplitTp];
#pragma unroll
                for (int j = 0; j < kSplitTp; ++j) a[j] = ldv4(p.ws + (size_t)(16 * tp.t[j] + i) * kPp + 4 * g + 16 * c);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
#pragma unroll
                    for (int j = 0; j < kSplitTp; ++j) z[j] = mfma4(a[j][v], b[v], z[j]);
                }
            }
            __syncthreads();   // every wave has read the whole old rows before any wave writes its columns (ws != NULL is uniform)
        }
    }
#pragma unroll
    for (int j = 0; j < kSplitTp; ++j) {
        const int f = 16 * tp.t[j] + 4 * g;
        if (tp.valid[j] && live && f + 4 <= p.pe_dim) {
            f32x4 old = ldv4(own_row + f);
#pragma unroll
            for (int v = 0; v < 4; ++v) old[v] += tanh_fast(z[j][v]);
            *reinterpret_cast<f32x4*>(own_row + f) = old;
            if (mir_row) *reinterpret_cast<f32x4*>(mir_row + f) = old;
        }
    }
}

constexpr int kUpdateSplitMaxSlabs = 700;     // measured: 11 000 rows 40 / 21 us split vs 43 / 24 us whole (plain / pre-multiplied); 12 800 rows 53 / 26 vs 44 / 24

static bool update_split(int64_t n) {
    const char* off = getenv("LSTEP_UPDATE_NO_SPLIT");   // A/B and the split-vs-whole parity test: read per call
    return !(off && off[0] == '1') && (n + 15) / 16 <= kUpdateSplitMaxSlabs;
}

}  // namespace lstep

using namespace lstep;

// number of 16-row slabs per wave: fewest rounds of 1024 waves (256 CUs x 4 SIMDs, one wave each), weighted by what a slab costs at
// that width (a wave re-reads all the weights whatever S is: measured 1.35 / 1.1 / 1.0 relative time per slab at S = 1 / 2 / 3)
static int tail_slabs_per_wave(int64_t m) {
    const int64_t slabs = (m + 15) / 16;
    const double per_slab[4] = {0.0, 1.35, 1.1, 1.0};
    int best = 3;
    double best_cost = -1.0;
    for (int s = 3; s >= 1; --s) {
        const int64_t tasks = (slabs + s - 1) / s;
        const double cost = (double)((tasks + 1023) / 1024) * s * per_slab[s];
        if (best_cost < 0 || cost < best_cost) { best = s; best_cost = cost; }
    }
    return best;
}

// update_pe's kernels: the S = 1 form is compiled for two waves per SIMD (<= 256 registers), so one wave's row loads and stores overlap the
// other's MFMAs.  LSTEP_UPDATE_S=1|2|3 forces a form (tuning).
static int update_slabs_per_wave(int64_t n) {
    const char* force = getenv("LSTEP_UPDATE_S");
    if (force && force[0] >= '1' && force[0] <= '3') return force[0] - '0';
    return tail_slabs_per_wave(n);
}

extern "C" int lstep_tail_fwd(const float* x_edge, int32_t ld_edge, const float* x_pe, int32_t ld_pe, float* cat1, float* cat2, float* out,
                              const float* w1, const float* b1, const float* wn1, const float* bn1, const float* wq, const float* bq,
                              const float* wall, const float* ball, int64_t m, void* stream) {
    if (m < 0 || ld_edge < kCe || ld_pe < kCe || (ld_edge & 3) || (ld_pe & 3)) return set_error(LSTEP_EINVAL, "lstep_tail_fwd: bad sizes");
    if (m == 0) return LSTEP_OK;
    if (!x_edge || !x_pe || !cat1 || !cat2 || !out || !w1 || !b1 || !wn1 || !bn1 || !wq || !bq || !wall || !ball)
        return set_error(LSTEP_EINVAL, "lstep_tail_fwd: NULL pointer");
    TailFwdParams p;
    p.xe = x_edge; p.xp = x_pe; p.c1 = cat1; p.c2 = cat2; p.out = out;
    p.w1 = w1; p.b1 = b1; p.wn1 = wn1; p.bn1 = bn1; p.wq = wq; p.bq = bq; p.wall = wall; p.ball = ball;
    p.m = m; p.ld_e = ld_edge; p.ld_p = ld_pe;
    hipStream_t s = (hipStream_t)stream;
    if (tail_split(m)) {
        hipLaunchKernelGGL(tail_fwd_split_kernel, dim3((unsigned)((m + 15) / 16)), dim3(kBlock), 0, s, p);
        return check_launch("lstep_tail_fwd<split>");
    }
    const int S = tail_slabs_per_wave(m);
    const int64_t tasks = ((m + 15) / 16 + S - 1) / S;
    const dim3 grid((unsigned)((tasks + kWavesPerBlock - 1) / kWavesPerBlock)), block(kBlock);
    if (S == 1) hipLaunchKernelGGL(tail_fwd_kernel<1>, grid, block, 0, s, p);
    else if (S == 2) hipLaunchKernelGGL(tail_fwd_kernel<2>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(tail_fwd_kernel<3>, grid, block, 0, s, p);
    return check_launch("lstep_tail_fwd");
}

extern "C" int lstep_tail_bwd(const float* grad_out, const float* cat1, const float* cat2, const float* w1t, const float* wn1t,
                              const float* wqt, const float* wallt, float* d_xedge, float* d_xpe, float* d_own, int32_t ld_down, float* d_h1,
                              float* d_p1, float* d_z, int64_t m, void* stream) {
    if (m < 0 || ld_down < kPp || (ld_down & 3)) return set_error(LSTEP_EINVAL, "lstep_tail_bwd: bad sizes");
    if (m == 0) return LSTEP_OK;
    if (!grad_out || !cat1 || !cat2 || !w1t || !wn1t || !wqt || !wallt || !d_xedge || !d_xpe || !d_own || !d_h1 || !d_p1 || !d_z)
        return set_error(LSTEP_EINVAL, "lstep_tail_bwd: NULL pointer");
    TailBwdParams p;
    p.g = grad_out; p.c1 = cat1; p.c2 = cat2; p.w1t = w1t; p.wn1t = wn1t; p.wqt = wqt; p.wallt = wallt;
    p.dxe = d_xedge; p.dxp = d_xpe; p.down = d_own; p.dh1 = d_h1; p.dp1 = d_p1; p.dz = d_z; p.m = m; p.ld_down = ld_down;
    hipStream_t s = (hipStream_t)stream;
    if (tail_split(m)) {
        hipLaunchKernelGGL(tail_bwd_split_kernel, dim3((unsigned)((m + 15) / 16)), dim3(kBlock), 0, s, p);
        return check_launch("lstep_tail_bwd<split>");
    }
    const int S = tail_slabs_per_wave(m);
    const int64_t tasks = ((m + 15) / 16 + S - 1) / S;
    const dim3 grid((unsigned)((tasks + kWavesPerBlock - 1) / kWavesPerBlock)), block(kBlock);
    if (S == 1) hipLaunchKernelGGL(tail_bwd_kernel<1>, grid, block, 0, s, p);
    else if (S == 2) hipLaunchKernelGGL(tail_bwd_kernel<2>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(tail_bwd_kernel<3>, grid, block, 0, s, p);
    return check_launch("lstep_tail_bwd");
}

extern "C" int lstep_update_rows(const float* agg, int32_t ld_agg, const int64_t* ids, int64_t n, const float* w1, const float* b1, const float* w2,
                                 const float* b2, const float* ws, const float* bs, float* table, float* mirror, int32_t pe_dim,
                                 const int32_t* num_live, const lstep_ring_ref_t* ring, int32_t mirror_world, int32_t mirror_rank, void* stream) {
    if (mirror_world < 1 || mirror_rank < 0 || mirror_rank >= mirror_world) return set_error(LSTEP_EINVAL, "lstep_update_rows: bad mirror shard");
    if (n < 0 || ld_agg < kCe || (ld_agg & 3) || pe_dim <= 0 || pe_dim > kPp || (pe_dim & 3)) return set_error(LSTEP_EINVAL, "lstep_update_rows: bad sizes");
    if (n == 0) return LSTEP_OK;
    if (!agg || !ids || !w1 || !b1 || !w2 || !b2 || !table || (ws && !bs)) return set_error(LSTEP_EINVAL, "lstep_update_rows: NULL pointer");
    if (((uintptr_t)mirror) & 15) return set_error(LSTEP_EINVAL, "lstep_update_rows: misaligned mirror table");
    if (ring && (!ring->start || ring->slots <= 0 || ring->add < 0 || (ring->slot_stride & 3)))
        return set_error(LSTEP_EINVAL, "lstep_update_rows: bad ring reference");
    UpdateParams p{agg, ids, w1, b1, w2, b2, ws, bs, table, mirror, num_live, ring ? ring->start : nullptr, ring ? ring->add : 0,
                   ring ? ring->slots : 1, ring ? ring->slot_stride : 0, n, ld_agg, pe_dim, 0, mirror_world, mirror_rank};
    hipStream_t s = (hipStream_t)stream;
    if (update_split(n)) {
        hipLaunchKernelGGL(update_rows_split_kernel<false>, dim3((unsigned)((n + 15) / 16)), dim3(kBlock), 0, s, p);
        return check_launch("lstep_update_rows<split>");
    }
    const int S = update_slabs_per_wave(n);
    const int64_t tasks = ((n + 15) / 16 + S - 1) / S;
    const dim3 grid((unsigned)((tasks + kWavesPerBlock - 1) / kWavesPerBlock)), block(kBlock);
    if (ws) {
        if (S == 1) hipLaunchKernelGGL((update_rows_kernel<1, false, true>), grid, block, 0, s, p);
        else if (S == 2) hipLaunchKernelGGL((update_rows_kernel<2, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((update_rows_kernel<3, false, true>), grid, block, 0, s, p);
    } else {
        if (S == 1) hipLaunchKernelGGL((update_rows_kernel<1, false, false>), grid, block, 0, s, p);
        else if (S == 2) hipLaunchKernelGGL((update_rows_kernel<2, false, false>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((update_rows_kernel<3, false, false>), grid, block, 0, s, p);
    }
    return check_launch("lstep_update_rows");
}

// update_pe phase 2 with the first layer pre-multiplied into the messages (see update_rows_kernel<S, true>): agg [>= n, ld_agg] =
// [sum of W1a pe rows (176 columns, 172 used) | sum of time features (time_dim)], w1b [176, 112] = pe_mlp_1.weight[:, pe_dim:] zero-padded.
extern "C" int lstep_update_rows_pre(const float* agg, int32_t ld_agg, const int64_t* ids, int64_t n, const float* w1b, const float* b1,
                                     const float* w2, const float* b2, float* table, float* mirror, int32_t pe_dim, int32_t time_dim,
                                     const int32_t* num_live, const lstep_ring_ref_t* ring, int32_t mirror_world, int32_t mirror_rank,
                                     void* stream) {
    if (mirror_world < 1 || mirror_rank < 0 || mirror_rank >= mirror_world) return set_error(LSTEP_EINVAL, "lstep_update_rows_pre: bad mirror shard");
    if (n < 0 || ld_agg < kPp + time_dim || (ld_agg & 3) || pe_dim <= 0 || pe_dim > kPp || (pe_dim & 3) || time_dim <= 0 || time_dim > kTd || (time_dim & 3))
        return set_error(LSTEP_EINVAL, "lstep_update_rows_pre: bad sizes");
    if (n == 0) return LSTEP_OK;
    if (!agg || !ids || !w1b || !b1 || !w2 || !b2 || !table) return set_error(LSTEP_EINVAL, "lstep_update_rows_pre: NULL pointer");
    if (((uintptr_t)mirror) & 15) return set_error(LSTEP_EINVAL, "lstep_update_rows_pre: misaligned mirror table");
    if (ring && (!ring->start || ring->slots <= 0 || ring->add < 0 || (ring->slot_stride & 3)))
        return set_error(LSTEP_EINVAL, "lstep_update_rows_pre: bad ring reference");
    UpdateParams p{agg, ids, w1b, b1, w2, b2, nullptr, nullptr, table, mirror, num_live, ring ? ring->start : nullptr, ring ? ring->add : 0,
                   ring ? ring->slots : 1, ring ? ring->slot_stride : 0, n, ld_agg, pe_dim, time_dim, mirror_world, mirror_rank};
    hipStream_t s = (hipStream_t)stream;
    if (update_split(n)) {
        hipLaunchKernelGGL(update_rows_split_kernel<true>, dim3((unsigned)((n + 15) / 16)), dim3(kBlock), 0, s, p);
        return check_launch("lstep_update_rows_pre<split>");
    }
    {
        // the persistent LDS-resident form: LSTEP_UPDATE_LDS=1 selects it.  Measured and NOT kept as the default (round 4): 449 us against 485 for
        // the slab chain at 290 k rows alone (-7 %), but 3.23-3.26 against 3.20 ms per c4 step
        const char* lds = getenv("LSTEP_UPDATE_LDS");
        if (lds && lds[0] == '1') {
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute((const void*)update_rows_lds_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, kUpdLdsBytes) != hipSuccess ||
                    hipFuncSetAttribute((const void*)update_rows_lds_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, kUpdLdsBytes) != hipSuccess)
                    return set_error(LSTEP_EHIP, "lstep_update_rows_pre: cannot reserve %d bytes of LDS", kUpdLdsBytes);
                attr_set = true;
            }
            const char* wv = getenv("LSTEP_UPDATE_LDS_WAVES");      // tuning: 8 (two waves per SIMD, the faster one) or 12 (three)
            const int waves = (wv && atoi(wv) == 12) ? 12 : 8;
            const int64_t slabs = (n + 15) / 16;
            const unsigned wgs = (unsigned)(slabs < 256 * (int64_t)waves ? (slabs + waves - 1) / waves : 256);
            if (waves == 8) hipLaunchKernelGGL(update_rows_lds_kernel<8>, dim3(wgs), dim3(8 * kWave), kUpdLdsBytes, s, p);
            else hipLaunchKernelGGL(update_rows_lds_kernel<12>, dim3(wgs), dim3(12 * kWave), kUpdLdsBytes, s, p);
            return check_launch("lstep_update_rows_pre<lds>");
        }
    }
    const int S = update_slabs_per_wave(n);
    const int64_t tasks = ((n + 15) / 16 + S - 1) / S;
    const dim3 grid((unsigned)((tasks + kWavesPerBlock - 1) / kWavesPerBlock)), block(kBlock);
    if (S == 1) hipLaunchKernelGGL((update_rows_kernel<1, true>), grid, block, 0, s, p);
    else if (S == 2) hipLaunchKernelGGL((update_rows_kernel<2, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((update_rows_kernel<3, true>), grid, block, 0, s, p);
    return check_launch("lstep_update_rows_pre");
}
