// Link predictor (models/modules.py:42-68 MergeLayer: fc2(relu(fc1(cat[a, b])))) on the embeddings of one batch, without
// materialising any concatenation: the rows of `emb` are [src | dst | negative dst] blocks of n rows each (training, where the
// negative source IS the source, train_LSTEP_link_prediction.py:245) or [src | dst | negative src | negative dst] (evaluation).
//   forward : h = relu(W [emb[first + e] ; emb[second + e]] + b1) for the positive and the negative pair of edge e, logit = w2 . h + b2
//   backward: from d_logit, the gradient of emb (all three row blocks, ready to be the dense tail's grad_out) and the dY operands
//             of the weight gradient (lstep_linear_wgrad).
// W is fc1.weight re-laid as [176, 352] = [first half | second half], both halves zero-padded from 172 to 176 columns / rows.
#include <stdlib.h>
#include <type_traits>

#include "lstep_mma.h"

namespace lstep {

constexpr int kHd = 176, kTh = kHd / 16;   // padded embedding / hidden width
constexpr int kHk = 2 * kHd;               // fc1 input width

struct HeadParams {
    const float* emb;       // [rows, kHd]
    const float* w;         // [kHd, kHk]
    const float* wt;        // [kHk, kHd]  (backward)
    const float *b1, *w2;   // [kHd]
    const float* d_logits;  // [2 n]       (backward)
    float* h;               // [2 n, kHd]  hidden activations (positive pairs, then negative pairs)
    float* logits;          // [2 n]
    float* d_emb;           // [3 n, kHd]  (backward)
    float* d_h;             // [2 n, kHd]  (backward) gradient of the pre-activation hidden layer
    float* d_hsum;          // [n, kHd]    (backward) d_h[pos] + d_h[neg]: the dY operand for the shared first half
    float* dw2_part;        // [waves, kHd] (backward) per-wave partial sums of d_logit * h (the fc2 weight gradient)
    int64_t n;
    int64_t first[2], second[2];   // row offsets into emb of the (positive, negative) pair's first / second half
    const float* b2;        // [1]
};

__global__ __launch_bounds__(kBlock, 1) void head_fwd_kernel(const HeadParams p) {
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const int64_t e0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave_in_block()) * 16;
    if (e0 >= p.n) return;
    int64_t e = e0 + i;
    const bool live = e < p.n;
    if (!live) e = p.n - 1;
    const float* first[2] = {p.emb + (p.first[0] + e) * kHd + 4 * g, p.emb + (p.first[1] + e) * kHd + 4 * g};
    const float* second[2] = {p.emb + (p.second[0] + e) * kHd + 4 * g, p.emb + (p.second[1] + e) * kHd + 4 * g};
    const float* wl = p.w + i * kHk + 4 * g;

    f32x4 h[kTh][2];
#pragma unroll
    for (int t = 0; t < kTh; ++t) {
        const f32x4 bv = ldv4(p.b1 + 16 * t + 4 * g);
        h[t][0] = bv;
        h[t][1] = bv;
    }
    mma_wx<kTh, 2>(h, wl, kHk, kTh, first);
    mma_wx<kTh, 2>(h, wl + kHd, kHk, kTh, second);
    float dot[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < kTh; ++t) {
        const f32x4 wv = ldv4(p.w2 + 16 * t + 4 * g);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                h[t][s][v] = fmaxf(h[t][s][v], 0.f);
                dot[s] = fmaf(h[t][s][v], wv[v], dot[s]);
            }
            if (live) *reinterpret_cast<f32x4*>(p.h + ((int64_t)s * p.n + e) * kHd + 16 * t + 4 * g) = h[t][s];
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {   // the four lane groups hold disjoint feature subsets of row i
        float d = dot[s];
        d += __shfl_xor(d, 16, kWave);
        d += __shfl_xor(d, 32, kWave);
        if (g == 0 && live) p.logits[(int64_t)s * p.n + e] = d + p.b2[0];
    }
}

// training layout only: first[0] == first[1] (the source block), second = the destination / negative blocks
__global__ __launch_bounds__(kBlock, 1) void head_bwd_kernel(const HeadParams p) {
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const int64_t e0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave_in_block()) * 16;
    if (e0 >= p.n) return;
    int64_t e = e0 + i;
    const bool live = e < p.n;
    if (!live) e = p.n - 1;
    const float dl[2] = {p.d_logits[e], p.d_logits[p.n + e]};

    const int64_t wave_id = e0 / 16;
    f32x4 dh[kTh][2], dsum[kTh][1];
#pragma unroll
    for (int t = 0; t < kTh; ++t) {
        const f32x4 wv = ldv4(p.w2 + 16 * t + 4 * g);
        f32x4 gw = f32x4{0.f, 0.f, 0.f, 0.f};   // this row's share of d fc2.weight = sum over rows of d_logit * h
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 hv = ldv4(p.h + ((int64_t)s * p.n + e) * kHd + 16 * t + 4 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                dh[t][s][v] = hv[v] > 0.f ? dl[s] * wv[v] : 0.f;
                if (live) gw[v] = fmaf(dl[s], hv[v], gw[v]);
            }
            if (live) *reinterpret_cast<f32x4*>(p.d_h + ((int64_t)s * p.n + e) * kHd + 16 * t + 4 * g) = dh[t][s];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {   // sum over the 16 rows of the slab (lanes with equal g)
            float x = gw[v];
            x += __shfl_xor(x, 1, kWave); x += __shfl_xor(x, 2, kWave); x += __shfl_xor(x, 4, kWave); x += __shfl_xor(x, 8, kWave);
            gw[v] = x;
        }
        // the padded hidden columns are identically 0, so their slots of the partial row are free: column 172 carries the slab's
        // sum of d_logit, the gradient of fc2.bias
        if (16 * t + 4 * g == 172) {
            float x = live ? dl[0] + dl[1] : 0.f;
            x += __shfl_xor(x, 1, kWave); x += __shfl_xor(x, 2, kWave); x += __shfl_xor(x, 4, kWave); x += __shfl_xor(x, 8, kWave);
            gw[0] = x;
        }
        if (i == 0) *reinterpret_cast<f32x4*>(p.dw2_part + wave_id * kHd + 16 * t + 4 * g) = gw;
        dsum[t][0] = dh[t][0] + dh[t][1];
        if (live) *reinterpret_cast<f32x4*>(p.d_hsum + e * kHd + 16 * t + 4 * g) = dsum[t][0];
    }
    const float* wtl = p.wt + i * kHd + 4 * g;
    {   // d emb[src] = Wt[first-half rows] (d_h[pos] + d_h[neg])
        f32x4 d[kTh][1];
#pragma unroll
        for (int t = 0; t < kTh; ++t) d[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_wr<kTh, 1, kTh>(d, wtl, kHd, dsum);
#pragma unroll
        for (int t = 0; t < kTh; ++t)
            if (live) *reinterpret_cast<f32x4*>(p.d_emb + (p.first[0] + e) * kHd + 16 * t + 4 * g) = d[t][0];
    }
    {   // d emb[dst], d emb[neg] = Wt[second-half rows] d_h[pos / neg]
        f32x4 d[kTh][2];
#pragma unroll
        for (int t = 0; t < kTh; ++t) {
            d[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            d[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        mma_wr<kTh, 2, kTh>(d, wtl + (size_t)kHd * kHd, kHd, dh);
#pragma unroll
        for (int t = 0; t < kTh; ++t) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (live) *reinterpret_cast<f32x4*>(p.d_emb + (p.second[s] + e) * kHd + 16 * t + 4 * g) = d[t][s];
        }
    }
}

// ---- few edges: one 16-edge slab per workgroup, the hidden layer's tiles dealt out to its four waves (see tail.hip) --------------------
constexpr int kHeadSplitMaxSlabs = 512;
constexpr int kSplitTh = (kTh + 3) / 4;   // 3 tiles per wave

__global__ __launch_bounds__(kBlock) void head_fwd_split_kernel(const HeadParams p) {
    __shared__ float s_dot[kWavesPerBlock][2][16];
    const int lane = lane_id(), w = wave_in_block();
    const int i = lane & 15, g = lane >> 4;
    int64_t e = (int64_t)blockIdx.x * 16 + i;
    const bool live = e < p.n;
    if (!live) e = p.n - 1;
    const float* first[2] = {p.emb + (p.first[0] + e) * kHd + 4 * g, p.emb + (p.first[1] + e) * kHd + 4 * g};
    const float* second[2] = {p.emb + (p.second[0] + e) * kHd + 4 * g, p.emb + (p.second[1] + e) * kHd + 4 * g};
    int tl[kSplitTh];
    bool valid[kSplitTh];
    f32x4 h[kSplitTh][2];
#pragma unroll
    for (int j = 0; j < kSplitTh; ++j) {
        valid[j] = w + 4 * j < kTh;
        tl[j] = valid[j] ? w + 4 * j : w;
        const f32x4 bv = ldv4(p.b1 + 16 * tl[j] + 4 * g);
        h[j][0] = bv;
        h[j][1] = bv;
    }
    auto half = [&](const float* const (&x)[2], int col0) {    // h += W[:, col0 : col0 + 176] x, operand chunks double-buffered by hand
        struct Ch { f32x4 a[kSplitTh]; f32x4 b[2]; };
        auto load = [&](Ch& o, int c) {
            o.b[0] = ldv4(x[0] + 16 * c);
            o.b[1] = ldv4(x[1] + 16 * c);
#pragma unroll
            for (int j = 0; j < kSplitTh; ++j) o.a[j] = ldv4(p.w + (size_t)(16 * tl[j] + i) * kHk + col0 + 4 * g + 16 * c);
        };
        auto run = [&](const Ch& o) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
#pragma unroll
                for (int j = 0; j < kSplitTh; ++j) {
                    h[j][0] = mfma4(o.a[j][v], o.b[0][v], h[j][0]);
                    h[j][1] = mfma4(o.a[j][v], o.b[1][v], h[j][1]);
                }
            }
        };
        Ch c0, c1;
        load(c0, 0);
        for (int c = 0; c < kTh; c += 2) {
            const bool two = c + 1 < kTh;
            if (two) load(c1, c + 1);
            __builtin_amdgcn_sched_barrier(0);
            run(c0);
            __builtin_amdgcn_sched_barrier(0);
            if (two) {
                if (c + 2 < kTh) load(c0, c + 2);
                __builtin_amdgcn_sched_barrier(0);
                run(c1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    half(first, 0);
    half(second, kHd);
    float dot[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < kSplitTh; ++j) {
        if (!valid[j]) continue;    // wave-uniform
        const f32x4 wv = ldv4(p.w2 + 16 * tl[j] + 4 * g);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                h[j][s][v] = fmaxf(h[j][s][v], 0.f);
                dot[s] = fmaf(h[j][s][v], wv[v], dot[s]);
            }
            if (live) *reinterpret_cast<f32x4*>(p.h + ((int64_t)s * p.n + e) * kHd + 16 * tl[j] + 4 * g) = h[j][s];
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float d = dot[s];
        d += __shfl_xor(d, 16, kWave);
        d += __shfl_xor(d, 32, kWave);
        if (g == 0) s_dot[w][s][i] = d;
    }
    __syncthreads();
    if (w == 0 && lane < 32) {
        const int s = lane >> 4;
        const float d = ((s_dot[0][s][i] + s_dot[1][s][i]) + s_dot[2][s][i]) + s_dot[3][s][i];
        if (live) p.logits[(int64_t)s * p.n + e] = d + p.b2[0];
    }
}

// training layout only (first[0] == first[1]).  Every wave forms ALL tiles of d_h (element-wise from h: they are the B operand of its
// products) but stores and reduces only its own; the three d_emb products are split by output tile.
__global__ __launch_bounds__(kBlock) void head_bwd_split_kernel(const HeadParams p) {
    const int lane = lane_id(), w = wave_in_block();
    const int i = lane & 15, g = lane >> 4;
    int64_t e = (int64_t)blockIdx.x * 16 + i;
    const bool live = e < p.n;
    if (!live) e = p.n - 1;
    const float dl[2] = {p.d_logits[e], p.d_logits[p.n + e]};
    f32x4 dh[kTh][2], dsum[kTh][1];
#pragma unroll
    for (int t = 0; t < kTh; ++t) {
        const bool mine = (t & 3) == w;   // wave-uniform
        const f32x4 wv = ldv4(p.w2 + 16 * t + 4 * g);
        f32x4 gw = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 hv = ldv4(p.h + ((int64_t)s * p.n + e) * kHd + 16 * t + 4 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                dh[t][s][v] = hv[v] > 0.f ? dl[s] * wv[v] : 0.f;
                if (live) gw[v] = fmaf(dl[s], hv[v], gw[v]);
            }
            if (mine && live) *reinterpret_cast<f32x4*>(p.d_h + ((int64_t)s * p.n + e) * kHd + 16 * t + 4 * g) = dh[t][s];
        }
        dsum[t][0] = dh[t][0] + dh[t][1];
        if (mine) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float x = gw[v];
                x += __shfl_xor(x, 1, kWave); x += __shfl_xor(x, 2, kWave); x += __shfl_xor(x, 4, kWave); x += __shfl_xor(x, 8, kWave);
                gw[v] = x;
            }
            if (16 * t + 4 * g == 172) {      // (see head_bwd_kernel: the padded column carries the slab's sum of d_logit)
                float x = live ? dl[0] + dl[1] : 0.f;
                x += __shfl_xor(x, 1, kWave); x += __shfl_xor(x, 2, kWave); x += __shfl_xor(x, 4, kWave); x += __shfl_xor(x, 8, kWave);
                gw[0] = x;
            }
            if (i == 0) *reinterpret_cast<f32x4*>(p.dw2_part + (int64_t)blockIdx.x * kHd + 16 * t + 4 * g) = gw;
            if (live) *reinterpret_cast<f32x4*>(p.d_hsum + e * kHd + 16 * t + 4 * g) = dsum[t][0];
        }
    }
    int tl[kSplitTh];
    bool valid[kSplitTh];
#pragma unroll
    for (int j = 0; j < kSplitTh; ++j) {
        valid[j] = w + 4 * j < kTh;
        tl[j] = valid[j] ? w + 4 * j : w;
    }
    // acc[j][s] = sum over the hidden tiles tk of Wt[row0 + 16 tl[j] + i][16 tk + k] * r[tk][s]
    auto product = [&](auto& acc, int64_t row0, const auto& r, auto nslab) {
        constexpr int S = decltype(nslab)::value;
        f32x4 a[2][kSplitTh];
#pragma unroll
        for (int j = 0; j < kSplitTh; ++j) a[0][j] = ldv4(p.wt + (size_t)(row0 + 16 * tl[j] + i) * kHd + 4 * g);
#pragma unroll
        for (int tk = 0; tk < kTh; ++tk) {
            if (tk + 1 < kTh) {
#pragma unroll
                for (int j = 0; j < kSplitTh; ++j) a[(tk + 1) & 1][j] = ldv4(p.wt + (size_t)(row0 + 16 * tl[j] + i) * kHd + 4 * g + 16 * (tk + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
#pragma unroll
                for (int j = 0; j < kSplitTh; ++j) {
#pragma unroll
                    for (int s = 0; s < S; ++s) acc[j][s] = mfma4(a[tk & 1][j][v], r[tk][s][v], acc[j][s]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        f32x4 d[kSplitTh][1];
#pragma unroll
        for (int j = 0; j < kSplitTh; ++j) d[j][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        product(d, 0, dsum, std::integral_constant<int, 1>{});
#pragma unroll
        for (int j = 0; j < kSplitTh; ++j)
            if (valid[j] && live) *reinterpret_cast<f32x4*>(p.d_emb + (p.first[0] + e) * kHd + 16 * tl[j] + 4 * g) = d[j][0];
    }
    {
        f32x4 d[kSplitTh][2];
#pragma unroll
        for (int j = 0; j < kSplitTh; ++j) {
            d[j][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            d[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        product(d, kHd, dh, std::integral_constant<int, 2>{});
#pragma unroll
        for (int j = 0; j < kSplitTh; ++j) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (valid[j] && live) *reinterpret_cast<f32x4*>(p.d_emb + (p.second[s] + e) * kHd + 16 * tl[j] + 4 * g) = d[j][s];
        }
    }
}

// fc1.weight [hid, 2 * half], fc1.bias [hid], fc2.weight [1, hid] -> the operands of the kernels above in one launch: w [176, 352] =
// [first half | second half] zero-padded, its transpose wt [352, 176], b1 [176], w2 [176]
__global__ __launch_bounds__(kBlock) void head_pack_kernel(const float* __restrict__ fc1_w, const float* __restrict__ fc1_b,
                                                            const float* __restrict__ fc2_w, int hid, int half, float* __restrict__ w,
                                                            float* __restrict__ wt, float* __restrict__ b1, float* __restrict__ w2) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= kHd * kHk) return;
    const int r = idx / kHk, c = idx - r * kHk;            // row of w (hidden unit), column (input feature of the padded pair)
    const int side = c >= kHd, k = c - side * kHd;
    const float v = (r < hid && k < half) ? fc1_w[(size_t)r * (2 * half) + side * half + k] : 0.f;
    w[idx] = v;
    wt[(size_t)c * kHd + r] = v;
    if (c == 0) {
        b1[r] = r < hid ? fc1_b[r] : 0.f;
        w2[r] = r < hid ? fc2_w[r] : 0.f;
    }
}

static bool head_split(int64_t n) {
    const char* off = getenv("LSTEP_HEAD_NO_SPLIT");   // A/B and the parity test: read per call
    return !(off && off[0] == '1') && (n + 15) / 16 <= kHeadSplitMaxSlabs;
}

}  // namespace lstep

using namespace lstep;

extern "C" int lstep_head_fwd(const float* emb, int64_t n, int64_t pos_first, int64_t pos_second, int64_t neg_first, int64_t neg_second,
                              const float* w, const float* b1, const float* w2, const float* b2, float* h, float* logits, void* stream) {
    if (n < 0 || pos_first < 0 || pos_second < 0 || neg_first < 0 || neg_second < 0) return set_error(LSTEP_EINVAL, "lstep_head_fwd: bad sizes");
    if (n == 0) return LSTEP_OK;
    if (!emb || !w || !b1 || !w2 || !b2 || !h || !logits) return set_error(LSTEP_EINVAL, "lstep_head_fwd: NULL pointer");
    HeadParams p{};
    p.emb = emb; p.w = w; p.b1 = b1; p.w2 = w2; p.h = h; p.logits = logits; p.n = n; p.b2 = b2;
    p.first[0] = pos_first; p.first[1] = neg_first; p.second[0] = pos_second; p.second[1] = neg_second;
    const int64_t tasks = (n + 15) / 16;
    if (head_split(n)) {
        hipLaunchKernelGGL(head_fwd_split_kernel, dim3((unsigned)tasks), dim3(kBlock), 0, (hipStream_t)stream, p);
        return check_launch("lstep_head_fwd<split>");
    }
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((tasks + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, p);
    return check_launch("lstep_head_fwd");
}

extern "C" int lstep_head_bwd(const float* d_logits, const float* h, int64_t n, const float* wt, const float* w2, float* d_emb, float* d_h,
                              float* d_hsum, float* dw2_partial, void* stream) {
    if (n < 0) return set_error(LSTEP_EINVAL, "lstep_head_bwd: bad sizes");
    if (n == 0) return LSTEP_OK;
    if (!d_logits || !h || !wt || !w2 || !d_emb || !d_h || !d_hsum || !dw2_partial) return set_error(LSTEP_EINVAL, "lstep_head_bwd: NULL pointer");
    HeadParams p{};
    p.d_logits = d_logits; p.h = const_cast<float*>(h); p.wt = wt; p.w2 = w2; p.d_emb = d_emb; p.d_h = d_h; p.d_hsum = d_hsum; p.dw2_part = dw2_partial; p.n = n;
    p.first[0] = p.first[1] = 0; p.second[0] = n; p.second[1] = 2 * n;
    const int64_t tasks = (n + 15) / 16;
    if (head_split(n)) {
        hipLaunchKernelGGL(head_bwd_split_kernel, dim3((unsigned)tasks), dim3(kBlock), 0, (hipStream_t)stream, p);
        return check_launch("lstep_head_bwd<split>");
    }
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((tasks + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, p);
    return check_launch("lstep_head_bwd");
}

extern "C" int lstep_head_pack(const float* fc1_w, const float* fc1_b, const float* fc2_w, int32_t hidden, int32_t half, float* w, float* wt,
                               float* b1, float* w2, void* stream) {
    if (hidden <= 0 || hidden > kHd || half <= 0 || half > kHd) return set_error(LSTEP_EINVAL, "lstep_head_pack: bad sizes");
    if (!fc1_w || !fc1_b || !fc2_w || !w || !wt || !b1 || !w2) return set_error(LSTEP_EINVAL, "lstep_head_pack: NULL pointer");
    hipLaunchKernelGGL(head_pack_kernel, dim3((kHd * kHk + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, fc1_w, fc1_b, fc2_w,
                       (int)hidden, (int)half, w, wt, b1, w2);
    return check_launch("lstep_head_pack");
}
