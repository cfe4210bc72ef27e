// The scalar end of a training iteration (train_LSTEP_link_prediction.py:257-275) and its gradient in one pass:
//   lp_loss = BCE(sigmoid(logit).clamp(0, 1), [1]*n + [0]*n)                               (mean over 2 n)
//   pe_loss = MSE(e_src, e_dst) - neg_weight * MSE(e_src, e_neg)                            (means over n * P)
//   loss    = (1 - pe_weight) * lp_loss + pe_weight * pe_loss
// with e_x = the CURRENT positional-encoding row of node x: the spliced FFT row rows[slot_of[x]] when x is a batch node (that row
// carries gradient), else the constant table row.  The ~25 small launches of the framework's forward and the ~35 of its backward
// become one launch + a fixed-order reduction of the per-workgroup partial sums.
// The gradient of the spliced rows leaves as one row PER OCCURRENCE (edge i: its source, destination and negative endpoint), streamed out:
// the caller reduces the rows by spliced row with lstep_segment_rows_sum over the grouping of cat[src, dst] it already has (deterministic,
// hub-safe).  Float atomics into the spliced rows instead (8.4 M of them at B = 16384) cost 72 us of the 1 M-node step.
#include "lstep_common.h"

namespace lstep {

struct LinkLossParams {
    const float* logits;     // [2 n]
    const int64_t* ids;      // [3 n]  src | dst | neg
    const float* table;      // [N + 1, P]
    const float* rows;       // [U, P]
    const int32_t* slot_of;  // [N + 1]
    float* predicts;         // [2 n]
    float* d_logits;         // [2 n]
    float* g_rows;           // [3 n, P] gradient of e_src (row i), e_dst (n + i), e_neg (2 n + i) of edge i
    int32_t* neg_slot;       // [n] slot_of[negative endpoint of edge i]
    float* partial;          // [grid, 3]
    int64_t n;
    int32_t pe_dim;
    float pe_weight, neg_weight;
};

__device__ __forceinline__ float bce_term(float p, float y) {
    // torch clamps both logarithms at -100 (binary_cross_entropy)
    const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
    return -(y * lp + (1.f - y) * lq);
}

// d loss / d logit through clamp(0, 1) (identity on [0, 1]) and sigmoid, written the way autograd composes it
__device__ __forceinline__ float bce_grad_logit(float p, float y, float scale) {
    const float g_p = (p - y) / fmaxf((1.f - p) * p, 1e-12f) * scale;
    return g_p * (1.f - p) * p;
}

__global__ __launch_bounds__(kBlock) void link_loss_kernel(const LinkLossParams q) {
    __shared__ float sh[kWavesPerBlock][3];
    const int lane = lane_id(), wave = wave_in_block();
    const int P = q.pe_dim;
    float bce = 0.f, sp = 0.f, sn = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave; i < q.n; i += (int64_t)gridDim.x * kWavesPerBlock) {
        if (lane < 2) {   // the positive and the negative logit of edge i
            const int64_t j = lane == 0 ? i : q.n + i;
            const float y = lane == 0 ? 1.f : 0.f;
            const float p = fminf(fmaxf(1.f / (1.f + expf(-q.logits[j])), 0.f), 1.f);
            q.predicts[j] = p;
            bce += bce_term(p, y);
            q.d_logits[j] = bce_grad_logit(p, y, (1.f - q.pe_weight) / (float)(2 * q.n));
        }
        const int64_t a = LSTEP_CHECKED(q.ids[i], LSTEP_NODE_ROWS(), kCheckLossNode), b = LSTEP_CHECKED(q.ids[q.n + i], LSTEP_NODE_ROWS(), kCheckLossNode),
                      c = LSTEP_CHECKED(q.ids[2 * q.n + i], LSTEP_NODE_ROWS(), kCheckLossNode);      // (checked builds only: lstep_common.h)
        const int sa = q.slot_of[a], sb = q.slot_of[b], sc = q.slot_of[c];
        const float* ra = sa >= 0 ? q.rows + (int64_t)sa * P : q.table + a * P;
        const float* rb = sb >= 0 ? q.rows + (int64_t)sb * P : q.table + b * P;
        const float* rc = sc >= 0 ? q.rows + (int64_t)sc * P : q.table + c * P;
        const float gs = q.pe_weight * 2.f / ((float)q.n * (float)P);
        for (int c0 = lane * 4; c0 < P; c0 += kWave * 4) {
            const float4 va = ld4(ra + c0), vb = ld4(rb + c0), vc = ld4(rc + c0);
            const float dpx = va.x - vb.x, dpy = va.y - vb.y, dpz = va.z - vb.z, dpw = va.w - vb.w;
            const float dnx = va.x - vc.x, dny = va.y - vc.y, dnz = va.z - vc.z, dnw = va.w - vc.w;
            sp += dpx * dpx + dpy * dpy + dpz * dpz + dpw * dpw;
            sn += dnx * dnx + dny * dny + dnz * dnz + dnw * dnw;
            const float w = q.neg_weight;
            st4(q.g_rows + i * P + c0, make_float4(gs * (dpx - w * dnx), gs * (dpy - w * dny), gs * (dpz - w * dnz), gs * (dpw - w * dnw)));
            st4(q.g_rows + (q.n + i) * P + c0, make_float4(-gs * dpx, -gs * dpy, -gs * dpz, -gs * dpw));
            st4(q.g_rows + (2 * q.n + i) * P + c0, make_float4(gs * w * dnx, gs * w * dny, gs * w * dnz, gs * w * dnw));
        }
        if (lane == 0) q.neg_slot[i] = sc;
    }
    bce = wave_sum(bce); sp = wave_sum(sp); sn = wave_sum(sn);
    if (lane == 0) { sh[wave][0] = bce; sh[wave][1] = sp; sh[wave][2] = sn; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) t += sh[w][threadIdx.x];
        q.partial[(int64_t)blockIdx.x * 3 + threadIdx.x] = t;
    }
}

// losses[0..2] = lp_loss, pe_loss, loss: one wave, fixed summation order
__global__ void link_loss_finish_kernel(const float* __restrict__ partial, int num_partial, int64_t n, int32_t pe_dim, float pe_weight,
                                        float neg_weight, float* __restrict__ losses) {
    const int lane = threadIdx.x;
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < num_partial; i += kWave) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += partial[(int64_t)i * 3 + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = wave_sum(s[k]);
    if (lane == 0) {
        const float lp = s[0] / (float)(2 * n);
        const float pe = s[1] / ((float)n * (float)pe_dim) - neg_weight * (s[2] / ((float)n * (float)pe_dim));
        losses[0] = lp;
        losses[1] = pe;
        losses[2] = (1.f - pe_weight) * lp + pe_weight * pe;
    }
}

}  // namespace lstep

using namespace lstep;

extern "C" int64_t lstep_link_loss_workspace(int64_t n) {
    const int64_t blocks = n <= 0 ? 1 : ((n + kWavesPerBlock - 1) / kWavesPerBlock < 2048 ? (n + kWavesPerBlock - 1) / kWavesPerBlock : 2048);
    return blocks * 3 * (int64_t)sizeof(float);
}

extern "C" int lstep_link_loss(const float* logits, const int64_t* ids, int64_t n, const float* table, const float* rows, const int32_t* slot_of,
                               int32_t pe_dim, float pe_weight, float neg_weight, float* predicts, float* d_logits, float* g_rows, int32_t* neg_slot,
                               float* losses, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n <= 0 || pe_dim <= 0 || (pe_dim & 3)) return set_error(LSTEP_EINVAL, "lstep_link_loss: bad sizes (n > 0, pe_dim a multiple of 4)");
    if (!logits || !ids || !table || !rows || !slot_of || !predicts || !d_logits || !g_rows || !neg_slot || !losses || !workspace)
        return set_error(LSTEP_EINVAL, "lstep_link_loss: NULL pointer");
    if (((uintptr_t)g_rows) & 15) return set_error(LSTEP_EINVAL, "lstep_link_loss: g_rows must be 16-byte aligned");
    if (workspace_bytes < lstep_link_loss_workspace(n)) return set_error(LSTEP_EINVAL, "lstep_link_loss: workspace too small");
    const int blocks = (int)(lstep_link_loss_workspace(n) / (3 * sizeof(float)));
    LinkLossParams q{logits, ids, table, rows, slot_of, predicts, d_logits, g_rows, neg_slot, (float*)workspace, n, pe_dim, pe_weight, neg_weight};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(link_loss_kernel, dim3(blocks), dim3(kBlock), 0, s, q);
    hipLaunchKernelGGL(link_loss_finish_kernel, dim3(1), dim3(kWave), 0, s, (const float*)workspace, blocks, n, pe_dim, pe_weight, neg_weight, losses);
    return check_launch("lstep_link_loss");
}
