// Weight composition of the dense tail (models/LSTEP.py:161-170,219,240-247,264 re-associated: lstep_amd/model.py `_combined_tail`):
// everything that is NOT a matrix product -- zero-padded copies of the parameters into the 16-aligned operand block, the composed biases,
// the four transposes the backward kernel reads -- in ONE launch per direction, instead of ~20 framework launches (fills, slice copies,
// addcmul / add / addmv, .t().contiguous()) around the three / six small products (lstep_small_gemm).  The operands are rebuilt every
// optimiser step (the parameters change), so at the reference's own batch sizes, where an iteration is a chain of ~100 dependent
// micro-kernels, these launches were a quarter of the step.
//
//   forward  (lstep_tail_weights_pack):   after  M = Wo_a Wn_b,  Wall[:F, :F] = Wo_a Wn_a,  Wall[:F, Fn:Fn+C] = M W2  (three products)
//       W1p  [Ce, Ce]  = edge_mlp_1.weight                      b1p  = (sum a) edge_mlp_1.bias + edge_agg.bias
//       Wn1p [Pp, Cp]  = pe_neighbor_mlp_1.weight               bn1p = pe_neighbor_mlp_1.bias
//       Wq   [Pp, 2Pp] = [self_update_neighbor_pe.weight | pe_neighbor_mlp_2.weight]       bq = sum of their biases
//       Wall [Fn, Fn+Ce+Pp] = [. | . | Wo_b] (the two product blocks are left as the products wrote them)
//       const [Fn] = bo + M b2 + Wo_a bn;   W1p^T, Wn1p^T, Wq^T, Wall^T;   a_sum
//   backward (lstep_tail_weights_unpack): after  dM = dA2 W2^T  (one product), before the five others
//       the slice copies (d W1, d Wn1, d Ws, d Wn2, d Wo_b and the biases), d b1 = a_sum db1p, d a = <db1p, b1>, d ab = sum db1p,
//       d b2 = M^T dc, d bn = Wo_a^T dc, d bo = dc, dM += dc b2^T, d Wo_a = dc bn^T (the two products that follow add to it).
#include "lstep_common.h"

namespace lstep {

struct PackDims {
    int Fd, C, P, CP, Ce, Fn, Cp, Pp, K;
};

struct PackParams {
    const float *W1, *b1, *aw, *ab, *b2, *Wn, *bn, *Wo, *bo, *Ws, *bs, *Wn1, *bn1, *Wn2, *bn2, *M;
    float *W1p, *b1p, *Wn1p, *bn1p, *Wq, *bq, *Wall, *constp;     // the operand block
    float *w1t, *wn1t, *wqt, *wallt;                               // transposes
    float* a_sum;
    PackDims d;
};

__device__ __forceinline__ float sum_small(const float* __restrict__ v, int n) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += v[i];      // (same order on every thread: K <= a few dozen)
    return s;
}

__global__ __launch_bounds__(kBlock) void tail_weights_pack_kernel(const PackParams p) {
    const PackDims d = p.d;
    const int64_t n1 = (int64_t)d.Ce * d.Ce, n2 = (int64_t)d.Pp * d.Cp, n3 = (int64_t)d.Pp * 2 * d.Pp, wall_ld = d.Fn + d.Ce + d.Pp,
                  n4 = (int64_t)d.Fn * wall_ld, nb = d.Ce + d.Pp + d.Pp;
    const int64_t total = n1 + n2 + n3 + n4 + nb;
    const int64_t waves_for_const = d.Fd;       // one wave per row of const
    const int64_t elem_blocks = (total + kBlock - 1) / kBlock;
    if ((int64_t)blockIdx.x >= elem_blocks) {
        // const[f] = bo[f] + <M[f, :C], b2> + <Wo[f, :Fd], bn>   (Wo_a = the first Fd columns of out_node_emb.weight)
        const int lane = lane_id();
        const int64_t f = ((int64_t)blockIdx.x - elem_blocks) * kWavesPerBlock + wave_in_block();
        if (f >= waves_for_const) {
            if (f < d.Fn && lane == 0) p.constp[f] = 0.f;
            return;
        }
        float s = 0.f;
        for (int c = lane; c < d.C; c += kWave) s = fmaf(p.M[f * d.C + c], p.b2[c], s);
        for (int c = lane; c < d.Fd; c += kWave) s = fmaf(p.Wo[f * (int64_t)(d.Fd + d.P) + c], p.bn[c], s);
        s = wave_sum(s);
        if (lane == 0) p.constp[f] = s + p.bo[f];
        return;
    }
    int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total) return;
    if (e < n1) {                                   // W1p [Ce, Ce] and its transpose
        const int i = (int)(e / d.Ce), j = (int)(e % d.Ce);
        const float v = (i < d.C && j < d.C) ? p.W1[(int64_t)i * d.C + j] : 0.f;
        p.W1p[e] = v;
        p.w1t[(int64_t)j * d.Ce + i] = v;
        return;
    }
    e -= n1;
    if (e < n2) {                                   // Wn1p [Pp, Cp]
        const int i = (int)(e / d.Cp), j = (int)(e % d.Cp);
        const float v = (i < d.P && j < d.CP) ? p.Wn1[(int64_t)i * d.CP + j] : 0.f;
        p.Wn1p[e] = v;
        p.wn1t[(int64_t)j * d.Pp + i] = v;
        return;
    }
    e -= n2;
    if (e < n3) {                                   // Wq [Pp, 2 Pp] = [Ws | Wn2]
        const int i = (int)(e / (2 * d.Pp)), j = (int)(e % (2 * d.Pp));
        float v = 0.f;
        if (i < d.P) {
            if (j < d.P) v = p.Ws[(int64_t)i * d.P + j];
            else if (j >= d.Pp && j < d.Pp + d.P) v = p.Wn2[(int64_t)i * d.P + (j - d.Pp)];
        }
        p.Wq[e] = v;
        p.wqt[(int64_t)j * d.Pp + i] = v;
        return;
    }
    e -= n3;
    if (e < n4) {                                   // Wall [Fn, Fn + Ce + Pp]: the two product blocks are final already
        const int i = (int)(e / wall_ld), j = (int)(e % wall_ld);
        float v;
        const bool product = i < d.Fd && (j < d.Fd || (j >= d.Fn && j < d.Fn + d.C));
        if (product) {
            v = p.Wall[e];
        } else {
            v = 0.f;
            if (i < d.Fd && j >= d.Fn + d.Ce && j < d.Fn + d.Ce + d.P) v = p.Wo[(int64_t)i * (d.Fd + d.P) + d.Fd + (j - d.Fn - d.Ce)];
            p.Wall[e] = v;
        }
        p.wallt[(int64_t)j * d.Fn + i] = v;
        return;
    }
    e -= n4;                                        // the bias vectors b1p [Ce], bn1p [Pp], bq [Pp]
    if (e < d.Ce) {
        const float a_sum = sum_small(p.aw, d.K);
        p.b1p[e] = e < d.C ? fmaf(a_sum, p.b1[e], p.ab[0]) : 0.f;
        if (e == 0) p.a_sum[0] = a_sum;
        return;
    }
    e -= d.Ce;
    if (e < d.Pp) { p.bn1p[e] = e < d.P ? p.bn1[e] : 0.f; return; }
    e -= d.Pp;
    p.bq[e] = e < d.P ? p.bs[e] + p.bn2[e] : 0.f;
}

struct UnpackParams {
    // operand gradients (what lstep_linear_wgrad produced) and forward values
    const float *gW1p, *gb1p, *gWn1p, *gbn1p, *gWq, *gbq, *gWall, *gconst;
    const float *b1, *b2, *bn, *Wo, *M, *a_sum;
    // parameter gradients (dense, parameter-shaped) + dM (in place)
    float *d_W1, *d_b1, *d_aw, *d_ab, *d_b2, *d_bn, *d_Wo, *d_bo, *d_Ws, *d_bs, *d_Wn1, *d_bn1, *d_Wn2, *d_bn2, *dM;
    PackDims d;
};

__global__ __launch_bounds__(kBlock) void tail_weights_unpack_kernel(const UnpackParams p) {
    const PackDims d = p.d;
    const int64_t wall_ld = d.Fn + d.Ce + d.Pp;
    const int64_t n1 = (int64_t)d.C * d.C, n2 = (int64_t)d.P * d.CP, n3 = (int64_t)d.P * d.P, n4 = (int64_t)d.Fd * (d.Fd + d.P), n5 = (int64_t)d.Fd * d.C,
                  nb = d.C + d.P + d.P + d.P + d.Fd;
    const int64_t total = n1 + n2 + 2 * n3 + n4 + n5 + nb;
    const int64_t elem_blocks = (total + kBlock - 1) / kBlock;
    const float* dc = p.gconst;                     // d const [:Fd]
    if ((int64_t)blockIdx.x >= elem_blocks) {
        // one wave per output of the three reductions: d b2[c] = sum_f M[f, c] dc[f] (c < C), d bn[c] = sum_f Wo[f, c] dc[f] (c < Fd),
        // and the two scalars d a = <db1p[:C], b1>, d ab = sum db1p[:C]
        const int lane = lane_id();
        int64_t w = ((int64_t)blockIdx.x - elem_blocks) * kWavesPerBlock + wave_in_block();
        if (w < d.C) {
            float s = 0.f;
            for (int f = lane; f < d.Fd; f += kWave) s = fmaf(p.M[(int64_t)f * d.C + w], dc[f], s);
            s = wave_sum(s);
            if (lane == 0) p.d_b2[w] = s;
            return;
        }
        w -= d.C;
        if (w < d.Fd) {
            float s = 0.f;
            for (int f = lane; f < d.Fd; f += kWave) s = fmaf(p.Wo[(int64_t)f * (d.Fd + d.P) + w], dc[f], s);
            s = wave_sum(s);
            if (lane == 0) p.d_bn[w] = s;
            return;
        }
        w -= d.Fd;
        if (w == 0) {
            float s = 0.f, t = 0.f;
            for (int c = lane; c < d.C; c += kWave) { s = fmaf(p.gb1p[c], p.b1[c], s); t += p.gb1p[c]; }
            s = wave_sum(s);
            t = wave_sum(t);
            for (int k = lane; k < d.K; k += kWave) p.d_aw[k] = s;
            if (lane == 0) p.d_ab[0] = t;
        }
        return;
    }
    int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total) return;
    if (e < n1) { p.d_W1[e] = p.gW1p[(e / d.C) * d.Ce + (e % d.C)]; return; }
    e -= n1;
    if (e < n2) { p.d_Wn1[e] = p.gWn1p[(e / d.CP) * d.Cp + (e % d.CP)]; return; }
    e -= n2;
    if (e < n3) { p.d_Ws[e] = p.gWq[(e / d.P) * (2 * d.Pp) + (e % d.P)]; return; }
    e -= n3;
    if (e < n3) { p.d_Wn2[e] = p.gWq[(e / d.P) * (2 * d.Pp) + d.Pp + (e % d.P)]; return; }
    e -= n3;
    if (e < n4) {                                   // d Wo [Fd, Fd + P]: [dc bn^T (the products add to it) | d Wo_b]
        const int f = (int)(e / (d.Fd + d.P)), j = (int)(e % (d.Fd + d.P));
        p.d_Wo[e] = j < d.Fd ? dc[f] * p.bn[j] : p.gWall[(int64_t)f * wall_ld + d.Fn + d.Ce + (j - d.Fd)];
        return;
    }
    e -= n4;
    if (e < n5) {                                   // dM [Fd, C] += dc b2^T
        const int f = (int)(e / d.C), c = (int)(e % d.C);
        p.dM[e] = fmaf(dc[f], p.b2[c], p.dM[e]);
        return;
    }
    e -= n5;
    if (e < d.C) { p.d_b1[e] = p.a_sum[0] * p.gb1p[e]; return; }
    e -= d.C;
    if (e < d.P) { p.d_bn1[e] = p.gbn1p[e]; return; }
    e -= d.P;
    if (e < d.P) { p.d_bs[e] = p.gbq[e]; return; }
    e -= d.P;
    if (e < d.P) { p.d_bn2[e] = p.gbq[e]; return; }
    e -= d.P;
    p.d_bo[e] = dc[e];
}

static bool bad_dims(const int32_t* dims) {
    // (Fd, C, P, CP, Ce, Fn, Cp, Pp, K): paddings at least as wide as what they pad
    return dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0 || dims[3] <= 0 || dims[4] < dims[1] || dims[5] < dims[0] || dims[6] < dims[3] ||
           dims[7] < dims[2] || dims[8] <= 0;
}

}  // namespace lstep

using namespace lstep;

extern "C" int lstep_tail_weights_pack(const float* const* params, const float* M, const int32_t* dims, float* flat, float* flat_t, float* a_sum,
                                       void* stream) {
    if (!params || !M || !dims || !flat || !flat_t || !a_sum) return set_error(LSTEP_EINVAL, "lstep_tail_weights_pack: NULL pointer");
    if (bad_dims(dims)) return set_error(LSTEP_EINVAL, "lstep_tail_weights_pack: bad dims");
    for (int i = 0; i < 16; ++i)
        if (!params[i]) return set_error(LSTEP_EINVAL, "lstep_tail_weights_pack: NULL parameter %d", i);
    const PackDims d{dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], dims[6], dims[7], dims[8]};
    PackParams p;
    // params: W1, b1, aw, ab, W2, b2, Wn, bn, Wo, bo, Ws, bs, Wn1, bn1, Wn2, bn2   (the order of model._tail_weights_forward)
    p.W1 = params[0]; p.b1 = params[1]; p.aw = params[2]; p.ab = params[3]; p.b2 = params[5]; p.Wn = params[6]; p.bn = params[7];
    p.Wo = params[8]; p.bo = params[9]; p.Ws = params[10]; p.bs = params[11]; p.Wn1 = params[12]; p.bn1 = params[13]; p.Wn2 = params[14];
    p.bn2 = params[15]; p.M = M;
    const int64_t wall_ld = d.Fn + d.Ce + d.Pp;
    const int64_t sizes[8] = {(int64_t)d.Ce * d.Ce, d.Ce, (int64_t)d.Pp * d.Cp, d.Pp, (int64_t)d.Pp * 2 * d.Pp, d.Pp, (int64_t)d.Fn * wall_ld, d.Fn};
    float* q = flat;
    p.W1p = q; q += sizes[0]; p.b1p = q; q += sizes[1]; p.Wn1p = q; q += sizes[2]; p.bn1p = q; q += sizes[3];
    p.Wq = q; q += sizes[4]; p.bq = q; q += sizes[5]; p.Wall = q; q += sizes[6]; p.constp = q;
    q = flat_t;
    p.w1t = q; q += sizes[0]; p.wn1t = q; q += sizes[2]; p.wqt = q; q += sizes[4]; p.wallt = q;
    p.a_sum = a_sum;
    p.d = d;
    const int64_t total = sizes[0] + sizes[2] + sizes[4] + sizes[6] + d.Ce + 2 * d.Pp;
    const int64_t blocks = (total + kBlock - 1) / kBlock + (d.Fn + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(tail_weights_pack_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p);
    return check_launch("tail_weights_pack_kernel");
}

extern "C" int lstep_tail_weights_unpack(const float* const* grads_in, const float* const* fwd, float* const* grads_out, float* dM, const int32_t* dims,
                                         void* stream) {
    if (!grads_in || !fwd || !grads_out || !dM || !dims) return set_error(LSTEP_EINVAL, "lstep_tail_weights_unpack: NULL pointer");
    if (bad_dims(dims)) return set_error(LSTEP_EINVAL, "lstep_tail_weights_unpack: bad dims");
    for (int i = 0; i < 8; ++i)
        if (!grads_in[i]) return set_error(LSTEP_EINVAL, "lstep_tail_weights_unpack: NULL operand gradient %d", i);
    for (int i = 0; i < 6; ++i)
        if (!fwd[i]) return set_error(LSTEP_EINVAL, "lstep_tail_weights_unpack: NULL forward value %d", i);
    for (int i = 0; i < 14; ++i)
        if (!grads_out[i]) return set_error(LSTEP_EINVAL, "lstep_tail_weights_unpack: NULL destination %d", i);
    const PackDims d{dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], dims[6], dims[7], dims[8]};
    UnpackParams p;
    p.gW1p = grads_in[0]; p.gb1p = grads_in[1]; p.gWn1p = grads_in[2]; p.gbn1p = grads_in[3]; p.gWq = grads_in[4]; p.gbq = grads_in[5];
    p.gWall = grads_in[6]; p.gconst = grads_in[7];
    p.b1 = fwd[0]; p.b2 = fwd[1]; p.bn = fwd[2]; p.Wo = fwd[3]; p.M = fwd[4]; p.a_sum = fwd[5];
    // grads_out: d_W1, d_b1, d_aw, d_ab, d_b2, d_bn, d_Wo, d_bo, d_Ws, d_bs, d_Wn1, d_bn1, d_Wn2, d_bn2
    p.d_W1 = grads_out[0]; p.d_b1 = grads_out[1]; p.d_aw = grads_out[2]; p.d_ab = grads_out[3]; p.d_b2 = grads_out[4]; p.d_bn = grads_out[5];
    p.d_Wo = grads_out[6]; p.d_bo = grads_out[7]; p.d_Ws = grads_out[8]; p.d_bs = grads_out[9]; p.d_Wn1 = grads_out[10]; p.d_bn1 = grads_out[11];
    p.d_Wn2 = grads_out[12]; p.d_bn2 = grads_out[13];
    p.dM = dM;
    p.d = d;
    const int64_t total = (int64_t)d.C * d.C + (int64_t)d.P * d.CP + 2 * (int64_t)d.P * d.P + (int64_t)d.Fd * (d.Fd + d.P) + (int64_t)d.Fd * d.C + d.C + 3 * d.P + d.Fd;
    const int64_t blocks = (total + kBlock - 1) / kBlock + (d.C + d.Fd + 1 + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(tail_weights_unpack_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p);
    return check_launch("tail_weights_unpack_kernel");
}
