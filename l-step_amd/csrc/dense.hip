// Dense tail of the path (row O and the MLPs of rows A/N/C in SURVEY.md section 8a): weight gradients on the fp32 matrix cores.
//
//   dW[n][k] = sum_r dY[r][n] * X[r][k]        db[n] = sum_r dY[r][n]          (torch.nn.Linear: y = x W^T + b, W is [N, K])
//
// r runs over the 3 * batch rows of a training iteration (49 152 at the bench workload) while the output is a small
// [N <= 288, K <= 640] matrix: the library's fp32 kernels reach 20-25 TFLOP/s on this shape (a 2 x 1 tile grid leaves most of the
// chip idle, and a split-M bmm pays for it with extra launches).  Here every workgroup owns a slice of the rows and keeps its
// whole [N, K] block of partial sums in accumulator registers (v_mfma_f32_16x16x4_f32, exact fp32): the operands go from
// global memory straight into the MFMA operand registers -- both are row-major with the reduction index r as the row, which is
// exactly the A[i][k] / B[k][j] lane layout of the 16x16x4 form (lane l: k = l >> 4, i or j = l & 15), so there is no LDS stage
// and no transpose anywhere.  A second small kernel adds the per-workgroup partials (deterministic, no atomics).
#include "lstep_common.h"

namespace lstep {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
    const float* dy;
    const float* x;
    float* part;
    int64_t m;
    int64_t rows_per_wg;   // multiple of 16
    int64_t part_stride;   // floats per partial block
    int32_t ldy, ldx, n, k;
    int32_t bias_off;      // offset of the db partial inside a partial block
};

// One workgroup = 4 waves; wave w owns k-tiles [(blockIdx.y * 4 + w) * KTW, +KTW) x n-tiles [blockIdx.z * NT, +NT).
// DEPTH row-steps (4 rows each) of operands are in flight per wave: one step is NT * KTW MFMAs = 32 * NT * KTW cycles, a load
// from HBM takes a few thousand.
template <int NT, int KTW, int DEPTH>
__global__ __launch_bounds__(kBlock, 1) void wgrad_partial_kernel(const WgradParams p) {
    struct Operands { float a[NT]; float b[KTW]; };

    const int lane = lane_id(), wave = wave_in_block();
    const int kk = lane >> 4, c = lane & 15;
    const int n0 = blockIdx.z * (NT * 16);
    const int k0 = (blockIdx.y * kWavesPerBlock + wave) * (KTW * 16);
    if (k0 >= p.k) return;  // no barriers in this kernel
    const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_wg;   // the host launches no empty slice: r_begin < m
    int64_t r_end = r_begin + p.rows_per_wg;
    if (r_end > p.m) r_end = p.m;

    // column of every operand tile, clamped into the matrix: tiles (or lanes) past N / K read a valid column and produce
    // values that are never stored
    int colA[NT], colB[KTW];
#pragma unroll
    for (int a = 0; a < NT; ++a) colA[a] = min(n0 + 16 * a + c, p.n - 1);
#pragma unroll
    for (int b = 0; b < KTW; ++b) colB[b] = min(k0 + 16 * b + c, p.k - 1);

    // Loads of one 4-row step; rows past the slice are clamped to its last row (consume() masks them).
    // Plain loads: the compiler places the wait counts.  (An earlier version issued them as inline-asm `global_load_dword` with
    // hand-placed `s_waitcnt vmcnt(N)`: exact when the kernel ran alone, but garbage in the products whenever another stream kept the
    // memory system busy -- the compiler cannot see asm loads in flight when it assigns registers -- `tools/wgrad_race.py`.  With the
    // sched barriers below the compiler-managed version keeps the same software pipeline and runs at the same speed.)
    auto issue = [&](Operands& o, int64_t r) {
        int64_t row = r + kk;
        if (row > r_end - 1) row = r_end - 1;
        const float* ra = p.dy + row * p.ldy;
        const float* rb = p.x + row * p.ldx;
#pragma unroll
        for (int a = 0; a < NT; ++a) o.a[a] = ra[colA[a]];
#pragma unroll
        for (int b = 0; b < KTW; ++b) o.b[b] = rb[colB[b]];
    };

    f32x4 acc[NT][KTW];
    float bsum[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        bsum[a] = 0.f;
#pragma unroll
        for (int b = 0; b < KTW; ++b) {
            acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc[a][b]));   // one distinct accumulator tile each (see the note on register shuffles in DESIGN.md)
        }
    }
    const int64_t steps = (r_end - r_begin + 3) >> 2;
    Operands buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(buf[d], r_begin + 4 * d);
    for (int64_t s = 0; s < steps; s += DEPTH) {   // the last round may run up to DEPTH - 1 all-zero steps
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // the sched barriers keep the loads of step s + d + DEPTH - 1 ahead of the MFMAs of step s + d
            issue(buf[(d + DEPTH - 1) % DEPTH], r_begin + 4 * (s + d + DEPTH - 1));
            __builtin_amdgcn_sched_barrier(0);
            const bool live = r_begin + 4 * (s + d) + kk < r_end;   // dead rows: zero dY operand, the products vanish
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const float va = live ? buf[d].a[a] : 0.f;
                bsum[a] += va;
#pragma unroll
                for (int b = 0; b < KTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, buf[d].b[b], acc[a][b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // C/D layout of the 16x16 forms: lane l, register v -> row 4 * (l >> 4) + v, column l & 15
    float* out = p.part + (int64_t)blockIdx.x * p.part_stride;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int b = 0; b < KTW; ++b) {
            const int k = k0 + 16 * b + c;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int n = n0 + 16 * a + 4 * kk + v;
                if (n < p.n && k < p.k) out[(int64_t)n * p.k + k] = acc[a][b][v];
            }
        }
    }
    if (blockIdx.y == 0 && wave == 0) {
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            float t = bsum[a];
            t += __shfl_xor(t, 16, kWave);
            t += __shfl_xor(t, 32, kWave);
            const int n = n0 + 16 * a + c;
            if (kk == 0 && n < p.n) out[p.bias_off + n] = t;
        }
    }
}

// out = sum over the S partial blocks; 64 float4 columns x 4 partial groups per workgroup
__global__ __launch_bounds__(kBlock) void wgrad_reduce_kernel(const float* __restrict__ part, int64_t part_stride, int32_t num_part, int32_t n,
                                                              int32_t k, int32_t bias_off, float* __restrict__ dw, int32_t ld_dw,
                                                              float* __restrict__ db) {
    __shared__ float4 sh[kBlock];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t e = ((int64_t)blockIdx.x * 64 + col) * 4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < part_stride) {
        const float* src = part + e;
#pragma unroll 8
        for (int s = grp; s < num_part; s += 4) {
            const float4 v = ld4_stream(src + (int64_t)s * part_stride);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
    }
    sh[threadIdx.x] = sum;
    __syncthreads();
    if (grp != 0 || e >= part_stride) return;
#pragma unroll
    for (int g = 1; g < 4; ++g) {
        const float4 v = sh[g * 64 + col];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const float vals[4] = {sum.x, sum.y, sum.z, sum.w};
    const int64_t nk = (int64_t)n * k;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = e + j;
        if (i < nk) dw[(i / k) * ld_dw + (i % k)] = vals[j];
        else if (db != nullptr && i >= bias_off && i < bias_off + n) db[i - bias_off] = vals[j];
    }
}

static inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

struct WgradPlan {
    int nt, ktw;          // template instance
    int gy, gz, splits;   // grid
    int64_t rows_per_wg, part_stride;
    int32_t bias_off;
};

static WgradPlan wgrad_plan(int64_t m, int32_t n, int32_t k) {
    WgradPlan pl;
    const int ntiles = (n + 15) / 16, ktiles = (k + 15) / 16;
    // n-tiles per workgroup: 11 (N = 176) or 9 (N = 288 in two halves), whichever wastes fewer tile slots
    const int waste11 = (ntiles + 10) / 11 * 11 - ntiles, waste9 = (ntiles + 8) / 9 * 9 - ntiles;
    pl.nt = waste9 < waste11 ? 9 : 11;
    pl.gz = (ntiles + pl.nt - 1) / pl.nt;
    // k-tiles per wave: 5 or 3 (NT * KTW * 4 accumulator registers per lane; 256 is the ceiling)
    const int slots5 = (ktiles + 19) / 20 * 20, slots3 = (ktiles + 11) / 12 * 12;
    pl.ktw = slots3 < slots5 ? 3 : 5;
    pl.gy = (ktiles + 4 * pl.ktw - 1) / (4 * pl.ktw);
    int splits = 256 / (pl.gy * pl.gz);  // one workgroup per CU
    if (splits < 1) splits = 1;
    pl.rows_per_wg = round_up((m + splits - 1) / splits, 16);   // whole software-pipeline rounds (4 rows x depth 4)
    if (pl.rows_per_wg < 16) pl.rows_per_wg = 16;
    pl.splits = (int)((m + pl.rows_per_wg - 1) / pl.rows_per_wg);
    if (pl.splits < 1) pl.splits = 1;
    pl.bias_off = (int32_t)round_up((int64_t)n * k, 4);
    pl.part_stride = pl.bias_off + round_up(n, 4);
    return pl;
}

}  // namespace lstep

using namespace lstep;

extern "C" int64_t lstep_linear_wgrad_workspace(int64_t m, int32_t n, int32_t k) {
    if (m < 0 || n <= 0 || k <= 0) return 0;
    const WgradPlan pl = wgrad_plan(m, n, k);
    return (int64_t)pl.splits * pl.part_stride * (int64_t)sizeof(float);
}

extern "C" int lstep_linear_wgrad(const float* dy, int32_t ldy, const float* x, int32_t ldx, int64_t m, int32_t n, int32_t k, float* dw,
                                  int32_t ld_dw, float* db, void* workspace, int64_t workspace_bytes, void* stream) {
    if (m < 0 || n <= 0 || k <= 0 || ldy < n || ldx < k || ld_dw < k) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: bad sizes");
    if ((int64_t)n * k > ((int64_t)1 << 30)) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: weight matrix too large");
    if (!dw || !workspace || (m > 0 && (!dy || !x))) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: NULL pointer");
    if (workspace_bytes < lstep_linear_wgrad_workspace(m, n, k)) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: workspace too small");
    if (((uintptr_t)workspace & 15) != 0) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const WgradPlan pl = wgrad_plan(m, n, k);
    WgradParams p;
    p.dy = dy; p.x = x; p.part = (float*)workspace;
    p.m = m; p.rows_per_wg = pl.rows_per_wg; p.part_stride = pl.part_stride;
    p.ldy = ldy; p.ldx = ldx; p.n = n; p.k = k; p.bias_off = pl.bias_off;
    const dim3 grid((unsigned)pl.splits, (unsigned)pl.gy, (unsigned)pl.gz), block(kBlock);
    if (pl.nt == 11 && pl.ktw == 5) hipLaunchKernelGGL((wgrad_partial_kernel<11, 5, 4>), grid, block, 0, s, p);
    else if (pl.nt == 11) hipLaunchKernelGGL((wgrad_partial_kernel<11, 3, 4>), grid, block, 0, s, p);
    else if (pl.ktw == 5) hipLaunchKernelGGL((wgrad_partial_kernel<9, 5, 4>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((wgrad_partial_kernel<9, 3, 4>), grid, block, 0, s, p);
    const unsigned rgrid = (unsigned)((pl.part_stride / 4 + 63) / 64);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), block, 0, s, (const float*)workspace, pl.part_stride, (int32_t)pl.splits, n, k,
                       pl.bias_off, dw, ld_dw, db);
    return check_launch("lstep_linear_wgrad");
}

// ---- small dense products between weight-sized matrices (<= a few hundred rows / columns): the weight composition of the dense
// tail and its backward (model._TailWeights).  The library runs these 8 - 25 MFLOP products as ONE workgroup (53 - 60 us each); here
// every 16 x 16 output tile is its own wave on the fp32 matrix cores.  General element strides, so transposes are free.
namespace lstep {

struct SmallGemmParams {
    const float *a, *b;
    float* c;
    int32_t m, n, k;
    int64_t sa_i, sa_k, sb_k, sb_j, sc_i, sc_j;
    float alpha, beta;
};

__global__ __launch_bounds__(kBlock) void small_gemm_kernel(const SmallGemmParams p) {
    const int lane = lane_id();
    const int tiles_n = (p.n + 15) / 16;
    const int tile = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (tile >= tiles_n * ((p.m + 15) / 16)) return;
    const int i0 = (tile / tiles_n) * 16, j0 = (tile % tiles_n) * 16;
    const int r = lane & 15, q = lane >> 4;
    const int ai = min(i0 + r, p.m - 1), bj = min(j0 + r, p.n - 1);   // clamped: out-of-range rows / columns are computed and dropped
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int kSteps = 8;     // 8 k-steps (32 k values) of operands in flight: the loop is latency-bound otherwise
    const float* pa = p.a + ai * p.sa_i;
    const float* pb = p.b + bj * p.sb_j;
    for (int k0 = 0; k0 < p.k; k0 += 4 * kSteps) {
        float av[kSteps], bv[kSteps];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const int kk = k0 + 4 * u + q;
            const bool ok = kk < p.k;
            const int kc = ok ? kk : p.k - 1;
            av[u] = pa[kc * p.sa_k];
            bv[u] = pb[kc * p.sb_k];
            if (!ok) av[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < kSteps; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = i0 + 4 * q + v, j = j0 + r;
        if (i < p.m && j < p.n) {
            float* dst = p.c + i * p.sc_i + j * p.sc_j;
            *dst = p.beta == 0.f ? p.alpha * acc[v] : p.alpha * acc[v] + p.beta * *dst;
        }
    }
}

}  // namespace lstep

extern "C" int lstep_small_gemm(const float* a, int64_t sa_i, int64_t sa_k, const float* b, int64_t sb_k, int64_t sb_j, float* c, int64_t sc_i,
                                int64_t sc_j, int32_t m, int32_t n, int32_t k, float alpha, float beta, void* stream) {
    if (m < 0 || n < 0 || k < 0) return set_error(LSTEP_EINVAL, "lstep_small_gemm: negative size");
    if (m == 0 || n == 0) return LSTEP_OK;
    if (!c || (k > 0 && (!a || !b))) return set_error(LSTEP_EINVAL, "lstep_small_gemm: NULL pointer");
    if (k == 0) return set_error(LSTEP_EINVAL, "lstep_small_gemm: empty contraction");
    SmallGemmParams p{a, b, c, m, n, k, sa_i, sa_k, sb_k, sb_j, sc_i, sc_j, alpha, beta};
    const int tiles = ((m + 15) / 16) * ((n + 15) / 16);
    hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)((tiles + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, p);
    return check_launch("lstep_small_gemm");
}
