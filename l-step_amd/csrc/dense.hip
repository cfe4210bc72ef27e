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
#include <stdlib.h>

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
    int32_t k_blocks;      // v2 kernels: wave task t of a row slice owns n-block t / k_blocks and k-block t % k_blocks (0: the (y, z) grid mapping)
    int32_t tasks;         // with k_blocks: wave-tasks per row slice; waves are numbered over (slice, task) so that no SIMD slot idles
    int32_t slices;
    int32_t no_fast;       // 0 with LSTEP_WGRAD_FAST=1: whole steps take the clamp-free path of wgrad_partial_v2_body (off by default, see there)
};

static int32_t wgrad_no_fast() {
    static const int32_t v = getenv("LSTEP_WGRAD_FAST") != nullptr ? 0 : 1;
    return v;
}

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

// ---- The same product with 16-byte operand loads wherever a wave's slice holds whole 64-column groups.
// The 16x16x4 form wants, per lane (c = l & 15, kk = l >> 4), ONE element A[i = c][k = kk] = dY[r + kk][column of tile row c].  Nothing says
// the 16 tile rows must be 16 CONSECUTIVE columns: lane c loads the float4 dY[r + kk][n0 + 4c .. 4c + 3], and component v of all lanes is
// the A operand of the tile whose row i is column n0 + 4i + v -- a 64-column group is four interleaved tiles fed by one
// global_load_dwordx4 (16 lanes = 256 contiguous bytes of a row) instead of four global_load_dword (4 x 64 bytes).  The same for X.
// What that buys is depth: a wave tracks at most 63 vector-memory instructions in flight (vmcnt), and the dword pipeline above spends 16
// of them per 4-row step.  Here a step costs 5-8 load instructions, so two blocks of 4 steps fit (see the loop below).
// Slices that are not a multiple of 64 columns keep dword loads for their last 1-3 tiles (NR / KR), so no tile slot is wasted.
// (the body is a device function: the one-product kernel and the batched kernel -- several products of one backward pass in ONE launch,
// lstep_linear_wgrad_batch -- share it; `gid` is the wave's flat (slice, task) number inside its product, `bx / by / bz` the grid mapping
// of the un-flattened plans)
template <int NG, int NR, int KG, int KR, int DEPTH>
__device__ __forceinline__ void wgrad_partial_v2_body(const WgradParams& p, int gid, int bx, int by, int bz) {
    constexpr int NT = 4 * NG + NR, KTW = 4 * KG + KR;
    struct Operands { f32x4 a4[NG > 0 ? NG : 1]; float a1[NR > 0 ? NR : 1]; f32x4 b4[KG > 0 ? KG : 1]; float b1[KR > 0 ? KR : 1]; };

    const int lane = lane_id(), wave = wave_in_block();
    const int kk = lane >> 4, c = lane & 15;
    int task = by * kWavesPerBlock + wave, slice = bx;
    if (p.k_blocks) {      // flattened: 4 consecutive (slice, task) pairs per workgroup
        slice = gid / p.tasks;
        task = gid - slice * p.tasks;
        if (slice >= p.slices) return;
    }
    const int nblock = p.k_blocks ? task / p.k_blocks : bz;
    const int kblock = p.k_blocks ? task % p.k_blocks : task;
    const int n0 = nblock * (NT * 16);
    const int k0 = kblock * (KTW * 16);
    if (k0 >= p.k || n0 >= p.n) return;  // no barriers in this kernel
    const int64_t r_begin = (int64_t)slice * p.rows_per_wg;
    int64_t r_end = r_begin + p.rows_per_wg;
    if (r_end > p.m) r_end = p.m;

    // columns, clamped into the matrix (n and k are multiples of 4 here): lanes / tiles past the edge read valid memory and produce
    // values that are never stored
    int colA4[NG > 0 ? NG : 1], colA1[NR > 0 ? NR : 1], colB4[KG > 0 ? KG : 1], colB1[KR > 0 ? KR : 1];
#pragma unroll
    for (int g = 0; g < NG; ++g) colA4[g] = min(n0 + 64 * g + 4 * c, p.n - 4);
#pragma unroll
    for (int t = 0; t < NR; ++t) colA1[t] = min(n0 + 64 * NG + 16 * t + c, p.n - 1);
#pragma unroll
    for (int g = 0; g < KG; ++g) colB4[g] = min(k0 + 64 * g + 4 * c, p.k - 4);
#pragma unroll
    for (int t = 0; t < KR; ++t) colB1[t] = min(k0 + 64 * KG + 16 * t + c, p.k - 1);

    auto issue = [&](Operands& o, int64_t r) {
        int64_t row = r + kk;
        if (row > r_end - 1) row = r_end - 1;
        const float* ra = p.dy + row * p.ldy;
        const float* rb = p.x + row * p.ldx;
#pragma unroll
        for (int g = 0; g < NG; ++g) o.a4[g] = *reinterpret_cast<const f32x4*>(ra + colA4[g]);
#pragma unroll
        for (int t = 0; t < NR; ++t) o.a1[t] = ra[colA1[t]];
#pragma unroll
        for (int g = 0; g < KG; ++g) o.b4[g] = *reinterpret_cast<const f32x4*>(rb + colB4[g]);
#pragma unroll
        for (int t = 0; t < KR; ++t) o.b1[t] = rb[colB1[t]];
    };

    // Steps whose four rows all exist (everything but the end of the wave's slice) need no clamp and no dead-row select, and their
    // addresses are a wave-uniform row base (scalar unit) plus a per-lane element offset that never changes: round 5 -- the per-load
    // 64-bit row products, clamps and selects were ~250 vector-ALU instructions (60 of them quarter-rate multiplies) per 240 MFMAs, and
    // a single resident wave does not hide them behind its own matrix instructions.  ALONE the products gain 6-8 % (76 -> 71, 104 -> 96,
    // 85 -> 79, 134 -> 123 us: profiles/r05_wgrad_fast_path_ab.txt); INSIDE the training step, where the products share the chip with
    // update_rows and the backward pass's HBM chain on other queues, the denser matrix-instruction stream makes the step SLOWER
    // (3.13-3.16 vs 3.07-3.10 ms at c4) -- so the path is opt-in (LSTEP_WGRAD_FAST=1) and the default stays the clamped loop.
    unsigned offA4[NG > 0 ? NG : 1], offA1[NR > 0 ? NR : 1], offB4[KG > 0 ? KG : 1], offB1[KR > 0 ? KR : 1];
#pragma unroll
    for (int g = 0; g < NG; ++g) offA4[g] = (unsigned)(kk * p.ldy + colA4[g]);
#pragma unroll
    for (int t = 0; t < NR; ++t) offA1[t] = (unsigned)(kk * p.ldy + colA1[t]);
#pragma unroll
    for (int g = 0; g < KG; ++g) offB4[g] = (unsigned)(kk * p.ldx + colB4[g]);
#pragma unroll
    for (int t = 0; t < KR; ++t) offB1[t] = (unsigned)(kk * p.ldx + colB1[t]);
    const int64_t full_steps = p.no_fast ? 0 : (r_end - r_begin) >> 2;
    auto issue_full = [&](Operands& o, int64_t step) {
        const float* sa = p.dy + (r_begin + 4 * step) * (int64_t)p.ldy;      // wave-uniform
        const float* sb = p.x + (r_begin + 4 * step) * (int64_t)p.ldx;
#pragma unroll
        for (int g = 0; g < NG; ++g) o.a4[g] = *reinterpret_cast<const f32x4*>(sa + offA4[g]);
#pragma unroll
        for (int t = 0; t < NR; ++t) o.a1[t] = sa[offA1[t]];
#pragma unroll
        for (int g = 0; g < KG; ++g) o.b4[g] = *reinterpret_cast<const f32x4*>(sb + offB4[g]);
#pragma unroll
        for (int t = 0; t < KR; ++t) o.b1[t] = sb[offB1[t]];
    };

    f32x4 acc[NT][KTW];
    f32x4 bsum4[NG > 0 ? NG : 1];
    float bsum1[NR > 0 ? NR : 1];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int b = 0; b < KTW; ++b) {
            acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc[a][b]));   // one distinct accumulator tile each
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) bsum4[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NR; ++t) bsum1[t] = 0.f;

    // all products of one A element: tile row `a` against this wave's KTW tiles
    auto row_of_tiles = [&](int a, float va, const Operands& o) {
#pragma unroll
        for (int g = 0; g < KG; ++g) {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[a][4 * g + v] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, o.b4[g][v], acc[a][4 * g + v], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < KR; ++t) acc[a][4 * KG + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, o.b1[t], acc[a][4 * KG + t], 0, 0, 0);
    };

    // Two blocks of DEPTH steps, ping-pong: while block i is multiplied, the loads of block i + 1 are in flight, issued a whole block
    // (DEPTH * NT * KTW MFMAs = 7 000-10 000 cycles) before their first use.  (A ring that issues step s + DEPTH - 1 next to the MFMAs of
    // step s looks deeper but is not: at the loop header hipcc's wait-count pass forgets which loads of the previous trip are still in
    // flight and waits for all of them -- `s_waitcnt vmcnt(<loads of this trip so far>)` -- so the pipeline drains once per trip; the
    // dword kernel above has that shape.  Here "everything issued before this trip" is exactly what the first MFMA of a trip needs.)
    const int64_t steps = (r_end - r_begin + 3) >> 2;
    Operands buf[2][DEPTH];
    auto issue_block = [&](Operands (&blk)[DEPTH], int64_t s) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(blk[d], r_begin + 4 * (s + d));
    };
    auto issue_block_full = [&](Operands (&blk)[DEPTH], int64_t s) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue_full(blk[d], s + d);
    };
    auto multiply_block_full = [&](const Operands (&blk)[DEPTH]) {      // (every row of the block exists)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4 va = blk[d].a4[g];
                bsum4[g] += va;
#pragma unroll
                for (int v = 0; v < 4; ++v) row_of_tiles(4 * g + v, va[v], blk[d]);
            }
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const float va = blk[d].a1[t];
                bsum1[t] += va;
                row_of_tiles(4 * NG + t, va, blk[d]);
            }
        }
    };
    auto multiply_block = [&](const Operands (&blk)[DEPTH], int64_t s) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const bool live = r_begin + 4 * (s + d) + kk < r_end;   // dead rows: zero dY operand, the products vanish
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f32x4 va = blk[d].a4[g];
                if (!live) va = f32x4{0.f, 0.f, 0.f, 0.f};
                bsum4[g] += va;
#pragma unroll
                for (int v = 0; v < 4; ++v) row_of_tiles(4 * g + v, va[v], blk[d]);
            }
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const float va = live ? blk[d].a1[t] : 0.f;
                bsum1[t] += va;
                row_of_tiles(4 * NG + t, va, blk[d]);
            }
        }
    };
    issue_block(buf[0], 0);
    int64_t s = 0;
    // trips whose three blocks -- the two multiplied here and the one requested for the next trip -- hold whole steps only
    for (; s + 3 * DEPTH <= full_steps; s += 2 * DEPTH) {
        issue_block_full(buf[1], s + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        multiply_block_full(buf[0]);
        __builtin_amdgcn_sched_barrier(0);
        issue_block_full(buf[0], s + 2 * DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        multiply_block_full(buf[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; s < steps; s += 2 * DEPTH) {   // the end of the slice (rows past it are clamped loads and zero operands)
        issue_block(buf[1], s + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        multiply_block(buf[0], s);
        __builtin_amdgcn_sched_barrier(0);
        issue_block(buf[0], s + 2 * DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        multiply_block(buf[1], s + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
    }

    // C/D layout of the 16x16 forms: lane l, register q -> tile row 4 * (l >> 4) + q, tile column l & 15.  Tile rows / columns of a
    // 64-column group are the columns base + 4 * index + v: the four tiles v = 0..3 of a k-group hold four consecutive columns per
    // lane, stored as one float4.
    float* out = p.part + (int64_t)slice * p.part_stride;
    auto store_row = [&](int a, int q, int n) {
        if (n >= p.n) return;
        float* dst = out + (int64_t)n * p.k;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int k = k0 + 64 * g + 4 * c;
            if (k < p.k) *reinterpret_cast<f32x4*>(dst + k) = f32x4{acc[a][4 * g][q], acc[a][4 * g + 1][q], acc[a][4 * g + 2][q], acc[a][4 * g + 3][q]};
        }
#pragma unroll
        for (int t = 0; t < KR; ++t) {
            const int k = k0 + 64 * KG + 16 * t + c;
            if (k < p.k) dst[k] = acc[a][4 * KG + t][q];
        }
    };
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int q = 0; q < 4; ++q) store_row(4 * g + v, q, n0 + 64 * g + 4 * (4 * kk + q) + v);
        }
    }
#pragma unroll
    for (int t = 0; t < NR; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) store_row(4 * NG + t, q, n0 + 64 * NG + 16 * t + 4 * kk + q);
    }
    if (kblock == 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 t = bsum4[g];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                t[v] += __shfl_xor(t[v], 16, kWave);
                t[v] += __shfl_xor(t[v], 32, kWave);
            }
            const int n = n0 + 64 * g + 4 * c;
            if (kk == 0 && n < p.n) *reinterpret_cast<f32x4*>(out + p.bias_off + n) = t;
        }
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            float u = bsum1[t];
            u += __shfl_xor(u, 16, kWave);
            u += __shfl_xor(u, 32, kWave);
            const int n = n0 + 64 * NG + 16 * t + c;
            if (kk == 0 && n < p.n) out[p.bias_off + n] = u;
        }
    }
}

template <int NG, int NR, int KG, int KR, int DEPTH>
__global__ __launch_bounds__(kBlock, 1) void wgrad_partial_v2_kernel(const WgradParams p) {
    wgrad_partial_v2_body<NG, NR, KG, KR, DEPTH>(p, (int)blockIdx.x * kWavesPerBlock + wave_in_block(), (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Several products in one launch (the 6 x 4-tile plan only): the waves of all products are numbered in one flat range, product i owns
// [first_wave[i], first_wave[i + 1]).  A training step's six weight gradients are six dependent (partial, reduce) launch pairs otherwise --
// twelve graph nodes of ~4 us dispatch latency each on the critical chain of the small-batch step, and six kernels whose tails cannot
// overlap at the large one.
constexpr int kWgradBatchMax = 8;
struct WgradBatch {
    WgradParams p[kWgradBatchMax];
    int32_t first_wave[kWgradBatchMax + 1];
    int32_t count;
};
__global__ __launch_bounds__(kBlock, 1) void wgrad_partial_v2_batch_kernel(const WgradBatch b) {
    const int gid = (int)blockIdx.x * kWavesPerBlock + wave_in_block();
    int i = 0;
#pragma unroll
    for (int j = 1; j < kWgradBatchMax; ++j)
        if (j < b.count && gid >= b.first_wave[j]) i = j;
    i = __builtin_amdgcn_readfirstlane(i);
    if (gid >= b.first_wave[b.count]) return;
    wgrad_partial_v2_body<1, 2, 1, 0, 5>(b.p[i], gid - b.first_wave[i], 0, 0, 0);
}


// ---- The same product with BOTH operands staged through LDS and shared by the workgroup (round 4).
// The register-only kernels above give every WAVE its own operand stream: a 6 x 4-tile wave reads (96 + 64) columns per row for 24 MFMAs,
// 20 FLOP per byte -- at the matrix cores' 157 TFLOP/s that is 8 TB/s of L2 -> CU traffic, and the six products of a training step re-read
// their operands 4x (2 GB for 32 GFLOP).  They ran at 80 TFLOP/s alone and at 37 inside the step, where update_pe's and the tail's kernels
// pull on the same L2 -> CU path.  Here a workgroup of four waves owns [kWlNT n-tiles] x [4 waves x KTW k-tiles] of the output over its row
// slice; dY[rows, 16 kWlNT columns] and X[rows, 64 KTW columns] are fetched ONCE per workgroup (16-byte loads, 16 rows a stage, double
// buffered through registers into two LDS stages, one barrier per stage), and every wave takes its MFMA operands from LDS: lane
// (kk = l >> 4, c = l & 15) reads A[4 t + kk][16 a + c] and B[4 t + kk][16 (KTW w + b) + c] -- one ds_read_b32 per tile per 4-row step,
// conflict-free because the LDS row strides are = 16 or 48 (mod 64) banks, so the four rows of a step land in four different
// 16-bank groups.  Operand traffic from global memory: (96 + 64 KTW) columns per row per WORKGROUP for 4 x 6 x KTW MFMAs: 47-58 FLOP per
// byte.  <= 200 registers per wave: two workgroups per CU, so a workgroup's barrier waits and LDS fills hide under the other's MFMAs, and
// waves of other kernels still fit beside it.  Partials and the reduction as above (deterministic).
constexpr int kWlRows = 16;     // rows per LDS stage (4 MFMA steps)
constexpr int kWlNT = 6;        // n-tiles per workgroup (all four waves share them)
constexpr int kWlLda = 16 * kWlNT + 16;      // 112 floats: = 48 (mod 64)

template <int KTW>
__global__ __launch_bounds__(kBlock, 2) void wgrad_lds_kernel(const WgradParams p) {
    constexpr int kBCols = 64 * KTW;                 // X columns per workgroup
    constexpr int kLdb = kBCols + 16;                // 208 / 336 floats: = 16 (mod 64)
    constexpr int kA4 = kWlRows * (4 * kWlNT);       // float4 per A stage (384)
    constexpr int kB4 = kWlRows * (kBCols / 4);      // float4 per B stage
    constexpr int kAPer = (kA4 + kBlock - 1) / kBlock, kBPer = kB4 / kBlock;
    static_assert(kB4 % kBlock == 0, "B stage must divide over the workgroup");
    __shared__ __attribute__((aligned(16))) float sA[2][kWlRows * kWlLda];
    __shared__ __attribute__((aligned(16))) float sB[2][kWlRows * kLdb];

    const int tid = threadIdx.x, lane = lane_id(), wave = wave_in_block();
    const int kk = lane >> 4, c = lane & 15;
    // blockIdx.x = (slice * n_blocks + n-block) * k_blocks + k-block
    const int kblock = blockIdx.x % p.k_blocks;
    const int nb_total = p.tasks / p.k_blocks;
    const int nblock = (blockIdx.x / p.k_blocks) % nb_total;
    const int slice = blockIdx.x / p.tasks;
    const int n0 = nblock * (16 * kWlNT);
    const int k0 = kblock * kBCols;                  // first X column of the workgroup
    const int kw0 = k0 + wave * (16 * KTW);          // first X column of this wave's tiles
    const int64_t r_begin = (int64_t)slice * p.rows_per_wg;
    int64_t r_end = r_begin + p.rows_per_wg;
    if (r_end > p.m) r_end = p.m;

    // global -> register staging of one stage: thread -> (row, float4 column); columns past N / K are clamped into the matrix (whole float4s:
    // n and k are multiples of 4) and feed tiles that are never stored; rows past the slice read its last row and are ZEROED in A (so the
    // products vanish) -- B keeps finite values
    f32x4 ga[kAPer], gb[kBPer];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int j = 0; j < kAPer; ++j) {
            const int idx = tid + kBlock * j;
            if (idx < kA4) {
                const int row = idx / (4 * kWlNT), c4 = idx % (4 * kWlNT);
                int64_t r = r0 + row;
                const bool live = r < r_end;
                if (!live) r = r_end - 1;
                const int col = min(n0 + 4 * c4, p.n - 4);
                ga[j] = *reinterpret_cast<const f32x4*>(p.dy + r * p.ldy + col);
                if (!live) ga[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int j = 0; j < kBPer; ++j) {
            const int idx = tid + kBlock * j;
            const int row = idx / (kBCols / 4), c4 = idx % (kBCols / 4);
            int64_t r = r0 + row;
            if (r > r_end - 1) r = r_end - 1;
            const int col = min(k0 + 4 * c4, p.k - 4);
            gb[j] = *reinterpret_cast<const f32x4*>(p.x + r * p.ldx + col);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < kAPer; ++j) {
            const int idx = tid + kBlock * j;
            if (idx < kA4) *reinterpret_cast<f32x4*>(&sA[buf][(idx / (4 * kWlNT)) * kWlLda + 4 * (idx % (4 * kWlNT))]) = ga[j];
        }
#pragma unroll
        for (int j = 0; j < kBPer; ++j) {
            const int idx = tid + kBlock * j;
            *reinterpret_cast<f32x4*>(&sB[buf][(idx / (kBCols / 4)) * kLdb + 4 * (idx % (kBCols / 4))]) = gb[j];
        }
    };

    f32x4 acc[kWlNT][KTW];
    float bsum[kWlNT];
#pragma unroll
    for (int a = 0; a < kWlNT; ++a) {
        bsum[a] = 0.f;
#pragma unroll
        for (int b = 0; b < KTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t stages = (r_end - r_begin + kWlRows - 1) / kWlRows;
    fetch(r_begin);
    stash(0);
    __syncthreads();
    for (int64_t st = 0; st < stages; ++st) {
        const int buf = (int)(st & 1);
        if (st + 1 < stages) fetch(r_begin + (st + 1) * kWlRows);      // in flight under this stage's MFMAs
        const float* a_l = &sA[buf][kk * kWlLda + c];
        const float* b_l = &sB[buf][kk * kLdb + wave * (16 * KTW) + c];
        float va[2][kWlNT], vb[2][KTW];
#pragma unroll
        for (int a = 0; a < kWlNT; ++a) va[0][a] = a_l[16 * a];
#pragma unroll
        for (int b = 0; b < KTW; ++b) vb[0][b] = b_l[16 * b];
#pragma unroll
        for (int t = 0; t < kWlRows / 4; ++t) {
            const int cur = t & 1, nxt = cur ^ 1;
            if (t + 1 < kWlRows / 4) {           // the next step's operands: LDS reads under this step's MFMAs
#pragma unroll
                for (int a = 0; a < kWlNT; ++a) va[nxt][a] = a_l[(4 * (t + 1)) * kWlLda + 16 * a];
#pragma unroll
                for (int b = 0; b < KTW; ++b) vb[nxt][b] = b_l[(4 * (t + 1)) * kLdb + 16 * b];
            }
#pragma unroll
            for (int a = 0; a < kWlNT; ++a) {
                bsum[a] += va[cur][a];
#pragma unroll
                for (int b = 0; b < KTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[cur][a], vb[cur][b], acc[a][b], 0, 0, 0);
            }
        }
        if (st + 1 < stages) stash(buf ^ 1);     // (the other buffer: nobody reads it during this stage)
        __syncthreads();
    }

    // C/D layout of the 16x16 forms: lane l, register q -> tile row 4 * (l >> 4) + q, tile column l & 15
    float* out = p.part + (int64_t)slice * p.part_stride;
#pragma unroll
    for (int a = 0; a < kWlNT; ++a) {
#pragma unroll
        for (int b = 0; b < KTW; ++b) {
            const int k = kw0 + 16 * b + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 16 * a + 4 * kk + q;
                if (n < p.n && k < p.k && 16 * (KTW * wave + b) < kBCols) out[(int64_t)n * p.k + k] = acc[a][b][q];
            }
        }
    }
    if (kblock == 0 && wave == 0) {
#pragma unroll
        for (int a = 0; a < kWlNT; ++a) {
            float t = bsum[a];
            t += __shfl_xor(t, 16, kWave);
            t += __shfl_xor(t, 32, kWave);
            const int n = n0 + 16 * a + c;
            if (kk == 0 && n < p.n) out[p.bias_off + n] = t;
        }
    }
}

// out = sum over the S partial blocks: 16 float4 columns x 16 partial groups per workgroup (every thread has S / 16 independent
// 16-byte loads in flight; the first version gave each thread S / 4 and launched a quarter of the workgroups, 33 us per call for 48 MB
// that sit in the Infinity Cache), groups added in a fixed order: deterministic
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, int64_t part_stride, int32_t num_part, int32_t n, int32_t k,
                                                  int32_t bias_off, float* __restrict__ dw, int32_t ld_dw, float* __restrict__ db, int64_t block) {
    __shared__ float4 sh[kBlock];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t e = (block * 16 + col) * 4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < part_stride) {
        const float* src = part + e;
#pragma unroll 8
        for (int s = grp; s < num_part; s += 16) {
            const float4 v = ld4_stream(src + (int64_t)s * part_stride);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
    }
    sh[threadIdx.x] = sum;
    __syncthreads();
    if (grp != 0 || e >= part_stride) return;
#pragma unroll
    for (int g = 1; g < 16; ++g) {
        const float4 v = sh[g * 16 + col];
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

__global__ __launch_bounds__(kBlock) void wgrad_reduce_kernel(const float* __restrict__ part, int64_t part_stride, int32_t num_part, int32_t n,
                                                              int32_t k, int32_t bias_off, float* __restrict__ dw, int32_t ld_dw,
                                                              float* __restrict__ db) {
    wgrad_reduce_body(part, part_stride, num_part, n, k, bias_off, dw, ld_dw, db, (int64_t)blockIdx.x);
}

// the reductions of a batched launch (lstep_linear_wgrad_batch): product i owns the workgroups [first_block[i], first_block[i + 1])
constexpr int kWgradReduceBatchMax = 8;
struct WgradReduceItem {
    const float* part;
    float* dw;
    float* db;
    int64_t part_stride;
    int32_t num_part, n, k, bias_off, ld_dw;
};
struct WgradReduceBatch {
    WgradReduceItem it[kWgradReduceBatchMax];
    int32_t first_block[kWgradReduceBatchMax + 1];
    int32_t count;
};
__global__ __launch_bounds__(kBlock) void wgrad_reduce_batch_kernel(const WgradReduceBatch b) {
    const int blk = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int j = 1; j < kWgradReduceBatchMax; ++j)
        if (j < b.count && blk >= b.first_block[j]) i = j;
    const WgradReduceItem& r = b.it[i];
    wgrad_reduce_body(r.part, r.part_stride, r.num_part, r.n, r.k, r.bias_off, r.dw, r.ld_dw, r.db, (int64_t)(blk - b.first_block[i]));
}

static inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

struct WgradPlan {
    int nt, ktw;          // template instance
    int lds;              // 1: wgrad_lds_kernel<ktw> (operands staged through LDS, shared by the workgroup)
    int small;            // 1: 6 x 4 tiles per wave (<= 256 registers: leaves half of every SIMD's register file to other kernels)
    int k_blocks, tasks;  // small: flattened (slice, n-block, k-block) wave numbering
    int gy, gz, splits;   // grid
    int64_t rows_per_wg, part_stride;
    int32_t bias_off;
};

// Two tilings.  "big": 9-11 x 3-5 tiles per wave (up to 220 accumulator registers, 512 registers per lane, one wave per SIMD): the fewest
// partial blocks and operand loads, but its waves own the whole register file for the whole kernel -- NOTHING else runs on the chip
// meanwhile (tools/overlap_probe.py: 6 products beside a 2 GiB copy take 1.07 ms, more than the 0.93 ms of one after the other).
// "small": 6 x 4 tiles per wave (96 accumulators, <= 256 registers): a wave of another kernel fits beside it on every SIMD, so the
// HBM-bound backward chain of the training step (gather backward, gradient sort, segment sums, history filter backward) runs
// UNDERNEATH the weight gradients instead of around them; 2.5x fewer partial blocks on top.  Default for operands that allow 16-byte
// loads; LSTEP_WGRAD_BIG=1 keeps the big tiling (A/B).
static WgradPlan wgrad_plan(int64_t m, int32_t n, int32_t k, bool allow_small = true) {
    WgradPlan pl;
    const int ntiles = (n + 15) / 16, ktiles = (k + 15) / 16;
    static const bool force_big = getenv("LSTEP_WGRAD_BIG") != nullptr;
    // LSTEP_WGRAD_LDS=1 selects the LDS-staged kernel (round 4: measured and NOT kept as the default -- 59-79 TFLOP/s alone against 61-80 for the
    // register-only 6 x 4 tiling, 3.18 vs 3.17 ms per c4 step: operand delivery from L2 is not what limits these products)
    static const bool no_lds = getenv("LSTEP_WGRAD_LDS") == nullptr;
    pl.small = pl.lds = 0;
    pl.k_blocks = pl.tasks = 0;
    if (allow_small && !force_big && !no_lds && n % 4 == 0 && k % 4 == 0 && n >= 4 && k >= 4) {
        // k-tiles are dealt out evenly: workgroups of 4 waves x KTW tiles, KTW = 3 or 5 whichever wastes fewer tile slots
        pl.lds = 1;
        pl.nt = kWlNT;
        const int slots3 = (ktiles + 11) / 12 * 12, slots5 = (ktiles + 19) / 20 * 20;
        pl.ktw = (slots3 <= slots5) ? 3 : 5;
        const int n_blocks = (ntiles + kWlNT - 1) / kWlNT;
        pl.k_blocks = (ktiles + 4 * pl.ktw - 1) / (4 * pl.ktw);
        pl.tasks = n_blocks * pl.k_blocks;
        pl.gz = pl.gy = 1;
        const char* wgs_env = getenv("LSTEP_WGRAD_WGS");       // tuning: workgroups per launch (default 512: two per CU)
        const int wgs = (wgs_env && atoi(wgs_env) > 0) ? atoi(wgs_env) : 512;
        int splits = wgs / pl.tasks;
        if (splits < 1) splits = 1;
        pl.rows_per_wg = round_up((m + splits - 1) / splits, kWlRows);
        if (pl.rows_per_wg < kWlRows) pl.rows_per_wg = kWlRows;
        pl.splits = (int)((m + pl.rows_per_wg - 1) / pl.rows_per_wg);
        if (pl.splits < 1) pl.splits = 1;
        pl.bias_off = (int32_t)round_up((int64_t)n * k, 4);
        pl.part_stride = pl.bias_off + round_up(n, 4);
        return pl;
    }
    if (allow_small && !force_big && n % 4 == 0 && k % 4 == 0 && n >= 4 && k >= 4) {
        pl.small = 1;
        pl.nt = 6;
        pl.ktw = 4;
        const int n_blocks = (ntiles + 5) / 6;
        pl.k_blocks = (ktiles + 3) / 4;
        const int tasks = n_blocks * pl.k_blocks;
        pl.tasks = tasks;
        pl.gz = pl.gy = 1;
        static const int target_waves = [] {       // tuning knob (tools/wgrad_bench.py): waves per launch, default one per SIMD
            const char* e = getenv("LSTEP_WGRAD_WAVES");
            const int v = e ? atoi(e) : 0;
            return v >= 256 && v <= 8192 ? v : 1024;
        }();
        int splits = target_waves / tasks;       // one wave per SIMD, no idle slot: waves are numbered over (slice, task)
        if (splits < 1) splits = 1;
        pl.rows_per_wg = round_up((m + splits - 1) / splits, 40);   // whole ping-pong rounds (2 blocks x 5 steps x 4 rows)
        if (pl.rows_per_wg < 40) pl.rows_per_wg = 40;
        pl.splits = (int)((m + pl.rows_per_wg - 1) / pl.rows_per_wg);
        if (pl.splits < 1) pl.splits = 1;
        pl.bias_off = (int32_t)round_up((int64_t)n * k, 4);
        pl.part_stride = pl.bias_off + round_up(n, 4);
        return pl;
    }
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
    const WgradPlan a = wgrad_plan(m, n, k, true), b = wgrad_plan(m, n, k, false);     // (the operands' alignment decides at launch time)
    const int64_t need_a = (int64_t)a.splits * a.part_stride, need_b = (int64_t)b.splits * b.part_stride;
    return (need_a > need_b ? need_a : need_b) * (int64_t)sizeof(float);
}

// the split-M partial products of one weight gradient into `workspace` (no reduction); the plan it ran with comes back in *plan
static int wgrad_launch_partial(const float* dy, int32_t ldy, const float* x, int32_t ldx, int64_t m, int32_t n, int32_t k, int32_t ld_dw,
                                void* workspace, int64_t workspace_bytes, void* stream, WgradPlan* plan) {
    if (m < 0 || n <= 0 || k <= 0 || ldy < n || ldx < k || ld_dw < k) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: bad sizes");
    if ((int64_t)n * k > ((int64_t)1 << 30)) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: weight matrix too large");
    if (!workspace || (m > 0 && (!dy || !x))) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: NULL pointer");
    if (workspace_bytes < lstep_linear_wgrad_workspace(m, n, k)) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: workspace too small");
    if (((uintptr_t)workspace & 15) != 0) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // 16-byte operand loads need 16-byte aligned rows and whole float4s inside the matrices; anything else takes the dword kernel
    static const bool no_wide = getenv("LSTEP_WGRAD_DWORD") != nullptr;      // A/B switch (tools/wgrad_bench.py)
    const bool wide = !no_wide && n % 4 == 0 && k % 4 == 0 && n >= 4 && k >= 4 && ldy % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)dy & 15) == 0 &&
                      ((uintptr_t)x & 15) == 0;
    const WgradPlan pl = wgrad_plan(m, n, k, wide);
    WgradParams p;
    p.dy = dy; p.x = x; p.part = (float*)workspace;
    p.m = m; p.rows_per_wg = pl.rows_per_wg; p.part_stride = pl.part_stride;
    p.ldy = ldy; p.ldx = ldx; p.n = n; p.k = k; p.bias_off = pl.bias_off; p.k_blocks = pl.k_blocks;
    p.tasks = (pl.small || pl.lds) ? pl.tasks : 0; p.slices = pl.splits; p.no_fast = wgrad_no_fast();
    const dim3 grid((unsigned)pl.splits, (unsigned)pl.gy, (unsigned)pl.gz), block(kBlock);
    if (pl.lds) {
        const dim3 lgrid((unsigned)((int64_t)pl.splits * pl.tasks));
        if (pl.ktw == 3) hipLaunchKernelGGL((wgrad_lds_kernel<3>), lgrid, block, 0, s, p);
        else hipLaunchKernelGGL((wgrad_lds_kernel<5>), lgrid, block, 0, s, p);
    } else if (pl.small) {
        const dim3 flat((unsigned)(((int64_t)pl.splits * pl.tasks + kWavesPerBlock - 1) / kWavesPerBlock));
        hipLaunchKernelGGL((wgrad_partial_v2_kernel<1, 2, 1, 0, 5>), flat, block, 0, s, p);
    } else if (wide) {
        if (pl.nt == 11 && pl.ktw == 5) hipLaunchKernelGGL((wgrad_partial_v2_kernel<2, 3, 1, 1, 4>), grid, block, 0, s, p);
        else if (pl.nt == 11) hipLaunchKernelGGL((wgrad_partial_v2_kernel<2, 3, 0, 3, 4>), grid, block, 0, s, p);
        else if (pl.ktw == 5) hipLaunchKernelGGL((wgrad_partial_v2_kernel<2, 1, 1, 1, 4>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((wgrad_partial_v2_kernel<2, 1, 0, 3, 4>), grid, block, 0, s, p);
    } else if (pl.nt == 11 && pl.ktw == 5) hipLaunchKernelGGL((wgrad_partial_kernel<11, 5, 4>), grid, block, 0, s, p);
    else if (pl.nt == 11) hipLaunchKernelGGL((wgrad_partial_kernel<11, 3, 4>), grid, block, 0, s, p);
    else if (pl.ktw == 5) hipLaunchKernelGGL((wgrad_partial_kernel<9, 5, 4>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((wgrad_partial_kernel<9, 3, 4>), grid, block, 0, s, p);
    *plan = pl;
    return LSTEP_OK;
}

extern "C" int lstep_linear_wgrad(const float* dy, int32_t ldy, const float* x, int32_t ldx, int64_t m, int32_t n, int32_t k, float* dw,
                                  int32_t ld_dw, float* db, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!dw) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad: NULL pointer");
    WgradPlan pl;
    if (int rc = wgrad_launch_partial(dy, ldy, x, ldx, m, n, k, ld_dw, workspace, workspace_bytes, stream, &pl)) return rc;
    const unsigned rgrid = (unsigned)((pl.part_stride / 4 + 15) / 16);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(kBlock), 0, (hipStream_t)stream, (const float*)workspace, pl.part_stride,
                       (int32_t)pl.splits, n, k, pl.bias_off, dw, ld_dw, db);
    return check_launch("lstep_linear_wgrad");
}

static inline int64_t align256f(int64_t bytes) { return (bytes + 255) / 256 * 256; }

extern "C" int64_t lstep_linear_wgrad_batch_workspace(int32_t count, const lstep_wgrad_desc_t* descs) {
    if (count <= 0 || !descs) return 0;
    int64_t total = 0;
    for (int i = 0; i < count; ++i) total += align256f(lstep_linear_wgrad_workspace(descs[i].m, descs[i].n, descs[i].k));
    return total;
}

extern "C" int lstep_linear_wgrad_batch(int32_t count, const lstep_wgrad_desc_t* descs, void* workspace, int64_t workspace_bytes, void* stream) {
    if (count < 0 || count > kWgradBatchMax) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad_batch: 0 .. %d products per call", kWgradBatchMax);
    if (count == 0) return LSTEP_OK;
    if (!descs || !workspace) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad_batch: NULL pointer");
    if (((uintptr_t)workspace & 15) != 0) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad_batch: workspace must be 16-byte aligned");
    if (workspace_bytes < lstep_linear_wgrad_batch_workspace(count, descs)) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad_batch: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    static const bool no_wide = getenv("LSTEP_WGRAD_DWORD") != nullptr;
    static const bool no_batch = getenv("LSTEP_WGRAD_NO_BATCH") != nullptr;      // A/B switch: one (partial, reduce) launch pair per product
    static const bool reduce_each = no_batch || getenv("LSTEP_WGRAD_REDUCE_EACH") != nullptr;
    // Products of more than this many rows keep their own launch pairs: one launch of all six 49 152-row products puts 6 144 waves of
    // 240 registers on the chip at once, two per SIMD, for ~0.5 ms -- alone that is 5 % faster than six launches, but under it the step's
    // other kernels (update_rows, gather backward) run at half speed and the c4 step does not gain (3.18 vs 3.15 ms; under the kernel
    // tracer 7.5 vs 3.7 ms per iteration).  Small batches are launch-latency-bound: there the single pair is 2x faster.
    static const int64_t batch_rows = [] {
        const char* e = getenv("LSTEP_WGRAD_BATCH_ROWS");
        const long long v = e ? atoll(e) : 0;
        return (int64_t)(v > 0 ? v : 16384);
    }();
    WgradBatch pb;
    WgradReduceBatch rb;
    pb.count = rb.count = 0;
    pb.first_wave[0] = rb.first_block[0] = 0;
    char* ws = (char*)workspace;
    for (int i = 0; i < count; ++i) {
        const lstep_wgrad_desc_t& d = descs[i];
        const int64_t bytes = align256f(lstep_linear_wgrad_workspace(d.m, d.n, d.k));
        if (d.m < 0 || d.n <= 0 || d.k <= 0 || d.ldy < d.n || d.ldx < d.k || d.ld_dw < d.k) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad_batch: bad sizes in product %d", i);
        if (!d.dw || (d.m > 0 && (!d.dy || !d.x))) return set_error(LSTEP_EINVAL, "lstep_linear_wgrad_batch: NULL pointer in product %d", i);
        const bool wide = !no_wide && d.n % 4 == 0 && d.k % 4 == 0 && d.n >= 4 && d.k >= 4 && d.ldy % 4 == 0 && d.ldx % 4 == 0 &&
                          ((uintptr_t)d.dy & 15) == 0 && ((uintptr_t)d.x & 15) == 0;
        const WgradPlan pl = wgrad_plan(d.m, d.n, d.k, wide);
        if (no_batch || !pl.small || d.m == 0 || d.m > batch_rows) {      // plans the batched kernel does not cover: their own partial launches, same stream, same result
            // round 5: their REDUCTIONS wait for the end of the call and go out as one launch (each product has its own slice of the
            // workspace): on the auxiliary stream the five 49 152-row products of a c4 step ran partial -> reduce -> partial -> ..., and that
            // chain is the longest dependent chain of the whole step (tools/graph_critical_path.py) -- 5 x 34 us of reductions that nothing
            // in front of the optimiser needs early.  LSTEP_WGRAD_REDUCE_EACH=1: a reduction behind every product, as before.
            if (reduce_each || rb.count >= kWgradReduceBatchMax) {
                if (int rc = lstep_linear_wgrad(d.dy, d.ldy, d.x, d.ldx, d.m, d.n, d.k, d.dw, d.ld_dw, d.db, ws, bytes, stream)) return rc;
            } else {
                WgradPlan ran;
                if (int rc = wgrad_launch_partial(d.dy, d.ldy, d.x, d.ldx, d.m, d.n, d.k, d.ld_dw, ws, bytes, stream, &ran)) return rc;
                WgradReduceItem& r = rb.it[rb.count];
                r.part = (const float*)ws; r.dw = d.dw; r.db = d.db; r.part_stride = ran.part_stride;
                r.num_part = ran.splits; r.n = d.n; r.k = d.k; r.bias_off = ran.bias_off; r.ld_dw = d.ld_dw;
                rb.first_block[rb.count + 1] = rb.first_block[rb.count] + (int32_t)((ran.part_stride / 4 + 15) / 16);
                ++rb.count;
            }
        } else {
            WgradParams& p = pb.p[pb.count];
            p.dy = d.dy; p.x = d.x; p.part = (float*)ws;
            p.m = d.m; p.rows_per_wg = pl.rows_per_wg; p.part_stride = pl.part_stride;
            p.ldy = d.ldy; p.ldx = d.ldx; p.n = d.n; p.k = d.k; p.bias_off = pl.bias_off; p.k_blocks = pl.k_blocks;
            p.tasks = pl.tasks; p.slices = pl.splits; p.no_fast = wgrad_no_fast();
            pb.first_wave[pb.count + 1] = pb.first_wave[pb.count] + pl.splits * pl.tasks;
            ++pb.count;
            WgradReduceItem& r = rb.it[rb.count];
            r.part = (const float*)ws; r.dw = d.dw; r.db = d.db; r.part_stride = pl.part_stride;
            r.num_part = pl.splits; r.n = d.n; r.k = d.k; r.bias_off = pl.bias_off; r.ld_dw = d.ld_dw;
            rb.first_block[rb.count + 1] = rb.first_block[rb.count] + (int32_t)((pl.part_stride / 4 + 15) / 16);
            ++rb.count;
        }
        ws += bytes;
    }
    if (pb.count > 0) {
        const unsigned waves = (unsigned)pb.first_wave[pb.count];
        hipLaunchKernelGGL(wgrad_partial_v2_batch_kernel, dim3((waves + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, pb);
    }
    if (rb.count > 0) hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)rb.first_block[rb.count]), dim3(kBlock), 0, s, rb);
    return check_launch("lstep_linear_wgrad_batch");
}

// ---- small dense products between weight-sized matrices (<= a few hundred rows / columns): the weight composition of the dense
// tail and its backward (model._TailWeights).  The library runs these 8 - 25 MFLOP products as ONE workgroup (53 - 60 us each); here
// every 16 x 16 output tile is its own workgroup on the fp32 matrix cores.  General element strides, so transposes are free.
namespace lstep {

struct SmallGemmParams {
    const float *a, *b;
    float* c;
    int32_t m, n, k;
    int64_t sa_i, sa_k, sb_k, sb_j, sc_i, sc_j;
    float alpha, beta;
};

// One WORKGROUP per 16 x 16 output tile: its four waves split the contraction (whole 4-wide MFMA steps), each issues all its operand
// loads at once and runs two independent accumulator chains (the 16x16x4 form has a 40-cycle dependent latency), the four partial
// tiles meet in LDS and are added in a fixed order.  (One WAVE per tile ran the 43-68 steps of these products as one dependent chain
// behind 6-9 rounds of exposed load latency: ~30 us per product, 0.27 ms per training iteration on the auxiliary stream.)
__global__ __launch_bounds__(kBlock) void small_gemm_kernel(const SmallGemmParams p) {
    __shared__ float red[kWavesPerBlock][4][kWave];
    const int lane = lane_id(), wave = wave_in_block();
    const int tiles_n = (p.n + 15) / 16;
    const int tile = blockIdx.x;
    const int i0 = (tile / tiles_n) * 16, j0 = (tile % tiles_n) * 16;
    const int r = lane & 15, q = lane >> 4;
    const int ai = min(i0 + r, p.m - 1), bj = min(j0 + r, p.n - 1);   // clamped: out-of-range rows / columns are computed and dropped
    const int steps_total = (p.k + 3) / 4;
    const int per = (steps_total + kWavesPerBlock - 1) / kWavesPerBlock;
    const int s_begin = wave * per;
    const int s_end = min(s_begin + per, steps_total);
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int kSteps = 24;    // one round covers contractions up to 4 * 4 * 24 = 384
    const float* pa = p.a + ai * p.sa_i;
    const float* pb = p.b + bj * p.sb_j;
    for (int s0 = s_begin; s0 < s_end; s0 += kSteps) {
        float av[kSteps], bv[kSteps];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const int kk = 4 * (s0 + u) + q;
            const bool ok = s0 + u < s_end && kk < p.k;
            const int kc = ok ? kk : 0;
            av[u] = pa[kc * p.sa_k];
            bv[u] = pb[kc * p.sb_k];
            if (!ok) av[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < kSteps; u += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) red[wave][v][lane] = acc0[v] + acc1[v];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float sum = ((red[0][v][lane] + red[1][v][lane]) + red[2][v][lane]) + red[3][v][lane];
        const int i = i0 + 4 * q + v, j = j0 + r;
        if (i < p.m && j < p.n) {
            float* dst = p.c + i * p.sc_i + j * p.sc_j;
            *dst = p.beta == 0.f ? p.alpha * sum : p.alpha * sum + p.beta * *dst;
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
    hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, (hipStream_t)stream, p);
    return check_launch("lstep_small_gemm");
}
