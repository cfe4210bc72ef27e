// fp32 MFMA rate of one wave per SIMD while the same wave streams rows from memory (loads consumed one iteration later)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NL, int GATHER = 0, int SS = 3>   // loads (1 KB per wave each) per 132 MFMAs
__global__ __launch_bounds__(256, 1) void k(const float* in, size_t span_f4, float* out, int iters) {
    f32x4 acc[11][SS];
    for (int t = 0; t < 11; ++t) for (int s = 0; s < SS; ++s) acc[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    f32x4 a[11], b[SS], buf[2][NL > 0 ? NL : 1];
    for (int t = 0; t < 11; ++t) a[t] = f32x4{1.f, 2.f, 3.f, 4.f};
    for (int s = 0; s < SS; ++s) b[s] = f32x4{1.f, 0.5f, 0.25f, 2.f};
    const f32x4* src = reinterpret_cast<const f32x4*>(in);
    size_t pos = (wave * 7919u * 64u) & (span_f4 - 1);
    for (int u = 0; u < NL; ++u) buf[0][u] = src[(pos + (size_t)u * 64 * 1031 + lane) & (span_f4 - 1)];
    auto issue = [&](f32x4 (&dst)[NL > 0 ? NL : 1]) {
        pos = (pos + 64u * 104729u) & (span_f4 - 1);
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            if (GATHER) dst[u] = src[(pos + (size_t)(lane & 15) * 69 * 997 + (lane >> 4) + 4 * u) & (span_f4 - 1)];   // 16 rows far apart, 64 B of each, next 64 B per u
            else dst[u] = src[(pos + (size_t)u * 64 * 1031 + lane) & (span_f4 - 1)];
        }
    };
    auto mm = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int t = 0; t < 11; ++t) {
#pragma unroll
                for (int s = 0; s < SS; ++s) acc[t][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][v], b[s][v], acc[t][s], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto use = [&](f32x4 (&srcb)[NL > 0 ? NL : 1]) {
#pragma unroll
        for (int u = 0; u < NL; ++u) a[u % 11] += srcb[u];
    };
    for (int it = 0; it < iters; it += 2) {
        issue(buf[1]); mm(); use(buf[0]);
        issue(buf[0]); mm(); use(buf[1]);
    }
    f32x4 sum = acc[0][0];
    for (int t = 0; t < 11; ++t) for (int s = 0; s < SS; ++s) sum += acc[t][s];
    if (sum[0] == 123.456f) out[0] = sum[1] + sum[2] + sum[3];
}
template <int NL, int GATHER = 0, int SS = 3>
void run(int blocks, int iters, const float* in, size_t span_f4, float* out, const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NL, GATHER, SS>), dim3(blocks), dim3(256), 0, 0, in, span_f4, out, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NL, GATHER, SS>), dim3(blocks), dim3(256), 0, 0, in, span_f4, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double flops = (double)blocks * 4 * iters * 44 * SS * 2048.0; printf("S=%d ", SS);
    double bytes = (double)blocks * 4 * iters * NL * 1024.0;
    printf("%s loads/132mfma %2d: %.3f ms  %.1f TFLOP/s  %.2f TB/s\n", what, NL, ms, flops / ms * 1e-9, bytes / ms * 1e-9);
}
int main() {
    size_t big = (size_t)1 << 30, small = (size_t)1 << 20;   // bytes
    float *in, *out; (void)hipMalloc(&in, big); (void)hipMemset(in, 0, big); (void)hipMalloc(&out, 4);
    run<0>(1024, 2000, in, small / 16, out, "none ");
    run<14, 1, 3>(1024, 2000, in, small / 16, out, "L2 g "); run<14, 1, 2>(1024, 2000, in, small / 16, out, "L2 g "); run<14, 1, 1>(1024, 2000, in, small / 16, out, "L2 g ");
    run<14, 1, 3>(1024, 2000, in, big / 16, out, "HBM g"); run<14, 1, 2>(1024, 2000, in, big / 16, out, "HBM g"); run<14, 1, 1>(1024, 2000, in, big / 16, out, "HBM g");
    run<20, 1, 2>(1024, 2000, in, small / 16, out, "L2 g "); run<20, 1, 2>(1024, 2000, in, big / 16, out, "HBM g");
    return 0;
}
