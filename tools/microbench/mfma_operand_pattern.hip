// fp32 MFMA issue rate with the dense kernels' operand pattern: 11 x 3 accumulators, A changes every 3 MFMAs, B cycles over 3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int UNROLL = 1>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, int iters) {
    f32x4 acc[11][3];
    for (int t = 0; t < 11; ++t) for (int s = 0; s < 3; ++s) acc[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[11], b[3];
    for (int t = 0; t < 11; ++t) a[t] = *reinterpret_cast<const f32x4*>(in + 4 * (threadIdx.x + 256 * t));
    for (int s = 0; s < 3; ++s) b[s] = *reinterpret_cast<const f32x4*>(in + 4 * (threadIdx.x + 256 * (11 + s)));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int t = 0; t < 11; ++t) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (MODE == 0) acc[t][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][v], b[s][v], acc[t][s], 0, 0, 0);
                    if (MODE == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[t][s]) : "v"(a[t][v]), "a"(b[s][v]));
                    if (MODE == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[t][s]) : "v"(a[t][v]), "v"(b[s][v]));
                    if (MODE == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[t][s]) : "v"(a[t][v]), "v"(b[s][v]));
                }
            }
        }
    }
    f32x4 sum = acc[0][0];
    for (int t = 0; t < 11; ++t) for (int s = 0; s < 3; ++s) sum += acc[t][s];
    if (sum[0] == 123.456f) out[0] = sum[1] + sum[2] + sum[3];
}
template <int MODE, int UNROLL = 1>
void run(int blocks, int iters, const float* in, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * 4 * iters * 132 * 2048.0 * UNROLL; printf("unroll %d ", UNROLL);
    printf("mode %d blocks %5d iters %d: %.3f ms  %.1f TFLOP/s\n", MODE, blocks, iters, ms, flops / ms * 1e-9);
}
int main(int argc, char** argv) { bool zero = argc > 1;
    float *in, *out; (void)hipMalloc(&in, 4 * 4 * 256 * 14); { float* h = (float*)malloc(4 * 4 * 256 * 14); for (int i = 0; i < 4 * 256 * 14; ++i) h[i] = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f; (void)hipMemcpy(in, h, 4 * 4 * 256 * 14, hipMemcpyHostToDevice); } (void)hipMalloc(&out, 4);
    run<0>(256, 1000, in, out); run<0, 4>(256, 250, in, out); run<0, 18>(256, 50, in, out); run<0, 18>(1511, 1, in, out); run<0, 4>(1511, 4, in, out); run<0, 1>(1511, 18, in, out);
    return 0;
}
