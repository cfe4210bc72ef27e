// Accuracy of the hardware cosine (v_cos_f32: cos(2*pi*x) for x in revolutions) behind a float64 range reduction, against the
// cos_full_range of csrc/lstep_common.h and against cos() in double, over the argument range of the time encoder (|x| up to 2e9).
//   hipcc -O3 --offload-arch=gfx950 -I../../include -I../../l-step_amd/csrc vcos_accuracy.hip -o /tmp/vcos && /tmp/vcos
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "lstep_common.h"

__device__ __forceinline__ float cos_hw(float x) {
    const double rev = (double)x * 0.15915494309189533577;      // x / (2 pi), exact enough in float64 up to 2e9 (54 - 31 bits left)
    const float f = (float)(rev - __builtin_rint(rev));          // [-0.5, 0.5]
    return __builtin_amdgcn_cosf(f);
}

__global__ void eval(const float* x, float* hw, float* exact32, double* ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hw[i] = cos_hw(x[i]);
    exact32[i] = lstep::cos_full_range(x[i]);
    ref[i] = cos((double)x[i]);
}

int main() {
    const int n = 1 << 22;
    float* hx = (float*)malloc(n * sizeof(float));
    srand(1);
    for (int i = 0; i < n; ++i) {
        const double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
        const double mag = pow(10.0, -6.0 + 15.3 * u);            // 1e-6 .. 2e9, log-uniform
        hx[i] = (float)((v < 0.5 ? -1.0 : 1.0) * mag);
    }
    float *dx, *dhw, *dex;
    double* dref;
    hipMalloc(&dx, n * 4); hipMalloc(&dhw, n * 4); hipMalloc(&dex, n * 4); hipMalloc(&dref, n * 8);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(eval, dim3(n / 256), dim3(256), 0, 0, dx, dhw, dex, dref, n);
    float* hhw = (float*)malloc(n * 4); float* hex = (float*)malloc(n * 4); double* href = (double*)malloc(n * 8);
    hipMemcpy(hhw, dhw, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hex, dex, n * 4, hipMemcpyDeviceToHost); hipMemcpy(href, dref, n * 8, hipMemcpyDeviceToHost);
    double worst_hw = 0, worst_ex = 0, sum_hw = 0, sum_ex = 0;
    float at_hw = 0;
    for (int i = 0; i < n; ++i) {
        const double e1 = fabs((double)hhw[i] - href[i]), e2 = fabs((double)hex[i] - href[i]);
        if (e1 > worst_hw) { worst_hw = e1; at_hw = hx[i]; }
        if (e2 > worst_ex) worst_ex = e2;
        sum_hw += e1 * e1; sum_ex += e2 * e2;
    }
    printf("%d arguments, |x| in [1e-6, 2e9]\n", n);
    printf("v_cos_f32 behind a float64 reduction: max |err| %.3e (at x = %.9g), rms %.3e\n", worst_hw, at_hw, sqrt(sum_hw / n));
    printf("cos_full_range (shipped):             max |err| %.3e, rms %.3e\n", worst_ex, sqrt(sum_ex / n));
    return 0;
}
