// F, parameter side: the real [T, P] coefficient table of the FFT filter and its gradient (models/LSTEP.py:104-137 is linear in the
// history):
//   A[f] = sum_t e^{+2 pi i f t / T} a[t] m[t]        c[f] = m[f] A[f] / T        coef[s, p] = Re( sum_f e^{-2 pi i f s / T} W[f, p] c[f] )
// with W = fft_filter.weight (complex64 [T, P]), a = fft_agg.weight (float32 [T]), m the 0/1 mask of the not-yet-full history window.
// Everything in complex128, like the framework formulation it replaces (~15 launches each way): one kernel forward, two backward.
#include "lstep_common.h"

namespace lstep {

constexpr int kMaxFftT = 256;

struct cd {
    double re, im;
};
__device__ __forceinline__ cd cmul(cd a, cd b) { return cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// twiddle[k] = e^{+2 pi i k / T}, k in [0, T)
__device__ __forceinline__ void fill_twiddle(cd* tw, int T) {
    for (int k = threadIdx.x; k < T; k += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)k / (double)T, &s, &c);
        tw[k] = cd{c, s};
    }
}

// block s: coef[s, :].  Every block recomputes c[f] (T^2 complex MACs); block 0 stores it for the backward pass.
__global__ __launch_bounds__(kBlock) void fft_coef_fwd_kernel(const float* __restrict__ w, const float* __restrict__ a, const double* __restrict__ m,
                                                              int T, int P, float* __restrict__ coef, double* __restrict__ c_out) {
    __shared__ cd tw[kMaxFftT], c[kMaxFftT];
    __shared__ double am[kMaxFftT];
    const int s = blockIdx.x;
    fill_twiddle(tw, T);
    for (int t = threadIdx.x; t < T; t += blockDim.x) am[t] = (double)a[t] * m[t];
    __syncthreads();
    for (int f = threadIdx.x; f < T; f += blockDim.x) {
        cd acc{0.0, 0.0};
        for (int t = 0, k = 0; t < T; ++t) {          // k = (f * t) mod T, stepped
            const cd e = tw[k];
            acc.re += e.re * am[t];
            acc.im += e.im * am[t];
            k += f;
            if (k >= T) k -= T;
        }
        const double sc = m[f] / (double)T;
        c[f] = cd{acc.re * sc, acc.im * sc};
        if (s == 0) { c_out[2 * f] = c[f].re; c_out[2 * f + 1] = c[f].im; }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        double acc = 0.0;
        // (unrolled: the filter weights come from global memory inside this loop -- one dependent-latency load per step otherwise; these
        // three parameter-side kernels sit on the critical chain of EVERY configuration: 14.6 + 11 + 12 us at T = 100 before, round 5)
#pragma unroll 10
        for (int f = 0, k = 0; f < T; ++f) {           // k = (f * s) mod T
            const float2 wv = *reinterpret_cast<const float2*>(w + ((size_t)f * P + p) * 2);
            const cd q = cmul(cd{(double)wv.x, (double)wv.y}, c[f]);
            const cd e = tw[k];                                    // e^{-i theta} = conj
            acc += e.re * q.re + e.im * q.im;                      // Re(conj(e) q)
            k += s;
            if (k >= T) k -= T;
        }
        coef[(size_t)s * P + p] = (float)acc;
    }
}

// block f: gQ[f, p] = sum_s e^{+i theta_fs} g[s, p];  g_w[f, p] = conj(c[f]) gQ;  g_c[f] = sum_p conj(W[f, p]) gQ[f, p]
__global__ __launch_bounds__(kBlock) void fft_coef_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w, const double* __restrict__ c,
                                                              int T, int P, float* __restrict__ g_w, double* __restrict__ g_c) {
    __shared__ cd tw[kMaxFftT];
    __shared__ double red[2][kBlock];
    const int f = blockIdx.x;
    fill_twiddle(tw, T);
    __syncthreads();
    const cd cc{c[2 * f], -c[2 * f + 1]};
    double sr = 0.0, si = 0.0;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        cd gq{0.0, 0.0};
#pragma unroll 10
        for (int s = 0, k = 0; s < T; ++s) {           // k = (f * s) mod T
            const cd e = tw[k];
            const double gv = (double)g[(size_t)s * P + p];
            gq.re += e.re * gv;
            gq.im += e.im * gv;
            k += f;
            if (k >= T) k -= T;
        }
        const cd gw = cmul(gq, cc);
        g_w[((size_t)f * P + p) * 2] = (float)gw.re;
        g_w[((size_t)f * P + p) * 2 + 1] = (float)gw.im;
        const cd wc{(double)w[((size_t)f * P + p) * 2], -(double)w[((size_t)f * P + p) * 2 + 1]};
        const cd t = cmul(wc, gq);
        sr += t.re;
        si += t.im;
    }
    red[0][threadIdx.x] = sr;
    red[1][threadIdx.x] = si;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { g_c[2 * f] = red[0][0]; g_c[2 * f + 1] = red[1][0]; }
}

// g_a[t] = m[t] Re( sum_f e^{-i theta_tf} g_c[f] m[f] / T )
__global__ __launch_bounds__(kBlock) void fft_coef_bwd_agg_kernel(const double* __restrict__ g_c, const double* __restrict__ m, int T, float* __restrict__ g_a) {
    __shared__ cd tw[kMaxFftT];
    fill_twiddle(tw, T);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        double acc = 0.0;
#pragma unroll 10
        for (int f = 0, k = 0; f < T; ++f) {           // k = (f * t) mod T
            const cd e = tw[k];
            const double sc = m[f] / (double)T;
            acc += (e.re * g_c[2 * f] + e.im * g_c[2 * f + 1]) * sc;   // Re(conj(e) g_c)
            k += t;
            if (k >= T) k -= T;
        }
        g_a[t] = (float)(acc * m[t]);
    }
}

}  // namespace lstep

using namespace lstep;

extern "C" int lstep_fft_coef_fwd(const float* filter_weight, const float* agg_weight, const double* mask, int32_t t_len, int32_t pe_dim, float* coef,
                                  double* c_out, void* stream) {
    if (t_len <= 0 || t_len > kMaxFftT || pe_dim <= 0) return set_error(LSTEP_EINVAL, "lstep_fft_coef_fwd: bad sizes (T <= 256)");
    if (!filter_weight || !agg_weight || !mask || !coef || !c_out) return set_error(LSTEP_EINVAL, "lstep_fft_coef_fwd: NULL pointer");
    hipLaunchKernelGGL(fft_coef_fwd_kernel, dim3((unsigned)t_len), dim3(kBlock), 0, (hipStream_t)stream, filter_weight, agg_weight, mask, (int)t_len,
                       (int)pe_dim, coef, c_out);
    return check_launch("fft_coef_fwd_kernel");
}

extern "C" int lstep_fft_coef_bwd(const float* grad_coef, const float* filter_weight, const double* c, const double* mask, int32_t t_len, int32_t pe_dim,
                                  float* grad_filter, float* grad_agg, double* scratch, void* stream) {
    if (t_len <= 0 || t_len > kMaxFftT || pe_dim <= 0) return set_error(LSTEP_EINVAL, "lstep_fft_coef_bwd: bad sizes (T <= 256)");
    if (!grad_coef || !filter_weight || !c || !mask || !grad_filter || !grad_agg || !scratch) return set_error(LSTEP_EINVAL, "lstep_fft_coef_bwd: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fft_coef_bwd_kernel, dim3((unsigned)t_len), dim3(kBlock), 0, s, grad_coef, filter_weight, c, (int)t_len, (int)pe_dim, grad_filter,
                       scratch);
    hipLaunchKernelGGL(fft_coef_bwd_agg_kernel, dim3(1), dim3(kBlock), 0, s, (const double*)scratch, mask, (int)t_len, grad_agg);
    return check_launch("fft_coef_bwd_kernel");
}
