// S: temporal-neighbour lookup ('recent'), and T: standalone time encoder.
//   reference: utils/utils.py:129-146,148-213 ; models/modules.py:27-39
#include "lstep_common.h"

namespace lstep {

// One wave per query row.  HBM traffic per row: <= ~3 probes rounds of the time array + K * 16 B of CSR
// entries read + K * 20 B written; the write of the [M, K] outputs dominates for K = time_gap.
__global__ __launch_bounds__(kBlock) void sample_recent_kernel(lstep_csr_t csr, const int64_t* __restrict__ node_ids,
                                                                int64_t num_ids, const double* __restrict__ times,
                                                                int64_t num_pairs, int K, int64_t* __restrict__ out_nbr,
                                                                int64_t* __restrict__ out_eid, float* __restrict__ out_nt,
                                                                int32_t* __restrict__ out_count) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (row >= num_ids) return;
    int64_t lo = 0, cnt = 0;
    if (row < num_pairs) {  // rows past the shorter input stay all-padding (zip truncation)
        const int64_t node = node_ids[row];
        if (node >= 0 && node < csr.num_rows) {
            lo = csr.indptr[node];
            cnt = wave_count_before(csr.ts, lo, csr.indptr[node + 1], times[row], lane);
        }
    }
    const int64_t take = cnt < K ? cnt : K;
    const int64_t first = lo + cnt - take;  // CSR entry that lands in slot K - take
    int64_t* nb = out_nbr + row * K;
    int64_t* ei = out_eid + row * K;
    float* nt = out_nt + row * K;
    for (int s = lane; s < K; s += kWave) {
        const int64_t j = s - (K - take);
        int64_t a = 0, b = 0;
        float c = 0.0f;
        if (j >= 0) {
            a = csr.nbr[first + j];
            b = csr.eid[first + j];
            c = (float)csr.ts[first + j];
        }
        nb[s] = a;
        ei[s] = b;
        nt[s] = c;
    }
    if (out_count != nullptr && lane == 0) out_count[row] = (int32_t)cnt;
}

__global__ __launch_bounds__(kBlock) void time_encode_kernel(const float* __restrict__ dt, const uint8_t* __restrict__ zero_mask,
                                                              int64_t n, const float* __restrict__ w, const float* __restrict__ b,
                                                              int D, float* __restrict__ out) {
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        const bool zero = zero_mask != nullptr && zero_mask[r] != 0;
        out[i] = zero ? 0.0f : time_feat(dt[r], w[d], b[d]);
    }
}

}  // namespace lstep

using namespace lstep;

extern "C" int lstep_sample_recent(const lstep_csr_t* csr, const int64_t* node_ids, int64_t num_ids, const double* times,
                                   int64_t num_times, int32_t num_neighbors, int64_t* out_nbr, int64_t* out_eid,
                                   float* out_nt, int32_t* out_count, void* stream) {
    if (num_neighbors <= 0) return set_error(LSTEP_EINVAL, "Number of sampled neighbors for each node should be greater than 0!");
    if (csr == nullptr || num_ids < 0 || num_times < 0) return set_error(LSTEP_EINVAL, "lstep_sample_recent: bad arguments");
    if (num_ids == 0) return LSTEP_OK;
    if (!csr->indptr || !csr->nbr || !csr->eid || !csr->ts || !node_ids || !times || !out_nbr || !out_eid || !out_nt)
        return set_error(LSTEP_EINVAL, "lstep_sample_recent: NULL pointer");
    const int64_t pairs = num_ids < num_times ? num_ids : num_times;
    const unsigned grid = (unsigned)((num_ids + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(sample_recent_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, *csr, node_ids, num_ids, times,
                       pairs, (int)num_neighbors, out_nbr, out_eid, out_nt, out_count);
    return check_launch("sample_recent_kernel");
}

extern "C" int lstep_time_encode(const float* dt, const uint8_t* zero_mask, int64_t n, const float* w, const float* b,
                                 int32_t time_dim, float* out, void* stream) {
    if (n < 0 || time_dim <= 0) return set_error(LSTEP_EINVAL, "lstep_time_encode: bad sizes");
    if (n == 0) return LSTEP_OK;
    if (!dt || !w || !b || !out) return set_error(LSTEP_EINVAL, "lstep_time_encode: NULL pointer");
    int64_t blocks = (n * time_dim + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(time_encode_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, dt, zero_mask, n, w, b,
                       (int)time_dim, out);
    return check_launch("time_encode_kernel");
}
