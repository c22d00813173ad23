// Observation ingestion on the device: what happens to the series
// BEFORE the Kalman filter -- standardisation, masking, and the reference's packed arrays -- for
// thousands of models at once.  All three kernels are single-pass-over-HBM byte movers; no LDS
// tiling beyond the block reductions, no MFMA.
//
//   standardize_kernel      Metran.standardize            /root/reference/metran/metran.py:102-121
//   mask_kernel             Metran.mask_observations      metran/metran.py:464-494 (DataFrame.mask)
//   pack_kernel             SPKalmanFilter.set_observations   metran/kalmanfilter.py:646-674
#include <hip/hip_runtime.h>

#include "mk_internal.h"

namespace mk {

// One 256-thread block per model.  Thread i owns series j = i % N and rows i / N, i / N + RP, ...
// (RP = 256 / N rows per pass), so a pass reads RP*N consecutive doubles of the [T,N] slab.
// pandas semantics: NaN is skipped, +-inf is a value; std uses ddof = 1 and a two-pass sum
// (mean first, then squared deviations, as pandas.core.nanops.nanvar does).
__global__ void __launch_bounds__(256)
standardize_kernel(long R, long T, int N, long bs, long ts, const double *in, double *out, double *mean,
                   double *stdev)
{
    __shared__ double s_sum[256];
    __shared__ long s_cnt[256];
    __shared__ double s_mean[64], s_std[64];
    const long r = blockIdx.x;
    const int RP = 256 / N, tid = threadIdx.x;
    const int j = tid % N, row0 = tid / N;
    const bool active = row0 < RP;
    const double *src = in + r * bs * N;

    double acc = 0.0;
    long cnt = 0;
    if (active)
        for (long t = row0; t < T; t += RP) {
            const double y = src[t * ts * N + j];
            if (y == y) {
                acc += y;
                ++cnt;
            }
        }
    s_sum[tid] = acc;
    s_cnt[tid] = cnt;
    __syncthreads();
    if (tid < N) { // fixed summation order over the RP partials: deterministic
        double a = 0.0;
        long c = 0;
        for (int k = 0; k < RP; ++k) {
            a += s_sum[k * N + tid];
            c += s_cnt[k * N + tid];
        }
        s_mean[tid] = c > 0 ? a / (double)c : __builtin_nan("");
        s_cnt[tid] = c;
    }
    __syncthreads();
    const double mu = s_mean[j];
    const long cj = s_cnt[j];
    __syncthreads();
    acc = 0.0;
    if (active)
        for (long t = row0; t < T; t += RP) {
            const double y = src[t * ts * N + j];
            if (y == y) acc = fma(y - mu, y - mu, acc);
        }
    s_sum[tid] = acc;
    __syncthreads();
    if (tid < N) {
        double a = 0.0;
        for (int k = 0; k < RP; ++k) a += s_sum[k * N + tid];
        const long c = s_cnt[tid];
        s_std[tid] = c > 1 ? sqrt(a / (double)(c - 1)) : __builtin_nan("");
        if (mean) mean[r * N + tid] = s_mean[tid];
        if (stdev) stdev[r * N + tid] = s_std[tid];
    }
    __syncthreads();
    (void)cj;
    if (out && active) {
        const double sd = s_std[j];
        double *dst = out + r * bs * N;
        for (long t = row0; t < T; t += RP) {
            const long o = t * ts * N + j;
            dst[o] = (src[o] - mu) / sd; // NaN stays NaN
        }
    }
}

// out = mask ? NaN : obs (DataFrame.mask); mask is one byte per observation, non-zero = hide
__global__ void mask_kernel(long count, const double *obs, const unsigned char *mask, double *out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = mask[i] ? __builtin_nan("") : obs[i];
}

// The reference's packed representation (kalmanfilter.py:646-674), one thread per (model, step):
// observations (missing -> 0.0), observation_indices (doubles holding ints, left-packed, rest 0.0),
// observation_count.  Same "+1e10 then nonzero()" quirk: a finite value of exactly -1e10 is dropped.
__global__ void pack_kernel(long RT, int N, const double *obs, double *observations, double *indices, long *count)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= RT) return;
    const double *y = obs + i * N;
    double *o = observations ? observations + i * N : nullptr;
    double *ix = indices ? indices + i * N : nullptr;
    int c = 0;
    for (int j = 0; j < N; ++j) {
        const double v = y[j];
        const bool keep = isfinite(v) && (v + 1e10 != 0.0);
        if (o) o[j] = keep ? v : 0.0;
        if (keep) {
            if (ix) ix[c] = (double)j;
            ++c;
        }
    }
    if (ix)
        for (int j = c; j < N; ++j) ix[j] = 0.0;
    if (count) count[i] = c;
}

hipError_t launch_standardize(long R, long T, int N, int time_major, const double *in, double *out, double *mean,
                              double *stdev, hipStream_t s)
{
    const long bs = time_major ? 1 : T, ts = time_major ? R : 1;
    hipLaunchKernelGGL(standardize_kernel, dim3((unsigned)R), dim3(256), 0, s, R, T, N, bs, ts, in, out, mean, stdev);
    return hipGetLastError();
}

hipError_t launch_mask(long count, const double *obs, const unsigned char *mask, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, count, obs, mask, out);
    return hipGetLastError();
}

hipError_t launch_pack(long RT, int N, const double *obs, double *observations, double *indices, long *count,
                       hipStream_t s)
{
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((RT + 255) / 256)), dim3(256), 0, s, RT, N, obs, observations,
                       indices, count);
    return hipGetLastError();
}

} // namespace mk
