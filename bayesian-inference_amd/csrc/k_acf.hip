// Normalised autocorrelation function of the chain stored on the device, averaged over walkers -- the arithmetic of
// emcee's integrated autocorrelation time (emcee/autocorr.py: function_1d + integrated_time, as called by
// ref: mcmc.py:111-119 through sampler.get_autocorr_time()).
//
// emcee transforms every (walker, parameter) series with an FFT over the WHOLE chain and then looks only at the lags up
// to Sokal's automatic window (the smallest M with M >= c tau(M), a few times tau).  The chain is already in HBM
// ([steps][W][d], one row per step), so here the lag products are formed directly, a block of lags at a time, and the
// host stops asking as soon as the window is found:
//     acf_s[l] = sum_t (x_s[t] - mean_s)(x_s[t + l] - mean_s)       (= what the zero-padded FFT yields)
//     f[l][dd] = 1/W sum_w acf_(w,dd)[l] / acf_(w,dd)[0]
// One thread owns one series s = (w, dd) -- consecutive threads read consecutive doubles of a chain row -- and 16
// consecutive lags, kept as 16 accumulators against a 31-value register window of the lagged series: 2 loads per 16 FMAs.
// The steps are cut into chunks (one per blockIdx.z) whose partial sums are added in a fixed order: deterministic.
#include <algorithm>

#include "internal.h"
#include "linalg_dev.h"
#include "sampler_internal.h"

namespace gpemu {

constexpr int ACF_LPT = 16;      // lags per thread
constexpr int ACF_TCHUNKS = 8;   // chunks of steps (partial sums)

// part[c][s] = sum of the series over chunk c
// (chain: first series of the block asked for; ld: doubles per chain row; S: series in the block)
__global__ __launch_bounds__(256) void acf_sum_kernel(const double *__restrict__ chain, int64_t n_t, int64_t S, int64_t ld,
                                                      int64_t tchunk, double *__restrict__ part) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const int64_t t0 = (int64_t)blockIdx.z * tchunk, t1 = std::min<int64_t>(t0 + tchunk, n_t);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int64_t t = t0;
  for (; t + 4 <= t1; t += 4) {
    a0 += chain[t * ld + s];
    a1 += chain[(t + 1) * ld + s];
    a2 += chain[(t + 2) * ld + s];
    a3 += chain[(t + 3) * ld + s];
  }
  for (; t < t1; ++t) a0 += chain[t * ld + s];
  part[(int64_t)blockIdx.z * S + s] = (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void acf_mean_kernel(const double *__restrict__ part, int64_t n_t, int64_t S, int nchunk,
                                                       double *__restrict__ mean) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double a = 0.0;
  for (int c = 0; c < nchunk; ++c) a += part[(int64_t)c * S + s];
  mean[s] = a / (double)n_t;
}

// part[(c * n_lags + l - lag0) * S + s] = sum over t in chunk c of y[t] y[t - l],  y = x - mean,  l in this thread's 16 lags
__global__ __launch_bounds__(256) void acf_lag_kernel(const double *__restrict__ chain, const double *__restrict__ mean,
                                                      int64_t n_t, int64_t S, int64_t ld, int64_t tchunk, int64_t lag0,
                                                      int n_lags, double *__restrict__ part) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const int64_t lb = lag0 + (int64_t)blockIdx.y * ACF_LPT;         // this thread's first lag
  const int64_t t0 = (int64_t)blockIdx.z * tchunk, t1 = std::min<int64_t>(t0 + tchunk, n_t);
  const double mu = mean[s];
  auto y = [&](int64_t t) -> double { return (t >= 0 && t < n_t) ? chain[t * ld + s] - mu : 0.0; };
  double acc[ACF_LPT];
#pragma unroll
  for (int j = 0; j < ACF_LPT; ++j) acc[j] = 0.0;
  // win[i] = y[tb - lb - 15 + i], i = 0..30, for the current base step tb; product (i, j): y[tb + i] y[tb + i - lb - j]
  double win[2 * ACF_LPT - 1];
#pragma unroll
  for (int i = 0; i < ACF_LPT - 1; ++i) win[i] = y(t0 - lb - (ACF_LPT - 1) + i);
  for (int64_t tb = t0; tb < t1; tb += ACF_LPT) {
    double cur[ACF_LPT];
#pragma unroll
    for (int i = 0; i < ACF_LPT; ++i) {
      cur[i] = (tb + i < t1) ? y(tb + i) : 0.0;                     // steps of the next chunk belong to its workgroup
      win[ACF_LPT - 1 + i] = y(tb - lb + i);
    }
#pragma unroll
    for (int i = 0; i < ACF_LPT; ++i)
#pragma unroll
      for (int j = 0; j < ACF_LPT; ++j) acc[j] = fma(cur[i], win[i - j + ACF_LPT - 1], acc[j]);
#pragma unroll
    for (int i = 0; i < ACF_LPT - 1; ++i) win[i] = win[i + ACF_LPT];
  }
#pragma unroll
  for (int j = 0; j < ACF_LPT; ++j) {
    const int64_t l = lb - lag0 + j;
    if (l < n_lags) part[((int64_t)blockIdx.z * n_lags + l) * S + s] = acc[j];
  }
}

// acf[l][s] = sum over chunks (fixed order); lag 0 is kept for the normalisation
__global__ __launch_bounds__(256) void acf_reduce_kernel(const double *__restrict__ part, int64_t S, int n_lags, int nchunk,
                                                         double *__restrict__ acf, double *__restrict__ acf0, int is_first) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int l = blockIdx.y;
  if (s >= S) return;
  double a = 0.0;
  for (int c = 0; c < nchunk; ++c) a += part[((int64_t)c * n_lags + l) * S + s];
  acf[(int64_t)l * S + s] = a;
  if (is_first && l == 0) acf0[s] = a;
}

// f[l][dd] = 1/nw sum over the block's nw walkers of acf[l][(w, dd)] / acf0[(w, dd)]; one workgroup per (l, dd)
__global__ __launch_bounds__(256) void acf_walker_mean_kernel(const double *__restrict__ acf, const double *__restrict__ acf0,
                                                              int64_t S, int d, int nw, double *__restrict__ f) {
  const int l = blockIdx.x, dd = blockIdx.y;
  double a = 0.0;
  for (int w = threadIdx.x; w < nw; w += 256) {
    const int64_t s = (int64_t)w * d + dd;
    a += acf[(int64_t)l * S + s] / acf0[s];
  }
  a = wg_sum(a);
  if (threadIdx.x == 0) f[(int64_t)l * d + dd] = a / (double)nw;
}

}  // namespace gpemu

using namespace gpemu;

extern "C" {

int gpemu_sampler_acf(gpemu_sampler *s, int64_t first, int64_t n_steps, int64_t w0, int64_t nw, int64_t lag0, int64_t n_lags,
                      double *f_out) {
  GP_ARG(s && f_out, "null pointer");
  GP_ARG(first >= 0 && n_steps >= 1 && first + n_steps <= s->chain_len, "chain range");
  GP_ARG(w0 >= 0 && nw >= 1 && w0 + nw <= s->W, "walker range");
  GP_ARG(lag0 >= 0 && n_lags >= 1 && n_lags <= 4096 && lag0 % ACF_LPT == 0, "lag block (lag0 must be a multiple of 16)");
  GP_ARG(n_lags <= n_steps, "more lags than steps");
  GP_HIP(hipSetDevice(s->device));
  hipStream_t st = s->stream;
  // only the series asked for -- walkers [w0, w0 + nw), e.g. one chain of a stacked closure run -- are transformed
  const int64_t ld = s->W * s->d, S = nw * s->d, n_t = n_steps;
  const double *chain = s->chain + first * ld + w0 * s->d;
  const int nchunk = (int)std::min<int64_t>(ACF_TCHUNKS, (n_t + ACF_LPT - 1) / ACF_LPT);
  const int64_t tchunk = round_up((n_t + nchunk - 1) / nchunk, ACF_LPT);
  const int nlg = (int)((n_lags + ACF_LPT - 1) / ACF_LPT);
  const dim3 block(256);
  const unsigned gs = (unsigned)((S + 255) / 256);
  // scratch: kept with the sampler between the calls of one estimate (mean and lag-0 products are made by the first);
  // every buffer has its own capacity (acf_acf is [n_lags][S], acf_part [nchunk][n_lags][S]: one does not bound the other)
  const bool fresh = lag0 == 0 || s->acf_first != first || s->acf_n != n_steps || s->acf_w0 != w0 || s->acf_nw != nw;
  const size_t need_part = sizeof(double) * (size_t)nchunk * (size_t)n_lags * (size_t)S;
  const size_t need_acf = sizeof(double) * (size_t)n_lags * (size_t)S, need_mean = sizeof(double) * (size_t)S;
  if (s->acf_part_bytes < need_part) {
    GP_HIP(hipStreamSynchronize(st));
    (void)hipFree(s->acf_part);
    s->acf_part = nullptr; s->acf_part_bytes = 0;
    GP_HIP(hipMalloc((void **)&s->acf_part, need_part));
    s->acf_part_bytes = need_part;
  }
  if (s->acf_acf_bytes < need_acf) {
    GP_HIP(hipStreamSynchronize(st));
    (void)hipFree(s->acf_acf);
    s->acf_acf = nullptr; s->acf_acf_bytes = 0;
    GP_HIP(hipMalloc((void **)&s->acf_acf, need_acf));
    s->acf_acf_bytes = need_acf;
  }
  if (s->acf_mean_bytes < need_mean) {
    GP_ARG(lag0 == 0, "the first block of an estimate must start at lag 0");
    GP_HIP(hipStreamSynchronize(st));
    (void)hipFree(s->acf_mean); (void)hipFree(s->acf_acf0);
    s->acf_mean = s->acf_acf0 = nullptr; s->acf_mean_bytes = 0;
    GP_HIP(hipMalloc((void **)&s->acf_mean, need_mean));
    GP_HIP(hipMalloc((void **)&s->acf_acf0, need_mean));
    s->acf_mean_bytes = need_mean;
  }
  if (fresh) {
    GP_ARG(lag0 == 0, "the first block of an estimate must start at lag 0");
    hipLaunchKernelGGL(acf_sum_kernel, dim3(gs, 1, (unsigned)nchunk), block, 0, st, chain, n_t, S, ld, tchunk, s->acf_part);
    hipLaunchKernelGGL(acf_mean_kernel, dim3(gs), block, 0, st, s->acf_part, n_t, S, nchunk, s->acf_mean);
    s->acf_first = first; s->acf_n = n_steps; s->acf_w0 = w0; s->acf_nw = nw;
  }
  hipLaunchKernelGGL(acf_lag_kernel, dim3(gs, (unsigned)nlg, (unsigned)nchunk), block, 0, st, chain, s->acf_mean, n_t, S, ld,
                     tchunk, lag0, (int)n_lags, s->acf_part);
  hipLaunchKernelGGL(acf_reduce_kernel, dim3(gs, (unsigned)n_lags), block, 0, st, s->acf_part, S, (int)n_lags, nchunk,
                     s->acf_acf, s->acf_acf0, lag0 == 0 ? 1 : 0);
  double *df = nullptr;
  GP_HIP(hipMalloc((void **)&df, sizeof(double) * (size_t)n_lags * s->d));
  hipLaunchKernelGGL(acf_walker_mean_kernel, dim3((unsigned)n_lags, (unsigned)s->d), block, 0, st, s->acf_acf, s->acf_acf0, S,
                     (int)s->d, (int)nw, df);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(f_out, df, sizeof(double) * (size_t)n_lags * s->d, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(df);
  if (e != hipSuccess) { set_error("sampler_acf: %s", hipGetErrorString(e)); return GPEMU_ERR_HIP; }
  return GPEMU_OK;
}

}  // extern "C"
