// Reference-form outputs: the full predictive covariance of emulation.predict and the exact-form
// (F x F Cholesky per walker) log-likelihood used to validate the low-rank path.
//
//   predict_full_kernel   ref: emulation.py:504-548  central_value (B,F) and cov (B,F,F)
//                         cov_b = (S diag(var_b) S^T + C_unexpl / n_div) o (s s^T)     HBM-write-bound
//   loglik_exact_kernel   ref: log_posterior.py:87-99, 104-146   one workgroup per walker
#include "internal.h"
#include "linalg_dev.h"

namespace gpemu {

constexpr int PF_TF = 32;    // rows (f) per tile
constexpr int PF_TG = 128;   // cols (g) per tile
constexpr int PF_NB = 8;     // walkers per workgroup (C_unexpl tile reused across them)

// grid (ceil(F/128), ceil(F/32), ceil(B/8)), block 256.  Thread: 8 rows x 2 columns.
// PAIR = true (F even): the two columns are adjacent (2 lane, 2 lane + 1) and each row is written with
// one 16-byte store per lane, i.e. 1 KiB contiguous per wave-instruction -- the kernel is store-issue
// bound with 8-byte stores.  PAIR = false: columns lane and lane + 64, 8-byte stores (any F).
typedef double d2x __attribute__((ext_vector_type(2)));

template <bool PAIR>
__global__ __launch_bounds__(256) void predict_full_kernel(
    const double *__restrict__ mean, const double *__restrict__ var, const double *__restrict__ comp,
    const double *__restrict__ smean, const double *__restrict__ sscale,
    const double *__restrict__ cun, double *__restrict__ cv, double *__restrict__ cov, int64_t B,
    int F, int k, double inv_ndiv) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *cf = sm;                   // [k][PF_TF]
  double *cg = cf + k * PF_TF;       // [k][PF_TG]
  double *wv = cg + k * PF_TG;       // [k]   var of the current walker
  const int tid = threadIdx.x, lane = tid & 63, wr = tid >> 6;
  const int g0 = blockIdx.x * PF_TG, f0 = blockIdx.y * PF_TF;
  const int64_t bb0 = (int64_t)blockIdx.z * PF_NB;
  const int c0 = PAIR ? 2 * lane : lane, c1 = PAIR ? 2 * lane + 1 : lane + 64;   // tile columns
  for (int idx = tid; idx < k * PF_TF; idx += 256) {
    int p = idx / PF_TF, f = f0 + idx % PF_TF;
    cf[idx] = (f < F) ? comp[(int64_t)p * F + f] : 0.0;
  }
  for (int idx = tid; idx < k * PF_TG; idx += 256) {
    int p = idx / PF_TG, g = g0 + idx % PF_TG;
    cg[idx] = (g < F) ? comp[(int64_t)p * F + g] : 0.0;
  }
  // per-thread constants: truncation covariance and scale products of its 8 x 2 elements
  double cu[8][2], ss[8][2];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int f = f0 + wr * 8 + r;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int g = g0 + (c == 0 ? c0 : c1);
      bool ok = (f < F) && (g < F);
      cu[r][c] = ok ? cun[(int64_t)f * F + g] * inv_ndiv : 0.0;
      ss[r][c] = ok ? sscale[f] * sscale[g] : 0.0;
    }
  }
  for (int ib = 0; ib < PF_NB; ++ib) {
    const int64_t b = bb0 + ib;
    if (b >= B) break;
    __syncthreads();
    if (tid < k) wv[tid] = var[b * k + tid];
    __syncthreads();
    double acc[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r][0] = acc[r][1] = 0.0;
    for (int p = 0; p < k; ++p) {
      const double v = wv[p];
      const double b0v = cg[p * PF_TG + c0] * v, b1v = cg[p * PF_TG + c1] * v;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const double a = cf[p * PF_TF + wr * 8 + r];
        acc[r][0] = fma(a, b0v, acc[r][0]);
        acc[r][1] = fma(a, b1v, acc[r][1]);
      }
    }
    double *covb = cov + (int64_t)b * F * F;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int f = f0 + wr * 8 + r;
      if (f < F) {
        const double o0 = (acc[r][0] + cu[r][0]) * ss[r][0], o1 = (acc[r][1] + cu[r][1]) * ss[r][1];
        if (PAIR) {
          if (g0 + c0 < F) *reinterpret_cast<d2x *>(covb + (int64_t)f * F + g0 + c0) = d2x{o0, o1};   // F even => c1 < F too
        } else {
          if (g0 + c0 < F) covb[(int64_t)f * F + g0 + c0] = o0;
          if (g0 + c1 < F) covb[(int64_t)f * F + g0 + c1] = o1;
        }
      }
    }
    // central value: done by the blocks of the first tile row (f0 == 0), one column chunk each
    if (blockIdx.y == 0) {
      for (int g = g0 + tid; g < g0 + PF_TG && g < F; g += 256) {
        double s = 0.0;
        for (int p = 0; p < k; ++p) s = fma(mean[b * k + p], cg[p * PF_TG + (g - g0)], s);
        cv[b * F + g] = s * sscale[g] + smean[g];  // ref: emulation.py:508-509
      }
    }
  }
}


// Row-streaming variant: a workgroup owns 8 whole rows (all columns, up to 512 per pass), so its stores
// form one contiguous 8 * F * 8-byte stream per walker instead of 1 KiB pieces at a 4 KB stride.
constexpr int PR_ROWS = 8;
constexpr int PR_COLS = 512;   // 256 threads x 2 adjacent columns

template <bool PAIR>
__global__ __launch_bounds__(256) void predict_full_rows_kernel(
    const double *__restrict__ mean, const double *__restrict__ var, const double *__restrict__ comp,
    const double *__restrict__ smean, const double *__restrict__ sscale,
    const double *__restrict__ cun, double *__restrict__ cv, double *__restrict__ cov, int64_t B,
    int F, int k, double inv_ndiv) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *cf = sm;                    // [k][PR_ROWS]
  double *cg = cf + k * PR_ROWS;      // [k][PR_COLS]
  double *wv = cg + k * PR_COLS;      // [PF_NB][k]  variances of this workgroup's samples
  const int tid = threadIdx.x;
  const int g0 = blockIdx.x * PR_COLS, f0 = blockIdx.y * PR_ROWS;
  const int64_t bb0 = (int64_t)blockIdx.z * PF_NB;
  const int c0 = PAIR ? 2 * tid : tid, c1 = PAIR ? 2 * tid + 1 : tid + 256;
  for (int idx = tid; idx < k * PR_ROWS; idx += 256) {
    int p = idx / PR_ROWS, f = f0 + idx % PR_ROWS;
    cf[idx] = (f < F) ? comp[(int64_t)p * F + f] : 0.0;
  }
  for (int idx = tid; idx < k * PR_COLS; idx += 256) {
    int p = idx / PR_COLS, g = g0 + idx % PR_COLS;
    cg[idx] = (g < F) ? comp[(int64_t)p * F + g] : 0.0;
  }
  for (int idx = tid; idx < k * PF_NB; idx += 256) {
    const int64_t b = bb0 + idx / k;
    wv[idx] = (b < B) ? var[b * k + idx % k] : 0.0;
  }
  double cu[PR_ROWS][2], ss[PR_ROWS][2];
#pragma unroll
  for (int r = 0; r < PR_ROWS; ++r) {
    int f = f0 + r;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int g = g0 + (c == 0 ? c0 : c1);
      bool ok = (f < F) && (g < F);
      cu[r][c] = ok ? cun[(int64_t)f * F + g] * inv_ndiv : 0.0;
      ss[r][c] = ok ? sscale[f] * sscale[g] : 0.0;
    }
  }
  for (int ib = 0; ib < PF_NB; ++ib) {
    const int64_t b = bb0 + ib;
    if (b >= B) break;
    if (ib == 0) __syncthreads();
    double acc[PR_ROWS][2];
#pragma unroll
    for (int r = 0; r < PR_ROWS; ++r) acc[r][0] = acc[r][1] = 0.0;
    for (int p = 0; p < k; ++p) {
      const double v = wv[ib * k + p];
      const double b0v = cg[p * PR_COLS + c0] * v, b1v = cg[p * PR_COLS + c1] * v;
#pragma unroll
      for (int r = 0; r < PR_ROWS; ++r) {
        const double a = cf[p * PR_ROWS + r];
        acc[r][0] = fma(a, b0v, acc[r][0]);
        acc[r][1] = fma(a, b1v, acc[r][1]);
      }
    }
    double *covb = cov + (int64_t)b * F * F;
#pragma unroll
    for (int r = 0; r < PR_ROWS; ++r) {
      int f = f0 + r;
      if (f < F) {
        const double o0 = (acc[r][0] + cu[r][0]) * ss[r][0], o1 = (acc[r][1] + cu[r][1]) * ss[r][1];
        if (PAIR) {
          if (g0 + c0 < F) *reinterpret_cast<d2x *>(covb + (int64_t)f * F + g0 + c0) = d2x{o0, o1};
        } else {
          if (g0 + c0 < F) covb[(int64_t)f * F + g0 + c0] = o0;
          if (g0 + c1 < F) covb[(int64_t)f * F + g0 + c1] = o1;
        }
      }
    }
    if (blockIdx.y == 0) {   // central value (ref: emulation.py:508-509), one column chunk per block
      for (int g = g0 + tid; g < g0 + PR_COLS && g < F; g += 256) {
        double s = 0.0;
        for (int p = 0; p < k; ++p) s = fma(mean[b * k + p], cg[p * PR_COLS + (g - g0)], s);
        cv[b * F + g] = s * sscale[g] + smean[g];
      }
    }
  }
}

int launch_predict_full(gpemu_model *m, int64_t B, double n_div, double *dcv, double *dcov, hipStream_t st) {
  const int F = (int)m->F, k = (int)m->k;
  const double *dmean = m->ws.mean, *dvar = m->ws.var;
  dim3 grid((unsigned)((F + PF_TG - 1) / PF_TG), (unsigned)((F + PF_TF - 1) / PF_TF),
            (unsigned)((B + PF_NB - 1) / PF_NB));
  size_t shm = sizeof(double) * (size_t)(k * (PF_TF + PF_TG) + k);
  // 16-byte stores need every row start of dcov 16-byte aligned: F even and an aligned base
  const bool pair = F % 2 == 0 && (reinterpret_cast<uintptr_t>(dcov) & 15) == 0;
  const size_t shm_rows = sizeof(double) * (size_t)(k * (PR_ROWS + PR_COLS) + k * PF_NB);
  if (shm_rows <= 64 * 1024) {   // whole rows per workgroup: contiguous store streams
    dim3 g2((unsigned)((F + PR_COLS - 1) / PR_COLS), (unsigned)((F + PR_ROWS - 1) / PR_ROWS),
            (unsigned)((B + PF_NB - 1) / PF_NB));
    if (pair)
      hipLaunchKernelGGL(predict_full_rows_kernel<true>, g2, dim3(256), shm_rows, st, dmean, dvar, m->comp,
                         m->smean, m->sscale, m->cunexpl, dcv, dcov, B, F, k, 1.0 / n_div);
    else
      hipLaunchKernelGGL(predict_full_rows_kernel<false>, g2, dim3(256), shm_rows, st, dmean, dvar, m->comp,
                         m->smean, m->sscale, m->cunexpl, dcv, dcov, B, F, k, 1.0 / n_div);
    GP_HIP(hipGetLastError());
    return GPEMU_OK;
  }
  if (pair)
    hipLaunchKernelGGL(predict_full_kernel<true>, grid, dim3(256), shm, st, dmean, dvar, m->comp,
                       m->smean, m->sscale, m->cunexpl, dcv, dcov, B, F, k, 1.0 / n_div);
  else
    hipLaunchKernelGGL(predict_full_kernel<false>, grid, dim3(256), shm, st, dmean, dvar, m->comp,
                       m->smean, m->sscale, m->cunexpl, dcv, dcov, B, F, k, 1.0 / n_div);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

// ------------------------------------------------------------------------------------------
// Exact form: Sigma_b = cov_b + diag(y_err^2) (lower triangle), Cholesky, forward solve, log-det.
// One workgroup per walker slot; walkers are processed grid-stride so the scratch stays bounded.
__global__ __launch_bounds__(CHOL_THREADS) void loglik_exact_kernel(
    const double *__restrict__ Xq, const double *__restrict__ lo, const double *__restrict__ hi,
    const double *__restrict__ mean_part, const double *__restrict__ vsq_part,
    const double *__restrict__ kdiag, const double *__restrict__ comp,
    const double *__restrict__ smean, const double *__restrict__ sscale,
    const double *__restrict__ cun, const double *__restrict__ yexp, const double *__restrict__ yerr,
    const int *__restrict__ blk_of, double *__restrict__ scratch, double *__restrict__ out, int64_t B,
    int64_t Bcap, int d, int F,
    int k, int nchunk, int nrb, double inv_ndiv) {
  __shared__ double s_mu[64], s_var[64];
  __shared__ int s_inside;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t per = (int64_t)F * F + chol_scratch_size(F) + F;
  double *S = scratch + (int64_t)blockIdx.x * per;
  double *PT = S + (int64_t)F * F;
  double *dy = PT + chol_scratch_size(F);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    if (tid == 0) {
      int in = 1;
      for (int dd = 0; dd < d; ++dd) in &= (Xq[b * DPAD + dd] > lo[dd]) && (Xq[b * DPAD + dd] < hi[dd]);
      s_inside = in;
    }
    if (tid < k) {
      double mu = 0.0, vs = 0.0;
      for (int c = 0; c < nchunk; ++c) mu += mean_part[(b * k + tid) * nchunk + c];
      for (int r = 0; r < nrb; ++r) vs += vsq_part[(b * k + tid) * nrb + r];
      double v = kdiag[tid] - vs;
      if (v < 0.0) v = 0.0;
      double sd = sqrt(v);
      s_mu[tid] = mu;
      s_var[tid] = sd * sd;
    }
    __syncthreads();
    if (!s_inside) {
      if (tid == 0) out[b] = -INFINITY;
      continue;
    }
    for (int64_t idx = tid; idx < (int64_t)F * F; idx += nthr) {
      int f = (int)(idx / F), g = (int)(idx - (int64_t)f * F);
      if (g > f) continue;
      if (blk_of[f] != blk_of[g]) {  // cross-observable covariance is dropped by the reference's merge
        S[idx] = 0.0;
        continue;
      }
      double acc = 0.0;
      for (int p = 0; p < k; ++p) acc = fma(comp[(int64_t)p * F + f] * s_var[p], comp[(int64_t)p * F + g], acc);
      double v = (acc + cun[idx] * inv_ndiv) * (sscale[f] * sscale[g]);
      if (f == g) v += yerr[f] * yerr[f];
      S[idx] = v;
    }
    for (int f = tid; f < F; f += nthr) {
      double s = 0.0;
      for (int p = 0; p < k; ++p) s = fma(s_mu[p], comp[(int64_t)p * F + f], s);
      dy[f] = (s * sscale[f] + smean[f]) - yexp[f];
    }
    __syncthreads();
    wg_cholesky_lower(S, F, F, PT, nullptr);
    wg_forward_solve_vec(S, F, F, dy);
    double q = 0.0, ld = 0.0;
    for (int f = tid; f < F; f += nthr) {
      q = fma(dy[f], dy[f], q);
      ld += log(S[(int64_t)f * F + f]);
    }
    q = wg_sum(q);
    ld = wg_sum(ld);
    if (tid == 0) out[b] = -0.5 * q - ld;  // ref: log_posterior.py:146
  }
}

int launch_loglik_exact(gpemu_model *m, int64_t B, const double *dXq, double *dout, hipStream_t st) {
  const int F = (int)m->F;
  const int64_t per = (int64_t)F * F + chol_scratch_size(F) + F;
  int nwg = (int)(B < 256 ? B : 256);
  if (m->exact_scratch_size < per * nwg) {
    GP_HIP(hipStreamSynchronize(st));
    hipFree(m->exact_scratch);
    m->exact_scratch = nullptr; m->exact_scratch_size = 0;
    GP_HIP(hipMalloc((void **)&m->exact_scratch, sizeof(double) * (size_t)(per * nwg)));
    m->exact_scratch_size = per * nwg;
  }
  const Workspace &w = m->ws;
  hipLaunchKernelGGL(loglik_exact_kernel, dim3((unsigned)nwg), dim3(CHOL_THREADS), 0, st, dXq, m->lo,
                     m->hi, w.mean_part, w.vsq_part, m->kdiag, m->comp, m->smean, m->sscale,
                     m->cunexpl, m->yexp, m->yerr, m->blk_of, m->exact_scratch, dout, B, w.Bcap, (int)m->d,
                     F, (int)m->k, w.cur_nchunk, w.cur_nrb, 1.0 / m->n_div);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
