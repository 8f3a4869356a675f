// Workgroup-level dense kernels (device functions) shared by the one-off setup kernels and the
// exact-form (reference-form) likelihood.  All matrices row-major float64.
#pragma once
#include <hip/hip_runtime.h>

namespace gpemu {

constexpr int CHOL_THREADS = 1024;
constexpr int CHOL_NB = 32;

__device__ __forceinline__ int chol_scratch_ld(int n) { return (n + 63) / 64 * 64; }
// scratch doubles needed by wg_cholesky_lower for an n x n matrix
__host__ __device__ static inline int64_t chol_scratch_size(int64_t n) { return CHOL_NB * ((n + 63) / 64 * 64); }

// In-place lower Cholesky A = C C^T of the n x n matrix A (leading dimension ld) by ONE workgroup of
// CHOL_THREADS threads; only the lower triangle is read and written.  Right-looking, panel width
// 32: diagonal block in LDS (one wave), panel solve (thread per row), trailing update with the panel
// kept transposed in `PT` ([32][chol_scratch_ld(n)], global/L2) so that reads are coalesced.
// *info (if not null) receives 1 + index of the first non-positive pivot (LAPACK dpotrf info), the
// factor then contains NaNs from that column on.
__device__ inline void wg_cholesky_lower(double *A, int n, int ld, double *PT, int *info) {
  __shared__ double D[CHOL_NB][CHOL_NB + 1];
  __shared__ int s_bad;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwave = nthr >> 6;
  const int ldp = chol_scratch_ld(n);
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += CHOL_NB) {
    const int nb = (n - j0 < CHOL_NB) ? (n - j0) : CHOL_NB;
    for (int idx = tid; idx < nb * nb; idx += nthr) {
      int r = idx / nb, c = idx - r * nb;
      D[r][c] = (c <= r) ? A[(int64_t)(j0 + r) * ld + j0 + c] : 0.0;
    }
    __syncthreads();
    if (wave == 0) {
      for (int jj = 0; jj < nb; ++jj) {
        double piv = D[jj][jj];
        if (!(piv > 0.0) && lane == 0 && s_bad == 0) s_bad = j0 + jj + 1;
        piv = sqrt(piv);
        __builtin_amdgcn_wave_barrier();
        if (lane == jj) D[jj][jj] = piv;
        if (lane > jj && lane < nb) D[lane][jj] = D[lane][jj] / piv;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane > jj && lane < nb) {
          double lij = D[lane][jj];
          for (int c = jj + 1; c <= lane; ++c) D[lane][c] -= lij * D[c][jj];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    for (int idx = tid; idx < nb * nb; idx += nthr) {
      int r = idx / nb, c = idx - r * nb;
      if (c <= r) A[(int64_t)(j0 + r) * ld + j0 + c] = D[r][c];
    }
    const int r0 = j0 + nb;  // first trailing row
    // panel: X D^T = A[r0:, j0:j0+nb], one thread per row, eight columns at a time in registers; the columns of the
    // earlier groups are read back from PT (coalesced over the rows, just written by this thread).  Same operation
    // order per element as a straight 32-column register solve -- which held x[32] plus the hoisted D operands of 496
    // unrolled FMAs and spilled 800-940 VGPRs in the kernels that inline this (lik_setup_kernel, loglik_exact_kernel).
    constexpr int PG = 8;
    for (int i = r0 + tid; i < n; i += nthr) {
      double *Ai = A + (int64_t)i * ld + j0;
      for (int c0 = 0; c0 < nb; c0 += PG) {
        double x[PG];
#pragma unroll
        for (int j = 0; j < PG; ++j) x[j] = (c0 + j < nb) ? Ai[c0 + j] : 0.0;
        for (int mm = 0; mm < c0; ++mm) {
          const double xm = PT[(int64_t)mm * ldp + i];
#pragma unroll
          for (int j = 0; j < PG; ++j)
            if (c0 + j < nb) x[j] = fma(-xm, D[c0 + j][mm], x[j]);
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
          if (c0 + j < nb) {
            double sacc = x[j];
#pragma unroll
            for (int mm = 0; mm < j; ++mm) sacc = fma(-x[mm], D[c0 + j][c0 + mm], sacc);
            x[j] = sacc / D[c0 + j][c0 + j];
            Ai[c0 + j] = x[j];
            PT[(int64_t)(c0 + j) * ldp + i] = x[j];
          }
        }
      }
    }
    __syncthreads();
    // trailing update A[i][c] -= sum_m P[i][m] P[c][m], r0 <= c <= i < n ; one wave per (row, 64 cols)
    for (int i = r0 + wave; i < n; i += nwave) {
      for (int cb = r0; cb <= i; cb += 64) {
        int c = cb + lane;
        double acc = 0.0;
        if (c <= i) {
          for (int mm = 0; mm < nb; ++mm) acc = fma(PT[(int64_t)mm * ldp + i], PT[(int64_t)mm * ldp + c], acc);
          A[(int64_t)i * ld + c] -= acc;
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0 && info) *info = s_bad;
}

// Forward substitution C Z = B for nrhs right-hand sides, B row-major [n][ldb], in place.
// One thread per right-hand side (nrhs <= blockDim.x); C lower, leading dimension ld.
__device__ inline void wg_forward_solve_multi(const double *C, int n, int ld, double *Bm, int ldb,
                                              int nrhs) {
  const int c = threadIdx.x;
  if (c < nrhs) {
    for (int i = 0; i < n; ++i) {
      const double *Ci = C + (int64_t)i * ld;
      double s0 = 0.0, s1 = 0.0;
      int mm = 0;
      for (; mm + 1 < i; mm += 2) {
        s0 = fma(Ci[mm], Bm[(int64_t)mm * ldb + c], s0);
        s1 = fma(Ci[mm + 1], Bm[(int64_t)(mm + 1) * ldb + c], s1);
      }
      if (mm < i) s0 = fma(Ci[mm], Bm[(int64_t)mm * ldb + c], s0);
      Bm[(int64_t)i * ldb + c] = (Bm[(int64_t)i * ldb + c] - (s0 + s1)) / Ci[i];
    }
  }
  __syncthreads();
}

// Forward substitution C z = y for ONE right-hand side by the whole workgroup, in place in y[n]
// (global or LDS).  Blocks of 32 rows: wave 0 solves the diagonal block, all threads then update
// the remaining rows.
__device__ inline void wg_forward_solve_vec(const double *C, int n, int ld, double *y) {
  __shared__ double zb[CHOL_NB];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int j0 = 0; j0 < n; j0 += CHOL_NB) {
    const int nb = (n - j0 < CHOL_NB) ? (n - j0) : CHOL_NB;
    if (wave == 0) {
      double yi = (lane < nb) ? y[j0 + lane] : 0.0;
      for (int jj = 0; jj < nb; ++jj) {
        double zj = __shfl(yi, jj) / C[(int64_t)(j0 + jj) * ld + j0 + jj];
        if (lane == jj) yi = zj;
        if (lane > jj && lane < nb) yi = fma(-C[(int64_t)(j0 + lane) * ld + j0 + jj], zj, yi);
      }
      if (lane < nb) {
        zb[lane] = yi;
        y[j0 + lane] = yi;
      }
    }
    __syncthreads();
    for (int i = j0 + nb + tid; i < n; i += nthr) {
      const double *Ci = C + (int64_t)i * ld + j0;
      double s = 0.0;
      for (int mm = 0; mm < nb; ++mm) s = fma(Ci[mm], zb[mm], s);
      y[i] -= s;
    }
    __syncthreads();
  }
}

// workgroup sum of one double per thread (result valid in every thread)
__device__ inline double wg_sum(double v) {
  __shared__ double part[CHOL_THREADS / 64];
  __shared__ double total;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < nwave; ++w) s += part[w];
    total = s;
  }
  __syncthreads();
  return total;
}


}  // namespace gpemu
