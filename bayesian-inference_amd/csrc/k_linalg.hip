// Small dense linear algebra that runs once per model (or once per walker in the exact-form
// validation mode): workgroup-level blocked Cholesky, forward substitution, triangular inverse.
// These are latency-bound helper kernels, not the throughput path.
#include "internal.h"
#include "linalg_dev.h"

namespace gpemu {

// ------------------------------------------------------------------------------------------
// W = L^-1 for k lower-triangular N x N factors, written transposed and zero-padded:
//   Wt[p][j][i] = W_p[i][j]  (i >= j), 0 elsewhere, leading dimension Npad.
// One thread per column j of W (forward substitution of e_j), 64 columns per workgroup.
__global__ __launch_bounds__(64) void trtri_kernel(const double *__restrict__ L,
                                                   double *__restrict__ Wt, int64_t N,
                                                   int64_t Npad) {
  const int p = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (j >= N) return;
  const double *Lp = L + (int64_t)p * N * N;
  double *w = Wt + ((int64_t)p * Npad + j) * Npad;  // row j of Wt = column j of W
  w[j] = 1.0 / Lp[j * N + j];
  for (int64_t i = j + 1; i < N; ++i) {
    const double *Li = Lp + i * N;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t mm = j;
    for (; mm + 3 < i; mm += 4) {
      s0 = fma(Li[mm], w[mm], s0);
      s1 = fma(Li[mm + 1], w[mm + 1], s1);
      s2 = fma(Li[mm + 2], w[mm + 2], s2);
      s3 = fma(Li[mm + 3], w[mm + 3], s3);
    }
    for (; mm < i; ++mm) s0 = fma(Li[mm], w[mm], s0);
    w[i] = -((s0 + s1) + (s2 + s3)) / Li[i];
  }
}

int launch_trtri_lower_to_Wt(const double *dL, double *dWt, int64_t k, int64_t N, int64_t Npad,
                             hipStream_t st) {
  GP_HIP(hipMemsetAsync(dWt, 0, sizeof(double) * k * Npad * Npad, st));
  hipLaunchKernelGGL(trtri_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)k), dim3(64), 0, st, dL,
                     dWt, N, Npad);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

// ------------------------------------------------------------------------------------------
// single-matrix wrappers around the workgroup-level routines of linalg_dev.h
__global__ __launch_bounds__(CHOL_THREADS) void chol_kernel(double *A, int n, int ld, double *scratch,
                                                            int *info) {
  wg_cholesky_lower(A, n, ld, scratch, info);
}

int launch_cholesky(double *dA, int n, int ld, double *dscratch, int *dinfo, hipStream_t st) {
  hipLaunchKernelGGL(chol_kernel, dim3(1), dim3(CHOL_THREADS), 0, st, dA, n, ld, dscratch, dinfo);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
