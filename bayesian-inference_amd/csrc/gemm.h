// fp64 MFMA GEMM used by the fit-side kernels (see k_gemm.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace gpemu {

struct GemmArgs {
  const double *A = nullptr;
  const double *B = nullptr;
  double *C = nullptr;
  int64_t lda = 0, ldb = 0, ldc = 0;
  int64_t strideA = 0, strideB = 0, strideC = 0;  // per batch entry (blockIdx.z)
  // second batch level (several matrices, each with its own inner batch): blockIdx.z = z1 + batch1 * z2, operands at
  // z1 * stride + z2 * stride2.  batch1 = 0: one level.
  int batch1 = 0;
  int64_t stride2A = 0, stride2B = 0, stride2C = 0;
  int M = 0, N = 0, K = 0;
  double alpha = 1.0, beta = 0.0;
  int lower_only = 0;  // skip 64 x 64 tiles strictly above the diagonal
  // triangular operands: restrict the K range of a tile to where the operands can be non-zero
  int k_from_m = 0;    // op(A)[m][k] = 0 for k < m  (A upper triangular in (m, k)): start at the tile's first row
  int k_from_n = 0;    // op(B)[k][n] = 0 for k < n  (B lower triangular in (k, n)): start at the tile's first column
  int k_to_m = 0;      // op(A)[m][k] = 0 for k > m  (A lower triangular in (m, k)): stop after the tile's last row
  int xcd_batch = 0;   // set by launch_gemm: whole problems of a batch per XCD (see the kernel)
  int packed = 0;      // set by launch_gemm: 1-D grid over the lower-triangle tiles (lower_only, square)
  int pair = 0;        // set by launch_gemm: a workgroup takes a tile and its mirror image along N (1) or M (2)
};

// C = alpha op(A) op(B) + beta C; a_kmajor: A stored [k][m] else [m][k]; b_kmajor: B stored [k][n]
// else [n][k].  M, N multiples of 64, K multiple of 16.
int launch_gemm(const GemmArgs &g, bool a_kmajor, bool b_kmajor, int batch, hipStream_t st);

}  // namespace gpemu
