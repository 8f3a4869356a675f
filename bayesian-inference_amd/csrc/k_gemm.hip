// General fp64 GEMM on the matrix cores (v_mfma_f64_16x16x4_f64), the workhorse of the fit side:
// Cholesky trailing updates (SYRK), panel solves, triangular inverse, K^-1 = W^T W.
//
//   C[M x N] = alpha * op(A)[M x K] * op(B)[K x N] + beta * C          (row-major, leading dims)
//
// Operand layouts are template flags so that every caller reads memory the way it lies:
//   A_KMAJOR: A is stored [k][m] (i.e. op(A) = A^T of a row-major K x M array), else [m][k]
//   B_KMAJOR: B is stored [k][n], else [n][k] (op(B) = B^T of a row-major N x K array)
// All of M, N must be multiples of 64 and K a multiple of 16 (callers pad; the fit workspace is
// padded to 64 with an identity tail).  Tile 64 x 64, 256 threads = 4 waves (2 x 2, 32 x 32 each),
// K step 16, LDS double buffered: 40 KiB per workgroup, four workgroups (16 waves) per CU -- with K step 32 only two
// fit and the matrix pipes starve (batched N = 1000: 5.4 -> 5.0 ms per 64 problems; K step 8 is slower again).
// `lower_only` skips tiles strictly above the diagonal (SYRK).
#include "internal.h"
#include "gemm.h"

namespace gpemu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int GT = 64;      // tile edge
constexpr int GK = 16;      // K step (16: 40 KiB of LDS per workgroup, four workgroups per CU)
constexpr int GSK = 80;     // LDS stride of a k-major tile  [32][80]   ((2*80) % 64 == 32)
constexpr int GSM = 18;     // LDS stride of an m-major tile [64][18]

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double sA[2][GK * GSK];   // either layout fits (16 x 80 >= 64 x 18)
  __shared__ __attribute__((aligned(16))) double sB[2][GK * GSK];
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  if (g.lower_only && n0 > m0) return;
  const int z1 = g.batch1 > 0 ? (int)(blockIdx.z % g.batch1) : (int)blockIdx.z;
  const int z2 = g.batch1 > 0 ? (int)(blockIdx.z / g.batch1) : 0;
  const double *A = g.A + (int64_t)z1 * g.strideA + (int64_t)z2 * g.stride2A;
  const double *B = g.B + (int64_t)z1 * g.strideB + (int64_t)z2 * g.stride2B;
  double *C = g.C + (int64_t)z1 * g.strideC + (int64_t)z2 * g.stride2C;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;

  // staging: each operand tile is 64 x 32 doubles = 1024 x 16 B -> 4 per thread
  //   k-major source [k][m]: 32 rows of 64 doubles  -> idx = tid + 256 r: row = idx >> 5, c2 = idx & 31
  //   m-major source [m][k]: 64 rows of 32 doubles  -> idx = tid + 256 r: row = idx >> 4, c2 = idx & 15
  constexpr int NST = GT * GK / 2 / 256;   // 16-byte pieces of an operand tile per thread
  // two register stages: the loads of k-tile t + 2 are issued while k-tile t is multiplied and k-tile t + 1 waits in the
  // other stage for its turn to be written to LDS -- two k-tiles of compute cover a load's trip (with one stage and
  // K step 16 the matrix pipes were busy 25 % of the time in the batched N = 1000 case)
  d2 ra0[NST], rb0[NST], ra1[NST], rb1[NST];
  auto gload = [&](int kt, d2 (&ra)[NST], d2 (&rb)[NST]) {
#pragma unroll
    for (int r = 0; r < NST; ++r) {
      const int idx = tid + 256 * r;
      if (A_KMAJOR) ra[r] = *reinterpret_cast<const d2 *>(A + (int64_t)(kt * GK + (idx >> 5)) * g.lda + m0 + 2 * (idx & 31));
      else ra[r] = *reinterpret_cast<const d2 *>(A + (int64_t)(m0 + idx / (GK / 2)) * g.lda + kt * GK + 2 * (idx % (GK / 2)));
      if (B_KMAJOR) rb[r] = *reinterpret_cast<const d2 *>(B + (int64_t)(kt * GK + (idx >> 5)) * g.ldb + n0 + 2 * (idx & 31));
      else rb[r] = *reinterpret_cast<const d2 *>(B + (int64_t)(n0 + idx / (GK / 2)) * g.ldb + kt * GK + 2 * (idx % (GK / 2)));
    }
  };
  auto sstore = [&](int buf, const d2 (&ra)[NST], const d2 (&rb)[NST]) {
#pragma unroll
    for (int r = 0; r < NST; ++r) {
      const int idx = tid + 256 * r;
      if (A_KMAJOR) *reinterpret_cast<d2 *>(&sA[buf][(idx >> 5) * GSK + 2 * (idx & 31)]) = ra[r];
      else *reinterpret_cast<d2 *>(&sA[buf][(idx / (GK / 2)) * GSM + 2 * (idx % (GK / 2))]) = ra[r];
      if (B_KMAJOR) *reinterpret_cast<d2 *>(&sB[buf][(idx >> 5) * GSK + 2 * (idx & 31)]) = rb[r];
      else *reinterpret_cast<d2 *>(&sB[buf][(idx / (GK / 2)) * GSM + 2 * (idx % (GK / 2))]) = rb[r];
    }
  };

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  // K range of this tile (whole k-tiles): triangular operands contribute nothing outside it
  int kt0 = 0, nk = g.K / GK;
  if (g.k_from_m) kt0 = m0 / GK;
  if (g.k_from_n && n0 / GK > kt0) kt0 = n0 / GK;
  if (g.k_to_m && (m0 + GT) / GK < nk) nk = (m0 + GT) / GK;
  if (nk > kt0) {
    gload(kt0, ra0, rb0);
    if (kt0 + 1 < nk) gload(kt0 + 1, ra1, rb1);
    sstore(0, ra0, rb0);
  }
  __syncthreads();
  // k-tile kt is in LDS buffer `buf`, k-tile kt + 1 in register stage `nxt`; stage `ld` takes k-tile kt + 2
  auto ktile = [&](int kt, int buf, d2 (&ra_ld)[NST], d2 (&rb_ld)[NST], const d2 (&ra_nxt)[NST], const d2 (&rb_nxt)[NST]) {
    if (kt + 2 < nk) gload(kt + 2, ra_ld, rb_ld);
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      double a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = wm * 32 + mi * 16 + lr, kk = ks * 4 + lk;
        a[mi] = A_KMAJOR ? sA[buf][kk * GSK + m] : sA[buf][m * GSM + kk];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = wn * 32 + ni * 16 + lr, kk = ks * 4 + lk;
        b[ni] = B_KMAJOR ? sB[buf][kk * GSK + n] : sB[buf][n * GSM + kk];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1, ra_nxt, rb_nxt);
    __syncthreads();
  };
  for (int kt = kt0; kt < nk; kt += 2) {
    // even step: tile kt in buffer 0, tile kt + 1 in stage 1, stage 0 is free (its tile kt is already in LDS)
    ktile(kt, 0, ra0, rb0, ra1, rb1);
    if (kt + 1 < nk) ktile(kt + 1, 1, ra1, rb1, ra0, rb0);
  }
  // D[reg] is row (lane >> 4) + 4 * reg, column lane & 15 of each 16 x 16 tile
  double *cp[2][2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cp[mi][ni][r] = C + (int64_t)(m0 + wm * 32 + mi * 16 + lk + 4 * r) * g.ldc + n0 + wn * 32 + ni * 16 + lr;
  if (g.beta == 0.0) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) *cp[mi][ni][r] = g.alpha * acc[mi][ni][r];
  } else {
    // all 16 reads of C in flight together (element by element, each behind its own wait, they cost a trip each)
    double cold[2][2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[mi][ni][r] = *cp[mi][ni][r];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) *cp[mi][ni][r] = fma(g.beta, cold[mi][ni][r], g.alpha * acc[mi][ni][r]);
  }
}

int launch_gemm(const GemmArgs &g, bool a_kmajor, bool b_kmajor, int batch, hipStream_t st) {
  if (g.M % GT || g.N % GT || g.K % GK) {
    set_error("gemm: M, N must be multiples of 64 and K of %d (got %d %d %d)", GK, g.M, g.N, g.K);
    return GPEMU_ERR_ARG;
  }
  if (g.M == 0 || g.N == 0) return GPEMU_OK;
  dim3 grid((unsigned)(g.N / GT), (unsigned)(g.M / GT), (unsigned)batch), block(256);
  if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, block, 0, st, g);
  else if (a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, block, 0, st, g);
  else if (!a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, block, 0, st, g);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
