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
constexpr int GSM = GK + 2; // LDS stride of an m-major tile [T][18]

// TW = MFMA tiles per side of a wave's part of C: 2 -> 64 x 64 workgroup tile (32 x 32 per wave), 4 -> 128 x 128
// (64 x 64 per wave: 8 LDS fragment reads feed 16 MFMAs instead of 4 feeding 4, and a tile moves half the operand bytes
// per FLOP -- the 64 x 64 tile needs 1 byte per 8 FLOP from L2 / HBM, which bounds it near 30 TF once the operands of a
// batch stop fitting in cache).  The arithmetic per element of C is the same either way (the k-steps of an MFMA chain
// run in the same order; the extra k-tiles a 128-row tile covers for half of its rows under a triangular K range
// multiply exact zeros), so both give the same bits.  TW = 4: M, N multiples of 64 -- the last tile row / column may
// be half a tile -- and under `lower_only` the upper-right quarter of a diagonal tile is neither computed nor stored.
template <bool A_KMAJOR, bool B_KMAJOR, int TW>
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgs g) {
  constexpr int T = 32 * TW;                 // workgroup tile edge
  constexpr int SK = T + 16;                 // LDS stride of a k-major tile [GK][T + 16]: (2 * SK) % 64 == 32
  constexpr int TSZ = (GK * SK > T * GSM) ? GK * SK : T * GSM;
  __shared__ __attribute__((aligned(16))) double sA[2][TSZ];
  __shared__ __attribute__((aligned(16))) double sB[2][TSZ];
  // Batch entry and tile of this workgroup.  Workgroups go to the 8 XCDs round-robin in launch order (x fastest, z
  // slowest) and every XCD has its own 4 MiB L2: in launch order each XCD works on every 8th tile of the ~4 problems of a
  // batch that are in flight and so streams all of their operands (64 problems of N = 1000: K^-1 = W^T W ran at 0.22 of
  // peak, HBM-bound).  With `xcd_batch` the launch positions are re-dealt so that XCD x takes whole problems
  // x, x + 8, ...: a problem's operands are read into one L2 only.  (A bijection of the grid when the number of
  // problems is a multiple of 8; launch_gemm sets the flag only then.)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.xcd_batch) {
    const int gx = gridDim.x, gy = gridDim.y, b1 = g.batch1 > 0 ? g.batch1 : 1;
    const int per_problem = gx * gy * b1;
    const int lin = (bz * gy + by) * gx + bx;
    const int xcd = lin & 7, j = lin >> 3;
    const int problem = (j / per_problem) * 8 + xcd;
    int rest = j % per_problem;
    bx = rest % gx; rest /= gx;
    by = rest % gy; rest /= gy;
    bz = rest + b1 * problem;
  }
  if (g.packed) {
    // lower_only on a square tile grid launched as a 1-D grid over the tiles on and under the diagonal only: position t
    // is tile (m, n), t = m (m + 1) / 2 + n -- no workgroups that start only to find themselves above the diagonal
    int m = (int)((sqrt(8.0 * (double)bx + 1.0) - 1.0) * 0.5);
    while (m * (m + 1) / 2 > bx) --m;
    while ((m + 1) * (m + 2) / 2 <= bx) ++m;
    by = m;
    bx -= m * (m + 1) / 2;
  }
  const int z1 = g.batch1 > 0 ? (int)(bz % g.batch1) : bz;
  const int z2 = g.batch1 > 0 ? (int)(bz / g.batch1) : 0;
  const double *A = g.A + (int64_t)z1 * g.strideA + (int64_t)z2 * g.stride2A;
  const double *B = g.B + (int64_t)z1 * g.strideB + (int64_t)z2 * g.stride2B;
  double *C = g.C + (int64_t)z1 * g.strideC + (int64_t)z2 * g.stride2C;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // Triangular operands make the K range of a tile depend on its column (k_from_n) or row (k_to_m): with about as many
  // tiles as the chip holds workgroups, the launch lasts as long as the longest tile while the average one has half its
  // k-steps (measured: 0.40 of peak against 0.77 for the dense product).  `pair` folds the tile grid: a workgroup takes
  // tile t and then its mirror image T - 1 - t along that dimension, which together always have the same K range.
  const int ntx = g.N / T, nty = g.M / T;
  for (int rep = 0; rep < (g.pair ? 2 : 1); ++rep) {
  int ty = by, tx = bx;
  if (g.pair == 1 && rep == 1) { tx = ntx - 1 - tx; if (tx == bx) break; }
  if (g.pair == 2 && rep == 1) { ty = nty - 1 - ty; if (ty == by) break; }
  const int m0 = ty * T, n0 = tx * T;
  if (g.lower_only && n0 > m0) continue;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  // a wave's part of C outside the matrix (half tile at the edge) or above the diagonal of a SYRK tile: nothing to do
  const bool dead = (m0 + wm * (T / 2) >= g.M) || (n0 + wn * (T / 2) >= g.N) ||
                    (TW == 4 && g.lower_only && m0 == n0 && wm == 0 && wn == 1);

  // staging: an operand tile is T x GK doubles; NST 16-byte pieces per thread
  //   k-major source [k][m]: GK rows of T doubles;  m-major source [m][k]: T rows of GK doubles
  constexpr int NST = T * GK / 2 / 256;
  // two register stages: the loads of k-tile t + 2 are issued while k-tile t is multiplied and k-tile t + 1 waits in the
  // other stage for its turn to be written to LDS -- two k-tiles of compute cover a load's trip
  d2 ra0[NST], rb0[NST], ra1[NST], rb1[NST];
  auto gload = [&](int kt, d2 (&ra)[NST], d2 (&rb)[NST]) {
#pragma unroll
    for (int r = 0; r < NST; ++r) {
      const int idx = tid + 256 * r;
      if (A_KMAJOR) {
        const int mm = m0 + 2 * (idx % (T / 2));
        ra[r] = mm < g.M ? *reinterpret_cast<const d2 *>(A + (int64_t)(kt * GK + idx / (T / 2)) * g.lda + mm) : d2{0.0, 0.0};
      } else {
        const int mm = m0 + idx / (GK / 2);
        ra[r] = mm < g.M ? *reinterpret_cast<const d2 *>(A + (int64_t)mm * g.lda + kt * GK + 2 * (idx % (GK / 2))) : d2{0.0, 0.0};
      }
      if (B_KMAJOR) {
        const int nn = n0 + 2 * (idx % (T / 2));
        rb[r] = nn < g.N ? *reinterpret_cast<const d2 *>(B + (int64_t)(kt * GK + idx / (T / 2)) * g.ldb + nn) : d2{0.0, 0.0};
      } else {
        const int nn = n0 + idx / (GK / 2);
        rb[r] = nn < g.N ? *reinterpret_cast<const d2 *>(B + (int64_t)nn * g.ldb + kt * GK + 2 * (idx % (GK / 2))) : d2{0.0, 0.0};
      }
    }
  };
  auto sstore = [&](int buf, const d2 (&ra)[NST], const d2 (&rb)[NST]) {
#pragma unroll
    for (int r = 0; r < NST; ++r) {
      const int idx = tid + 256 * r;
      if (A_KMAJOR) *reinterpret_cast<d2 *>(&sA[buf][(idx / (T / 2)) * SK + 2 * (idx % (T / 2))]) = ra[r];
      else *reinterpret_cast<d2 *>(&sA[buf][(idx / (GK / 2)) * GSM + 2 * (idx % (GK / 2))]) = ra[r];
      if (B_KMAJOR) *reinterpret_cast<d2 *>(&sB[buf][(idx / (T / 2)) * SK + 2 * (idx % (T / 2))]) = rb[r];
      else *reinterpret_cast<d2 *>(&sB[buf][(idx / (GK / 2)) * GSM + 2 * (idx % (GK / 2))]) = rb[r];
    }
  };

  d4 acc[TW][TW];
#pragma unroll
  for (int mi = 0; mi < TW; ++mi)
#pragma unroll
    for (int ni = 0; ni < TW; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  // K range of this tile (whole k-tiles): triangular operands contribute nothing outside it
  int kt0 = 0, nk = g.K / GK;
  if (g.k_from_m) kt0 = m0 / GK;
  if (g.k_from_n && n0 / GK > kt0) kt0 = n0 / GK;
  if (g.k_to_m && (m0 + T) / GK < nk) nk = (m0 + T) / GK;
  if (nk > kt0) {
    gload(kt0, ra0, rb0);
    if (kt0 + 1 < nk) gload(kt0 + 1, ra1, rb1);
    sstore(0, ra0, rb0);
  }
  __syncthreads();
  // k-tile kt is in LDS buffer `buf`, k-tile kt + 1 in register stage `nxt`; stage `ld` takes k-tile kt + 2
  auto ktile = [&](int kt, int buf, d2 (&ra_ld)[NST], d2 (&rb_ld)[NST], const d2 (&ra_nxt)[NST], const d2 (&rb_nxt)[NST]) {
    if (kt + 2 < nk) gload(kt + 2, ra_ld, rb_ld);
    if (!dead) {
#pragma unroll
      for (int ks = 0; ks < GK / 4; ++ks) {
        double a[TW], b[TW];
        const int kk = ks * 4 + lk;
#pragma unroll
        for (int mi = 0; mi < TW; ++mi) {
          const int m = wm * (T / 2) + mi * 16 + lr;
          a[mi] = A_KMAJOR ? sA[buf][kk * SK + m] : sA[buf][m * GSM + kk];
        }
#pragma unroll
        for (int ni = 0; ni < TW; ++ni) {
          const int n = wn * (T / 2) + ni * 16 + lr;
          b[ni] = B_KMAJOR ? sB[buf][kk * SK + n] : sB[buf][n * GSM + kk];
        }
#pragma unroll
        for (int mi = 0; mi < TW; ++mi)
#pragma unroll
          for (int ni = 0; ni < TW; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) sstore(buf ^ 1, ra_nxt, rb_nxt);
    __syncthreads();
  };
  for (int kt = kt0; kt < nk; kt += 2) {
    // even step: tile kt in buffer 0, tile kt + 1 in stage 1, stage 0 is free (its tile kt is already in LDS)
    ktile(kt, 0, ra0, rb0, ra1, rb1);
    if (kt + 1 < nk) ktile(kt + 1, 1, ra1, rb1, ra0, rb0);
  }
  if (dead) continue;
  // D[reg] is row (lane >> 4) + 4 * reg, column lane & 15 of each 16 x 16 tile
  auto cptr = [&](int mi, int ni, int r) {
    return C + (int64_t)(m0 + wm * (T / 2) + mi * 16 + lk + 4 * r) * g.ldc + n0 + wn * (T / 2) + ni * 16 + lr;
  };
  if (g.beta == 0.0) {
#pragma unroll
    for (int mi = 0; mi < TW; ++mi)
#pragma unroll
      for (int ni = 0; ni < TW; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) *cptr(mi, ni, r) = g.alpha * acc[mi][ni][r];
  } else {
    // the reads of C of a row tile in flight together (element by element, each behind its own wait, they cost a trip each)
#pragma unroll
    for (int mi = 0; mi < TW; ++mi) {
      double cold[TW][4];
#pragma unroll
      for (int ni = 0; ni < TW; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[ni][r] = *cptr(mi, ni, r);
#pragma unroll
      for (int ni = 0; ni < TW; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) *cptr(mi, ni, r) = fma(g.beta, cold[ni][r], g.alpha * acc[mi][ni][r]);
    }
  }
  }   // rep
}

int launch_gemm(const GemmArgs &g, bool a_kmajor, bool b_kmajor, int batch, hipStream_t st) {
  if (g.M % GT || g.N % GT || g.K % GK) {
    set_error("gemm: M, N must be multiples of 64 and K of %d (got %d %d %d)", GK, g.M, g.N, g.K);
    return GPEMU_ERR_ARG;
  }
  if (g.M == 0 || g.N == 0) return GPEMU_OK;
  // (128 x 128 tiles -- the kernel's R = 4 form, still instantiated by tools/gemm_probe -- lose at every size: with one
  // or two workgroups per CU (74 KiB of LDS, ~200 VGPRs) the big tile hides its LDS and global latencies worse than four
  // co-resident 64 x 64 workgroups do: batched N = 1000 5.2 vs 4.25 ms per 64 problems, N = 5000 7.8 vs 6.4 ms)
  constexpr bool big = false;
  GemmArgs gs = g;
  // fold the tile grid where a triangular operand makes the K range run along one dimension only (see the kernel)
  gs.pair = 0;
  // ... when there are enough tiles to fill the chip at least half (fewer: a launch lasts as long as its longest tile
  // either way, and folding only halves the workgroups -- b = 256 pairs batched: 17 -> 22 us)
  const int64_t tiles = (int64_t)(g.M / GT) * (g.N / GT) * batch;
  if (!big && !g.lower_only && tiles >= 400) {
    if (g.k_from_n && !g.k_from_m && !g.k_to_m && g.N >= 128) gs.pair = 1;
    else if (g.k_to_m && !g.k_from_n && !g.k_from_m && g.M >= 128) gs.pair = 2;
  }
  const int T = big ? 128 : 64;
  {
    const int problems = g.batch1 > 0 ? batch / g.batch1 : batch;
    gs.xcd_batch = (problems >= 8 && problems % 8 == 0 && (g.batch1 <= 0 || batch % g.batch1 == 0)) ? 1 : 0;
  }
  dim3 grid((unsigned)((g.N + T - 1) / T), (unsigned)((g.M + T - 1) / T), (unsigned)batch), block(256);
  if (gs.pair == 1) grid.x = (grid.x + 1) / 2;
  if (gs.pair == 2) grid.y = (grid.y + 1) / 2;
  gs.packed = (g.lower_only && !big && g.M == g.N) ? 1 : 0;
  if (gs.packed) { grid.x = grid.x * (grid.x + 1) / 2; grid.y = 1; }
#define GP_LAUNCH_GEMM(AK, BK)                                                                  \
  do {                                                                                          \
    hipLaunchKernelGGL((gemm_f64_kernel<AK, BK, 2>), grid, block, 0, st, gs);                   \
  } while (0)
  if (a_kmajor && b_kmajor) GP_LAUNCH_GEMM(true, true);
  else if (a_kmajor && !b_kmajor) GP_LAUNCH_GEMM(true, false);
  else if (!a_kmajor && b_kmajor) GP_LAUNCH_GEMM(false, true);
  else GP_LAUNCH_GEMM(false, false);
#undef GP_LAUNCH_GEMM
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
