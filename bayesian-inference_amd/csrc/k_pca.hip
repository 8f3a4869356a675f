// Standardisation + PCA of the design x observable matrix on the device.
//
// Replaces  ref: emulation.py:109-123:
//   StandardScaler().fit_transform(Y)            skl preprocessing/_data.py:1015-1051 (mean_, var_ by the
//                                                corrected two-pass formula of extmath._incremental_mean_and_var,
//                                                rows added in order, so the sums are bit-identical to numpy's)
//   PCA(svd_solver='full').fit_transform(Ys)     skl decomposition/_pca.py:544-702: centre, thin SVD,
//                                                explained_variance_ = S^2 / (N-1), svd_flip (v-based,
//                                                skl utils/extmath.py:944-952), Y_pca = U S
// The SVD is a one-sided Jacobi (Hestenes) iteration on the columns of the smaller dimension:
// pairs of columns are rotated until mutually orthogonal; a round-robin tournament gives n/2
// independent pairs per launch (one workgroup per pair).  Jacobi is chosen over a Gram-matrix
// eigen-decomposition because it does not square the condition number: small singular values (the
// truncated components that make up the "unexplained" covariance) keep full relative accuracy.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "internal.h"

namespace gpemu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Column f, rows r0 .. r0 + UN - 1 of a row-major matrix into registers: the loads of a batch are independent, so they
// are all in flight together, while the sums that use them run strictly in row order (as numpy reduces axis 0).
constexpr int COL_UN = 16;
__device__ __forceinline__ void load_column_batch(const double *__restrict__ Y, int N, int F, int f, int r0, double (&v)[COL_UN]) {
#pragma unroll
  for (int u = 0; u < COL_UN; ++u) v[u] = (r0 + u < N) ? Y[(int64_t)(r0 + u) * F + f] : 0.0;
}

// per column (thread = column, rows accumulated in order, as numpy reduces axis 0)
__global__ void column_stats_kernel(const double *__restrict__ Y, int N, int F, double *__restrict__ mean,
                                    double *__restrict__ var, double *__restrict__ scale) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double v[COL_UN];
  double s = 0.0;
  for (int r0 = 0; r0 < N; r0 += COL_UN) {
    load_column_batch(Y, N, F, f, r0, v);
#pragma unroll
    for (int u = 0; u < COL_UN; ++u)
      if (r0 + u < N) s += v[u];
  }
  const double T = s / N;
  double corr = 0.0, sq = 0.0;
  for (int r0 = 0; r0 < N; r0 += COL_UN) {
    load_column_batch(Y, N, F, f, r0, v);
#pragma unroll
    for (int u = 0; u < COL_UN; ++u)
      if (r0 + u < N) {
        const double t = v[u] - T;
        corr += t;
        sq += t * t;
      }
  }
  const double unnorm = sq - corr * corr / N;
  const double vv = unnorm / N;
  const double eps = 2.220446049250313e-16;
  const double bound = N * eps * vv + (N * T * eps) * (N * T * eps);   // skl _data.py:76-89
  double sc = sqrt(vv);
  if (vv <= bound || sc == 0.0) sc = 1.0;
  mean[f] = T;
  var[f] = vv;
  scale[f] = sc;
}

// Ys = (Y - mean) / scale; then PCA centring: pmean = mean_rows(Ys); Xc = Ys - pmean, written into the
// Jacobi work array G (column-major: column c of the work matrix is contiguous, length m, ld = ldg)
__global__ void standardise_kernel(const double *__restrict__ Y, int N, int F, const double *__restrict__ mean,
                                   const double *__restrict__ scale, double *__restrict__ Ys) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  const int f = (int)(idx % F);
  double v = Y[idx] - mean[f];
  Ys[idx] = v / scale[f];
}

__global__ void pca_centre_kernel(const double *__restrict__ Ys, int N, int F, double *__restrict__ pmean,
                                  double *__restrict__ G, int64_t ldg, int transpose_work, double *__restrict__ sumsq) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double b[COL_UN];
  double s = 0.0;
  for (int r0 = 0; r0 < N; r0 += COL_UN) {
    load_column_batch(Ys, N, F, f, r0, b);
#pragma unroll
    for (int u = 0; u < COL_UN; ++u)
      if (r0 + u < N) s += b[u];
  }
  const double mu = s / N;
  pmean[f] = mu;
  double ss = 0.0;
  for (int r0 = 0; r0 < N; r0 += COL_UN) {
    load_column_batch(Ys, N, F, f, r0, b);
#pragma unroll
    for (int u = 0; u < COL_UN; ++u) {
      const int i = r0 + u;
      if (i < N) {
        const double v = b[u] - mu;
        ss = fma(v, v, ss);
        if (!transpose_work) G[(int64_t)f * ldg + i] = v;   // work column = feature f (length N)
        else G[(int64_t)i * ldg + f] = v;                   // work column = design point i (length F)
      }
    }
  }
  sumsq[f] = ss;                                        // for the squared Frobenius norm of the centred matrix
}

// ---- block one-sided Jacobi ---------------------------------------------------------------------------------
// The work matrix W holds, per column, the m entries of the matrix being orthogonalised followed (from row mpad
// on) by the n entries of the accumulated rotations V; its columns are dealt into blocks of PB = 16 or 32.  One ROUND
// pairs the blocks up (round-robin tournament); for every pair of blocks, i.e. PP = 2 PB columns X:
//   gram   A = X^T X over the first mpad rows: MFMA, the rows split over several workgroups, partial sums to HBM
//   solve  one workgroup adds the partials in a fixed order and runs Jacobi rotations on the PP x PP Gram matrix in
//          LDS (two-sided: A <- R^T A R needs no dot products), accumulating them in J; every rotation also orders
//          the two columns by norm (de Rijk), which speeds up convergence
//   apply  X <- X J for all rows of W (MFMA, computed transposed so that loads and stores follow the columns)
// The rotations are chosen from the Gram matrix but applied to the columns themselves, as orthogonal
// transformations: like the scalar iteration this never squares the condition number of the data, and the
// convergence test (largest |x_i . x_j| / (|x_i| |x_j|) met in a sweep <= sqrt(m) eps) is made on Gram entries
// recomputed from the current columns.  Per sweep the matrix is read and written nb - 1 times instead of n - 1.
constexpr int PCA_GLD = 66;   // LDS leading dimension of the gram tile: a half-wave's 16 x 2 fragment reads hit 32 banks
constexpr int PCA_JLD = 80;   // and of J in the apply kernel

__device__ __forceinline__ void block_pair(int nb, int round, int i, int &P, int &Q) {
  const int nm1 = nb - 1;
  if (i == 0) { P = nm1; Q = round % nm1; }
  else { P = (round + i) % nm1; Q = (round - i + nm1) % nm1; }
  if (P > Q) { const int t = P; P = Q; Q = t; }
}

template <int PB>
__device__ __forceinline__ int64_t pair_column(int P, int Q, int c) { return (c < PB) ? (int64_t)P * PB + c : (int64_t)Q * PB + (c - PB); }

// 1 / x and 1 / sqrt(x) from the hardware seeds (v_rcp_f64 / v_rsq_f64) and two Newton steps: the rotation only needs
// c^2 + s^2 = 1 to rounding, which c = rsqrt(1 + t^2), s = c t gives whatever the last bit of t
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  return fma(y, fma(-x, y, 1.0), y);
}
__device__ __forceinline__ double fast_rsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = fma(y * 0.5, fma(-x * y, y, 1.0), y);
  return fma(y * 0.5, fma(-x * y, y, 1.0), y);
}

// LDS of the Jacobi solve on the PP x PP Gram matrix of one pair of blocks
template <int PB>
struct SolveLds {
  static constexpr int PP = 2 * PB;
  static constexpr int W2 = (PB == 32) ? 8 : 16;   // pairs along a row of 2 x 2 blocks owned by consecutive threads
  static constexpr int LDA = PP + W2;              // a half-wave's blocks then fall into 32 different banks
  static constexpr int LDJ = PP + 1;
  static constexpr int MAXSW = 8;
  double As[PP * LDA];
  double Js[PP * LDJ];
  alignas(32) double rot[2][PB][4];                // per pair of the round: c, s, new squared norms of its columns
  int pq[2][PB][2];                                // its columns p < q
  unsigned long long off;
  int any[MAXSW];
};

// Jacobi rotations on the Gram matrix of one pair of blocks (one workgroup, 256 threads).  full = 1: every pair of
// its columns once per inner sweep (PP - 1 tournament rounds); full = 0: only pairs with one column in either block
// (PB rounds) -- the pairs inside a block are dealt with in the first round of an outer sweep.  One barrier per round:
// every 2 x 2 block of the matrix (rows = one pair, columns = another) has one owner thread, which applies both
// rotations to it; the owner of the element that defines a pair of the NEXT round also derives that rotation, the
// new squared norms it needs being those the current rotations were derived with.
template <int PB>
__device__ void jacobi_solve(SolveLds<PB> &L, const double *__restrict__ src, int nsplit, int full, int inner_sweeps,
                             double tol, double noise2, double *__restrict__ dst, int *__restrict__ flag,
                             unsigned long long *__restrict__ offmax) {
  using S = SolveLds<PB>;
  constexpr int PP = S::PP, W2 = S::W2, LDA = S::LDA, LDJ = S::LDJ, M = PP - 1;
  const int tid = threadIdx.x;
  if (tid == 0) L.off = 0ull;
  if (tid < S::MAXSW) L.any[tid] = 0;
  {
    // all loads of a thread in flight at once (uncached memory: ~2 us a round trip), the sums in split order
    constexpr int NE = PP * PP / 512;           // d2 elements per thread
    d2 v[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) v[u] = reinterpret_cast<const d2 *>(src)[tid + 256 * u];
    for (int sp0 = 1; sp0 < nsplit; sp0 += 4) {
      d2 w[4][NE];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < NE; ++u)
          w[q][u] = (sp0 + q < nsplit) ? reinterpret_cast<const d2 *>(src + (int64_t)(sp0 + q) * (PP * PP))[tid + 256 * u] : d2{0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < NE; ++u)
          if (sp0 + q < nsplit) v[u] += w[q][u];
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int idx = 2 * (tid + 256 * u), i = idx / PP, j = idx % PP;
      L.As[i * LDA + j] = v[u][0]; L.As[i * LDA + j + 1] = v[u][1];
    }
    for (int idx = tid; idx < PP * PP; idx += 256) L.Js[(idx / PP) * LDJ + idx % PP] = (idx / PP == idx % PP) ? 1.0 : 0.0;
  }
  __syncthreads();
  {
    constexpr int TPR = 256 / PP;                // threads per row
    const int i = tid / TPR;
    const double aii = L.As[i * LDA + i];
    double worst = 0.0;
    for (int j = tid % TPR; j < PP; j += TPR) {
      if (j <= i) continue;
      const double ajj = L.As[j * LDA + j];
      const double den2 = aii * ajj;
      const double g = L.As[i * LDA + j];
      if (den2 > 0.0 && g != 0.0 && (aii > noise2 || ajj > noise2)) worst = fmax(worst, g * g / den2);
    }
    for (int off = 32; off > 0; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off));
    if ((tid & 63) == 0) atomicMax(&L.off, (unsigned long long)__double_as_longlong(sqrt(worst)));
  }
  __syncthreads();
  const double worst = __longlong_as_double((long long)L.off);
  if (tid == 0) {
    atomicMax(offmax, L.off);
    *flag = worst > tol ? 1 : 0;
  }
  if (!(worst > tol)) return;
  const long long t_clk = clock64(), t_wall = wall_clock64();
  if (inner_sweeps > S::MAXSW) inner_sweeps = S::MAXSW;
  const int rounds = full ? M : PB;
  const int total = inner_sweeps * rounds;
  const double tol2 = tol * tol;

  // rotation of columns p < q with squared norms a, b and inner product g, into slot idx of buffer buf
  auto derive = [&](int buf, int idx, int sweep, int p, int q, double a, double b, double g) {
    double c = 1.0, s = 0.0, an = a, bn = b;
    if (g * g > tol2 * (a * b) && (a > noise2 || b > noise2)) {
      // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (b - a) / (2 g)
      const double d = b - a, h = 2.0 * g;
      const double x = fma(d, d, h * h);
      const double rt = x * fast_rsq(x);
      double t = fabs(h) * fast_rcp(fabs(d) + rt);
      if ((d < 0.0) != (h < 0.0)) t = -t;
      c = fast_rsq(fma(t, t, 1.0));
      s = c * t;
      an = a - t * g;
      bn = b + t * g;
      if (an < bn) {                             // larger column first: (c, s) <- (s, -c)
        const double cc = c;
        c = s; s = -cc;
        const double tt = an;
        an = bn; bn = tt;
      }
      L.any[sweep] = 1;
    }
    L.rot[buf][idx][0] = c; L.rot[buf][idx][1] = s; L.rot[buf][idx][2] = an; L.rot[buf][idx][3] = bn;
    L.pq[buf][idx][0] = p; L.pq[buf][idx][1] = q;
  };
  // partner of column x in round r, and the slot of the pair (x < y) there
  auto partner = [&](int x, int r) -> int {
    if (!full) return x < PB ? PB + ((x + r) & (PB - 1)) : ((x - PB - r) & (PB - 1));
    if (x == M) return r;
    if (x == r) return M;
    int y = 2 * r - x;
    y %= M;
    return y < 0 ? y + M : y;
  };
  auto slot = [&](int x, int y, int r) -> int {
    if (!full) return x;
    if (y == M) return 0;
    int k = (x - r) % M;
    if (k < 0) k += M;
    return 2 * k <= M ? k : M - k;
  };

  if (tid < PB) {                                // round 0 from the matrix as loaded
    int p, q;
    if (full) {
      if (tid == 0) { p = M; q = 0; }
      else { p = tid % M; q = (M - tid) % M; }
      if (p > q) { const int t = p; p = q; q = t; }
    } else {
      p = tid; q = PB + tid;
    }
    derive(0, slot(p, q, 0), 0, p, q, L.As[p * LDA + p], L.As[q * LDA + q], L.As[p * LDA + q]);
  }
  __syncthreads();
  if (!full) {
    // rounds over pairs (P, PB + ((P + r) mod PB)): every address is known up front, so one LDS round trip brings in the
    // rotations, the thread's 2 x 2 blocks and its entries of J together
    constexpr int NBLK = PB / W2, NJ = PP * PB / 256;
    const int P1 = (tid / W2) % PB;
    int r = 0, sweep = 0;
    for (int rr = 0; rr < total; ++rr) {
      const int buf = rr & 1;
      const bool more = rr + 1 < total;
      const int rn = (r + 1 == PB) ? 0 : r + 1, sweep_n = (r + 1 == PB) ? sweep + 1 : sweep;
      const int p1 = P1, q1 = PB + ((P1 + r) & (PB - 1));
      const d4 r1 = *reinterpret_cast<const d4 *>(&L.rot[buf][P1][0]);
      d4 r2[NBLK];
      double ea[NBLK], eb[NBLK], ec[NBLK], ed[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        const int P2 = (tid % W2) + W2 * j;
        const int p2 = P2, q2 = PB + ((P2 + r) & (PB - 1));
        r2[j] = *reinterpret_cast<const d4 *>(&L.rot[buf][P2][0]);
        ea[j] = L.As[p1 * LDA + p2]; eb[j] = L.As[p1 * LDA + q2];
        ec[j] = L.As[q1 * LDA + p2]; ed[j] = L.As[q1 * LDA + q2];
      }
      d2 rj[NJ];
      double jx[NJ], jy[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int item = tid + 256 * j, Pj = item % PB, row = item / PB;
        rj[j] = *reinterpret_cast<const d2 *>(&L.rot[buf][Pj][0]);
        jx[j] = L.Js[row * LDJ + Pj];
        jy[j] = L.Js[row * LDJ + PB + ((Pj + r) & (PB - 1))];
      }
      const double c1 = r1[0], s1 = r1[1];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        const int P2 = (tid % W2) + W2 * j;
        const int p2 = P2, q2 = PB + ((P2 + r) & (PB - 1));
        const double c2 = r2[j][0], s2 = r2[j][1];
        if (P1 == P2) {
          if (s1 != 0.0) {
            L.As[p1 * LDA + p1] = r1[2]; L.As[q1 * LDA + q1] = r1[3];
            L.As[p1 * LDA + q1] = 0.0; L.As[q1 * LDA + p1] = 0.0;
          }
          continue;
        }
        double a = ea[j], b = eb[j], cc = ec[j], d = ed[j];
        if (s1 != 0.0 || s2 != 0.0) {
          const double a1 = c2 * a - s2 * b, b1 = s2 * a + c2 * b;       // columns p2, q2
          const double c1v = c2 * cc - s2 * d, d1 = s2 * cc + c2 * d;
          a = c1 * a1 - s1 * c1v;                                          // rows p1, q1
          cc = s1 * a1 + c1 * c1v;
          b = c1 * b1 - s1 * d1;
          d = s1 * b1 + c1 * d1;
          L.As[p1 * LDA + p2] = a; L.As[q1 * LDA + p2] = cc;
          L.As[p1 * LDA + q2] = b; L.As[q1 * LDA + q2] = d;
        }
        // the next round pairs column p1 with PB + ((p1 + r + 1) mod PB): column q2 of the block one pair to the right
        if (more && P2 == ((P1 + 1) & (PB - 1)))
          derive(buf ^ 1, p1, sweep_n, p1, PB + ((P1 + rn) & (PB - 1)), r1[2], r2[j][3], b);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {                                       // J <- J R
        const int item = tid + 256 * j, Pj = item % PB, row = item / PB;
        const double cj = rj[j][0], sj = rj[j][1];
        if (sj != 0.0) {
          L.Js[row * LDJ + Pj] = cj * jx[j] - sj * jy[j];
          L.Js[row * LDJ + PB + ((Pj + r) & (PB - 1))] = sj * jx[j] + cj * jy[j];
        }
      }
      __syncthreads();
      if (r + 1 == PB && !L.any[sweep]) break;   // a whole inner sweep without a rotation
      r = rn; sweep = sweep_n;
    }
  } else {
    int r = 0, sweep = 0;
    for (int rr = 0; rr < total; ++rr) {
      const int buf = rr & 1;
      const bool more = rr + 1 < total;
      const int rn = (r + 1 == M) ? 0 : r + 1, sweep_n = (r + 1 == M) ? sweep + 1 : sweep;
      const int P1 = (tid / W2) % PB;
      const int p1 = L.pq[buf][P1][0], q1 = L.pq[buf][P1][1];
      const double c1 = L.rot[buf][P1][0], s1 = L.rot[buf][P1][1];
      const int y_p = partner(p1, rn), y_q = partner(q1, rn);
#pragma unroll
      for (int j = 0; j < PB / W2; ++j) {
        const int P2 = (tid % W2) + W2 * j;
        const int p2 = L.pq[buf][P2][0], q2 = L.pq[buf][P2][1];
        const double c2 = L.rot[buf][P2][0], s2 = L.rot[buf][P2][1];
        if (P1 == P2) {
          if (s1 != 0.0) {
            L.As[p1 * LDA + p1] = L.rot[buf][P1][2]; L.As[q1 * LDA + q1] = L.rot[buf][P1][3];
            L.As[p1 * LDA + q1] = 0.0; L.As[q1 * LDA + p1] = 0.0;
          }
          continue;
        }
        double a = L.As[p1 * LDA + p2], b = L.As[p1 * LDA + q2], cc = L.As[q1 * LDA + p2], d = L.As[q1 * LDA + q2];
        if (s1 != 0.0 || s2 != 0.0) {
          const double a1 = c2 * a - s2 * b, b1 = s2 * a + c2 * b;       // columns p2, q2
          const double c1v = c2 * cc - s2 * d, d1 = s2 * cc + c2 * d;
          a = c1 * a1 - s1 * c1v;                                          // rows p1, q1
          cc = s1 * a1 + c1 * c1v;
          b = c1 * b1 - s1 * d1;
          d = s1 * b1 + c1 * d1;
          L.As[p1 * LDA + p2] = a; L.As[q1 * LDA + p2] = cc;
          L.As[p1 * LDA + q2] = b; L.As[q1 * LDA + q2] = d;
        }
        if (more) {
          // is one of the four elements the inner product of a pair of the next round?  (rows x < columns y only:
          // the mirrored block holds the same element)
          if (y_p > p1 && (y_p == p2 || y_p == q2)) {
            const bool second = y_p == q2;
            derive(buf ^ 1, slot(p1, y_p, rn), sweep_n, p1, y_p, L.rot[buf][P1][2], L.rot[buf][P2][second ? 3 : 2], second ? b : a);
          }
          if (y_q > q1 && (y_q == p2 || y_q == q2)) {
            const bool second = y_q == q2;
            derive(buf ^ 1, slot(q1, y_q, rn), sweep_n, q1, y_q, L.rot[buf][P1][3], L.rot[buf][P2][second ? 3 : 2], second ? d : cc);
          }
        }
      }
      for (int item = tid; item < PP * PB; item += 256) {               // J <- J R
        const int Pj = item % PB, row = item / PB;
        const double cj = L.rot[buf][Pj][0], sj = L.rot[buf][Pj][1];
        if (sj != 0.0) {
          const int pj = L.pq[buf][Pj][0], qj = L.pq[buf][Pj][1];
          const double x = L.Js[row * LDJ + pj], y = L.Js[row * LDJ + qj];
          L.Js[row * LDJ + pj] = cj * x - sj * y;
          L.Js[row * LDJ + qj] = sj * x + cj * y;
        }
      }
      __syncthreads();
      if (r + 1 == M && !L.any[sweep]) break;   // a whole inner sweep without a rotation
      r = rn; sweep = sweep_n;
    }
  }
  if (tid == 0 && blockIdx.x == 0) {               // shader clocks / 100 MHz ticks of the rounds (read with GPEMU_PCA_TRACE)
    offmax[1] += (unsigned long long)(clock64() - t_clk);
    offmax[2] += (unsigned long long)(wall_clock64() - t_wall);
    offmax[3] += 1ull;
  }
  for (int idx = tid; idx < PP * PP; idx += 256) dst[idx] = L.Js[(idx / PP) * LDJ + idx % PP];
}

// Gram matrix of every pair of blocks of the round (rows split over blockIdx.y), then -- by the workgroup of the pair
// that finishes last -- the Jacobi solve on it.  The partial sums are added in their fixed order whoever arrives last.
template <int PB>
__global__ __launch_bounds__(256) void pca_gram_solve_kernel(const double *__restrict__ W, int64_t ldw, int mpad, int nb,
                                                             int round, int nsplit, double *__restrict__ part,
                                                             int *__restrict__ tickets, int inner_sweeps, double tol,
                                                             double noise2, double *__restrict__ Jbuf, int *__restrict__ flags,
                                                             unsigned long long *__restrict__ offmax) {
  constexpr int PP = 2 * PB, NT = PP / 16;      // columns of the pair, 16 x 16 tiles per side
  constexpr int TPC = 256 / PP, RPT = 64 / TPC; // threads per column of the 64-row chunk, rows per thread
  __shared__ union U {
    double T[PP * PCA_GLD];
    SolveLds<PB> solve;
    __device__ U() {}
  } lds;
  __shared__ int s_last;
  double *T = lds.T;
  const long long t_start = wall_clock64();
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int P, Q;
  block_pair(nb, round, blockIdx.x, P, Q);
  const int chunks = mpad / 64;
  const int c0 = (int)((int64_t)chunks * blockIdx.y / nsplit), c1 = (int)((int64_t)chunks * (blockIdx.y + 1) / nsplit);
  d4 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
  const int lcol = tid / TPC, lrow = (tid % TPC) * RPT;
  const double *src = W + pair_column<PB>(P, Q, lcol) * ldw + lrow;
  d2 stage[RPT / 2];
  if (c0 < c1) {
#pragma unroll
    for (int v = 0; v < RPT / 2; ++v) stage[v] = *reinterpret_cast<const d2 *>(src + (int64_t)c0 * 64 + 2 * v);
  }
  for (int c = c0; c < c1; ++c) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < RPT / 2; ++v) *reinterpret_cast<d2 *>(&T[lcol * PCA_GLD + lrow + 2 * v]) = stage[v];
    __syncthreads();
    if (c + 1 < c1) {
#pragma unroll
      for (int v = 0; v < RPT / 2; ++v) stage[v] = *reinterpret_cast<const d2 *>(src + (int64_t)(c + 1) * 64 + 2 * v);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double f[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) f[a] = T[(16 * a + lr) * PCA_GLD + 16 * wave + 4 * ks + lk];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[a], f[b], acc[a][b], 0, 0, 0);
    }
  }
  // the four waves hold the sums over their quarter of every chunk: added in wave order through LDS
  __syncthreads();
  double *S = T;   // PP x PP, dense
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = (16 * a + lk + 4 * r) * PP + 16 * b + lr;
            S[idx] = (w == 0) ? acc[a][b][r] : S[idx] + acc[a][b][r];
          }
    }
    __syncthreads();
  }
  double *dst = part + ((int64_t)blockIdx.x * nsplit + blockIdx.y) * (PP * PP);
  for (int idx = tid; idx < PP * PP / 2; idx += 256) reinterpret_cast<d2 *>(dst)[idx] = reinterpret_cast<const d2 *>(S)[idx];
  // The last workgroup of the pair to get here solves.  `part` is uncached device memory: a store that has been
  // acknowledged (vmcnt) is in memory and a load fetches from memory, so the hand-over needs no L2 write-back /
  // invalidate (an agent-scope fence pair costs ~10 us here), only the order store -> ticket -> load.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == nsplit - 1);
    if (s_last) __hip_atomic_store(&tickets[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next round
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0 && blockIdx.x == 0) offmax[4] += (unsigned long long)(wall_clock64() - t_start);
  const long long t_solve = wall_clock64();
  jacobi_solve<PB>(lds.solve, part + (int64_t)blockIdx.x * nsplit * (PP * PP), nsplit, round == 0 ? 1 : 0, inner_sweeps, tol,
                   noise2, Jbuf + (int64_t)blockIdx.x * (PP * PP), flags + blockIdx.x, offmax);
  if (tid == 0 && blockIdx.x == 0) { offmax[5] += (unsigned long long)(wall_clock64() - t_solve); offmax[6] += 1ull; }
}

// X <- X J for the PP columns of every pair of blocks that rotated, rows in strips of 16 (one wave per strip):
// out^T (new column x row) = J^T (new column x old column) X^T (old column x row)
template <int PB>
__global__ __launch_bounds__(256) void pca_apply_kernel(double *__restrict__ W, int64_t ldw, int nstrips, int nb, int round,
                                                        const double *__restrict__ Jbuf, const int *__restrict__ flags) {
  constexpr int PP = 2 * PB;
  __shared__ double Js[PP * PCA_JLD];
  if (!flags[blockIdx.x]) return;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int P, Q;
  block_pair(nb, round, blockIdx.x, P, Q);
  const double *J = Jbuf + (int64_t)blockIdx.x * (PP * PP);
  for (int idx = tid; idx < PP * PP; idx += 256) Js[(idx / PP) * PCA_JLD + (idx % PP)] = J[idx];
  __syncthreads();
  for (int strip = blockIdx.y * 4 + wave; strip < nstrips; strip += gridDim.y * 4) {
    const int64_t r0 = (int64_t)strip * 16 + lr;
    double bf[PP / 4];
#pragma unroll
    for (int kk = 0; kk < PP / 4; ++kk) bf[kk] = W[pair_column<PB>(P, Q, 4 * kk + lk) * ldw + r0];
#pragma unroll
    for (int mt = 0; mt < PP / 16; ++mt) {
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < PP / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Js[(4 * kk + lk) * PCA_JLD + 16 * mt + lr], bf[kk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) W[pair_column<PB>(P, Q, 16 * mt + lk + 4 * r) * ldw + r0] = acc[r];
    }
  }
}

template <int PB>
static void jacobi_round(double *dW, int64_t ldw, int mpad, int nstrips, int nb, int r, int nsplit, int inner, double tol,
                         double noise2, double *dpart, int *dtickets, double *dJ, int *dflags, unsigned long long *doff,
                         int apply_y) {
  const unsigned npairs = (unsigned)(nb / 2);
  hipLaunchKernelGGL(pca_gram_solve_kernel<PB>, dim3(npairs, (unsigned)nsplit), dim3(256), 0, nullptr, dW, ldw, mpad, nb, r,
                     nsplit, dpart, dtickets, inner, tol, noise2, dJ, dflags, doff);
  hipLaunchKernelGGL(pca_apply_kernel<PB>, dim3(npairs, (unsigned)apply_y), dim3(256), 0, nullptr, dW, ldw, nstrips, nb, r, dJ,
                     dflags);
}

// W rows [mpad, mpad + n): identity
__global__ void pca_identity_rows_kernel(double *W, int64_t ldw, int mpad, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) W[(int64_t)j * ldw + mpad + j] = 1.0;
}

}  // namespace gpemu

using namespace gpemu;

extern "C" int gpemu_pca_fit(int device, int64_t N, int64_t F, const double *Y, int64_t n_components,
                             double *scaler_mean, double *scaler_scale, double *scaler_var, double *pca_mean,
                             double *components, double *explained_variance, double *explained_variance_ratio,
                             double *Y_pca, int64_t *flip_argmax, int64_t *n_sweeps) {
  GP_ARG(Y && scaler_mean && scaler_scale && scaler_var && pca_mean && components && explained_variance &&
             explained_variance_ratio && Y_pca, "null pointer");
  GP_ARG(N >= 2 && F >= 1, "need N >= 2 rows and F >= 1 columns");
  const int64_t nmin = std::min(N, F);
  const int64_t nc = (n_components <= 0) ? nmin : n_components;
  GP_ARG(nc <= nmin, "n_components must be <= min(N, F)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device available: libgpemu has no CPU implementation");
    return GPEMU_ERR_NO_DEVICE;
  }
  GP_ARG(device >= 0 && device < ndev, "device");
  GP_HIP(hipSetDevice(device));
  const bool tw = N < F;                          // work on the transpose when there are fewer rows
  const int m = (int)(tw ? F : N), n = (int)(tw ? N : F);
  const int PB = n <= 1024 ? 16 : 32;                                     // columns per block (more, smaller pairs keep a small matrix's rounds short)
  const int PP = 2 * PB;
  const int nb = std::max(2, (int)(round_up(n, 2 * PB) / PB));      // blocks of PB columns, an even number of them
  const int ncols = nb * PB, npairs = nb / 2;
  const int mpad = (int)round_up(m, 64);
  const int64_t ldw = (round_up(mpad + n, 32)) | 16;                // odd multiple of 16: columns start on different channels
  const int nstrips = (int)((mpad + round_up(n, 16)) / 16);
  const int chunks = mpad / 64;
  const int nsplit = std::max(1, std::min(chunks, std::min(8, 256 / npairs)));
  const int inner = PB == 16 ? 1 : 2;                                     // inner sweeps per encounter of two blocks
  const bool trace = getenv("GPEMU_PCA_TRACE") != nullptr;
  double *dY = nullptr, *dYs = nullptr, *dmean = nullptr, *dvar = nullptr, *dscale = nullptr, *dpm = nullptr,
         *dW = nullptr, *dpart = nullptr, *dJ = nullptr, *dss = nullptr;
  int *dflags = nullptr, *dtickets = nullptr;
  unsigned long long *doff = nullptr;
  hipError_t e = hipMalloc((void **)&dY, sizeof(double) * N * F);
  auto A = [&](double **p, int64_t cnt) { if (e == hipSuccess) e = hipMalloc((void **)p, sizeof(double) * (size_t)cnt); };
  A(&dYs, N * F); A(&dmean, F); A(&dvar, F); A(&dscale, F); A(&dpm, F); A(&dW, (int64_t)ncols * ldw);
  if (e == hipSuccess) e = hipExtMallocWithFlags((void **)&dpart, sizeof(double) * (size_t)npairs * nsplit * PP * PP, hipDeviceMallocUncached);
  A(&dJ, (int64_t)npairs * PP * PP); A(&dss, F);
  if (e == hipSuccess) e = hipMalloc((void **)&dflags, sizeof(int) * npairs);
  if (e == hipSuccess) e = hipMalloc((void **)&dtickets, sizeof(int) * npairs);
  if (e == hipSuccess) e = hipMemset(dtickets, 0, sizeof(int) * npairs);
  if (e == hipSuccess) e = hipMalloc((void **)&doff, 8 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(doff, 0, 8 * sizeof(unsigned long long));
  std::vector<double> hW;
  int sweeps = 0;
  if (e == hipSuccess) e = hipMemcpy(dY, Y, sizeof(double) * N * F, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(column_stats_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, nullptr, dY, (int)N, (int)F,
                       dmean, dvar, dscale);
    hipLaunchKernelGGL(standardise_kernel, dim3((unsigned)((N * F + 255) / 256)), dim3(256), 0, nullptr, dY, (int)N,
                       (int)F, dmean, dscale, dYs);
    e = hipMemsetAsync(dW, 0, sizeof(double) * (size_t)ncols * ldw, nullptr);
    hipLaunchKernelGGL(pca_centre_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, nullptr, dYs, (int)N, (int)F,
                       dpm, dW, ldw, tw ? 1 : 0, dss);
    hipLaunchKernelGGL(pca_identity_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, dW, ldw, mpad, n);
    const double tol = 4.0 * std::sqrt((double)m) * 2.220446049250313e-16;   // the cosines recomputed from rounded Gram sums bottom out near sqrt(m) eps
    const int apply_y = std::max(1, std::min((nstrips + 3) / 4, 2048 / npairs));
    // Columns shorter than noise_c sqrt(m) eps ||X||_F are rounding residue of the rotations (the matrix is numerically
    // rank deficient whenever the observables are smooth functions of a few parameters): rotating two of them against
    // each other changes nothing measurable -- singular values below 1e-13 of the largest, whose vectors stay orthonormal
    // in V either way -- but such pairs would otherwise dominate the sweep count.
    double noise2 = 0.0;
    {
      std::vector<double> hss((size_t)F);
      if (e == hipSuccess) e = hipMemcpy(hss.data(), dss, sizeof(double) * F, hipMemcpyDeviceToHost);
      double fro2 = 0.0;
      for (double v : hss) fro2 += v;
      const double noise_c = 4.0;
      const double lim = noise_c * std::sqrt((double)m) * 2.220446049250313e-16;
      noise2 = lim * lim * fro2;
    }
    for (sweeps = 0; sweeps < 60 && e == hipSuccess; ++sweeps) {
      e = hipMemsetAsync(doff, 0, sizeof(unsigned long long), nullptr);
      for (int r = 0; r < nb - 1; ++r) {
        if (PB == 16) jacobi_round<16>(dW, ldw, mpad, nstrips, nb, r, nsplit, inner, tol, noise2, dpart, dtickets, dJ, dflags, doff, apply_y);
        else jacobi_round<32>(dW, ldw, mpad, nstrips, nb, r, nsplit, inner, tol, noise2, dpart, dtickets, dJ, dflags, doff, apply_y);
      }
      unsigned long long bits = 0, dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (e == hipSuccess) e = hipMemcpy(dbg, doff, sizeof(dbg), hipMemcpyDeviceToHost);
      bits = dbg[0];
      if (trace && dbg[3])
        fprintf(stderr, "pca_fit:   pair 0 so far: %llu solves, %.0f shader clocks, %.2f us each (%.2f GHz)\n", dbg[3],
                (double)dbg[1] / dbg[3], (double)dbg[2] / dbg[3] * 0.01, dbg[2] ? (double)dbg[1] / dbg[2] * 0.1 : 0.0);
      if (trace && dbg[6])
        fprintf(stderr, "pca_fit:   pair 0: %llu rounds, gram + ticket %.2f us, solve (load, test, rotations, store) %.2f us\n", dbg[6],
                (double)dbg[4] / dbg[6] * 0.01, (double)dbg[5] / dbg[6] * 0.01);
      double off;
      std::memcpy(&off, &bits, sizeof(off));
      if (trace) fprintf(stderr, "pca_fit: sweep %d  largest cosine %.3e  (tol %.3e)\n", sweeps, off, tol);
      if (off <= tol) { ++sweeps; break; }
    }
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (e == hipSuccess) {
    hW.resize((size_t)ncols * ldw);
    e = hipMemcpy(hW.data(), dW, sizeof(double) * hW.size(), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(scaler_mean, dmean, sizeof(double) * F, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(scaler_var, dvar, sizeof(double) * F, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(scaler_scale, dscale, sizeof(double) * F, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(pca_mean, dpm, sizeof(double) * F, hipMemcpyDeviceToHost);
  }
  (void)hipFree(dY); (void)hipFree(dYs); (void)hipFree(dmean); (void)hipFree(dvar); (void)hipFree(dscale);
  (void)hipFree(dpm); (void)hipFree(dW); (void)hipFree(dpart); (void)hipFree(dJ); (void)hipFree(dflags); (void)hipFree(dtickets); (void)hipFree(doff); (void)hipFree(dss);
  if (e != hipSuccess) { set_error("pca_fit: %s", hipGetErrorString(e)); return GPEMU_ERR_HIP; }
  if (n_sweeps) *n_sweeps = sweeps;

  // ---- host assembly: order by singular value, normalise, sign convention -----------------------------
  // column j of the work matrix: hW[j * ldw + r], r < m; its accumulated rotations: hW[j * ldw + mpad + r], r < n.
  // The n columns that carry data are those with a non-zero rotation part (the padding columns stay zero).
  std::vector<double> sigma((size_t)ncols);
  std::vector<int> order;
  for (int j = 0; j < ncols; ++j) {
    const double *g = &hW[(size_t)j * ldw];
    double s2 = 0.0;
    for (int r = 0; r < m; ++r) s2 += g[r] * g[r];
    sigma[j] = std::sqrt(s2);
    bool real = false;
    for (int r = 0; r < n && !real; ++r) real = g[mpad + r] != 0.0;
    if (real) order.push_back(j);
  }
  if ((int)order.size() != n) { set_error("pca_fit: internal error (%d of %d columns carry data)", (int)order.size(), n); return GPEMU_ERR_ARG; }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sigma[a] > sigma[b]; });
  double total_var = 0.0;
  for (int c = 0; c < n; ++c) total_var += sigma[order[(size_t)c]] * sigma[order[(size_t)c]] / (double)(N - 1);
  std::vector<double> comp((size_t)F);
  const char *flip_env = getenv("GPEMU_SVD_FLIP");          // read per call
  const bool flip_u = flip_env && (flip_env[0] == 'u' || flip_env[0] == 'U');
  for (int64_t c = 0; c < nc; ++c) {
    const int j = order[(size_t)c];
    const double sg = sigma[j];
    const double *g = &hW[(size_t)j * ldw], *v = g + mpad;
    // component row (length F): right singular vector of Xc
    if (!tw) for (int64_t f = 0; f < F; ++f) comp[f] = v[f];
    else for (int64_t f = 0; f < F; ++f) comp[f] = sg > 0.0 ? g[f] / sg : 0.0;
    // svd_flip.  Default: the v-based decision of scikit-learn >= 1.5 (skl utils/extmath.py:944-952: per row of V^T the
    // sign of its max-|.| entry, first index on ties) -- the version the goldens were made with.  GPEMU_SVD_FLIP=u: the
    // u-based decision of the scikit-learn the reference pins (ref: pdm.lock:1998-1999 -> 1.3.0, PCA._fit_full calls
    // svd_flip(U, Vt) with u_based_decision=True): per COLUMN of U.  U S is the data part of the work column (its
    // rotation part when the transpose was factored), S > 0, so the deciding row of U is that of U S.
    int64_t arg = 0;
    double best = -1.0, at_arg = 0.0;
    if (flip_u) {
      const double *us = tw ? v : g;
      for (int64_t i = 0; i < N; ++i)
        if (std::fabs(us[i]) > best) { best = std::fabs(us[i]); arg = i; }
      at_arg = us[arg];
    } else {
      for (int64_t f = 0; f < F; ++f)
        if (std::fabs(comp[f]) > best) { best = std::fabs(comp[f]); arg = f; }
      at_arg = comp[arg];
    }
    const double sign = at_arg < 0.0 ? -1.0 : 1.0;
    for (int64_t f = 0; f < F; ++f) components[c * F + f] = sign * comp[f];
    if (flip_argmax) flip_argmax[c] = arg;
    explained_variance[c] = sg * sg / (double)(N - 1);
    explained_variance_ratio[c] = explained_variance[c] / total_var;
    // Y_pca[:, c] = U[:, c] * S[c]
    if (!tw) for (int64_t i = 0; i < N; ++i) Y_pca[i * nc + c] = sign * g[i];
    else for (int64_t i = 0; i < N; ++i) Y_pca[i * nc + c] = sign * sg * v[i];
  }
  return GPEMU_OK;
}
