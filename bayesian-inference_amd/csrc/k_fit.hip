// Fit side of the GP emulator: kernel matrix, blocked Cholesky, triangular inverse, log-marginal
// likelihood and its gradient.
//
// Replaces the arithmetic behind  ref: emulation.py:169-172  (GaussianProcessRegressor.fit):
//   skl _gpr.py:580      K, dK = kernel_(X_train, eval_gradient=True)        -> kmat_kernel, grad kernel
//   skl _gpr.py:585-587  K[diag] += alpha ; L = cholesky(K, lower)           -> blocked right-looking
//   skl _gpr.py:597      alpha = cho_solve(L, y)                             -> W = L^-1, alpha = W^T (W y)
//   skl _gpr.py:609-613  lml = -1/2 y.alpha - sum log diag L - N/2 log 2 pi
//   skl _gpr.py:625-647  grad_i = 1/2 sum_jl (alpha alpha^T - K^-1)_jl dK_jl/dtheta_i,  K^-1 = W^T W
// The optimiser itself (L-BFGS-B with restarts, skl _gpr.py:299-337) stays on the host and calls
// gpemu_fit_lml once per evaluation.
//
// Blocked Cholesky, panel width 64 (matrix padded to a multiple of 64 with an identity tail):
//   potrf_diag_kernel   factor the 64 x 64 diagonal block in LDS and invert it      (one workgroup)
//   GEMM                panel = A21 . inv(L11)^T                                     (MFMA f64)
//   GEMM (SYRK)         A22 -= panel . panel^T, lower tiles only                     (MFMA f64)
// Triangular inverse W = L^-1 by block rows with two MFMA GEMMs per block row; the inverted
// diagonal blocks come from the Cholesky step.
#include <algorithm>
#include <atomic>
#include <cmath>

#include "gemm.h"
#include "internal.h"

namespace gpemu {

constexpr int NB = 64;
typedef double d4t __attribute__((ext_vector_type(4)));

// ---- kernel matrix ---------------------------------------------------------------------------
__device__ __forceinline__ double base_from_r2(int kind, double r2) {
  if (kind == 0) return exp(-0.5 * r2);
  double r = sqrt(r2);
  if (kind == 1) return exp(-r);
  if (kind == 2) {
    double t = r * 1.7320508075688772;
    return (1.0 + t) * exp(-t);
  }
  double t = r * 2.23606797749979;
  return (1.0 + t + t * t / 3.0) * exp(-t);
}

// X [Np][DPAD] raw inputs; hp = {ls[DPAD], const, noise}; K[i][j] for i,j < N, identity tail
// blockIdx.z: problem of a batch (own hyper-parameters and matrix, shared inputs)
constexpr int KMAT_ROWS = 16;     // rows of K per workgroup: the scaled coordinates of column j are formed once for all of them
__global__ __launch_bounds__(256) void kmat_kernel(const double *__restrict__ X, const double *__restrict__ hp,
                                                   double *__restrict__ K, int N, int Np, int kind, double jitter) {
  __shared__ double s_xi[KMAT_ROWS][DPAD];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = blockIdx.y * KMAT_ROWS;
  hp += (int64_t)blockIdx.z * (DPAD + 2);
  K += (int64_t)blockIdx.z * Np * Np;
  // skl: X / length_scale, then the difference -- the quotients are the same numbers whoever forms them
  if (threadIdx.x < KMAT_ROWS * DPAD) {
    const int r = threadIdx.x / DPAD, dd = threadIdx.x % DPAD;
    s_xi[r][dd] = (i0 + r < N) ? X[(i0 + r) * DPAD + dd] / hp[dd] : 0.0;
  }
  double xj[DPAD];
#pragma unroll
  for (int dd = 0; dd < DPAD; ++dd) xj[dd] = (j < N) ? X[j * DPAD + dd] / hp[dd] : 0.0;
  const double cst = hp[DPAD], diag = 1.0 + hp[DPAD] + hp[DPAD + 1] + jitter;   // np.fill_diagonal(K, 1) + const + noise + alpha
  __syncthreads();
  if (j >= Np) return;
  for (int r = 0; r < KMAT_ROWS; ++r) {
    const int i = i0 + r;
    if (i >= Np) break;
    double v;
    if (i >= N || j >= N) {
      v = (i == j) ? 1.0 : 0.0;
    } else if (i == j) {
      v = diag;
    } else {
      double r2 = 0.0;
#pragma unroll
      for (int dd = 0; dd < DPAD; ++dd) {
        const double df = s_xi[r][dd] - xj[dd];
        r2 = fma(df, df, r2);
      }
      v = base_from_r2(kind, r2) + cst;
    }
    K[(int64_t)i * Np + j] = v;
  }
}

// ---- 64 x 64 diagonal block: Cholesky + inverse ------------------------------------------------
// 256 threads, block in LDS, both phases blocked by 16 so that only 4 x 16 pivots are serial:
//   factor   per 16-column panel a right-looking column sweep by three waves (below: the 16 x 16 diagonal block, the
//            rows under it, the diagonal block's inverse), then the rank-16 update of the trailing lower triangle in
//            LDS on the matrix cores;
//   inverse  the four 16 x 16 diagonal blocks come out of the sweeps (of a GIVEN factor, do_factor = 0: by forward
//            substitution, one wave each, lane = column), then two levels of  X21 = -X22 (L21 X11)  as small LDS
//            matrix products on all threads.
#ifdef GPEMU_POTRF_STAMPS      // tools/potrf_probe.hip: cycle stamps of the phases of one launch
__device__ long long g_potrf_stamps[16];
__device__ double g_potrf_flags[64];
#define POTRF_STAMP(i) do { if (threadIdx.x == 0) g_potrf_stamps[i] = clock64(); } while (0)
#define POTRF_WAVE_STAMP(i) do { if ((threadIdx.x & 63) == 0) g_potrf_stamps[i] = clock64(); } while (0)
__device__ long long g_panel_stamps[128][16];     // [row of the panel][event]: wall_clock64 (100 MHz, one clock for all XCDs)
#define PANEL_STAMP(k) do { if (threadIdx.x == 0 && q < 128) g_panel_stamps[q][k] = wall_clock64(); } while (0)
#else
#define POTRF_STAMP(i) do { } while (0)
#define POTRF_WAVE_STAMP(i) do { } while (0)
#define PANEL_STAMP(k) do { } while (0)
#endif
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// ---- the 16-column sweep of a panel as a pipeline of three waves -----------------------------------------------
// Until round 4 ONE wave swept a panel: lane = row (the 16 rows of the diagonal block and the rows below it), pivot by
// pivot, every L[c][j] fetched with two v_readlane into an SGPR pair -- 750 instructions per sweep, 5 480 ticks, four
// sweeps = half of a 64 x 64 block's time on the serial path of the factorisation.  ONE wave issues an instruction every
// ~6 ticks, whatever it is, an s_nop included (tools/dp_latency_probe; in the sweep: replacing the three instructions of
// an update by two DPP ones changed nothing), so the instructions are dealt to three waves (three SIMDs) instead:
//   wave 0  factors the 16 x 16 diagonal block alone.  Lane & 15 = row, the block replicated in every row of 16 lanes,
//           so that L[c][j] is lane c of the own row: ONE v_fmac_f64_dpp (row_newbcast is legal for 64-bit operations
//           on gfx90a+) per update and no SGPR.  The nine dependent steps of the next pivot (broadcast, 1 / sqrt as
//           hardware estimate + two Newton steps, scaling) are written between the updates of the current one -- a wave
//           issues in order.  Each finished column goes to LDS at once (a contiguous 16 x 16 column buffer), then its
//           1 / sqrt as the column's flag.
//   wave 1  owns the rows below (lane = row) and follows column by column: r[j] *= rinv_j, r[c] -= r[j] L[c][j] with
//           L[c][j] a broadcast LDS read.
//   wave 2  inverts the diagonal block as the columns arrive (lane = column of the inverse, forward substitution) -- the
//           separate pass over the four diagonal blocks is gone.
//   wave 3  copies the PREVIOUS panel's diagonal block from its column buffer into the block and applies the part of the
//           previous panel's rank-16 update that this sweep does not read.
// A flag slot holds a NaN with a payload no arithmetic produces until its value is there (LDS executes a wave's
// instructions in order: column first, flag second; volatile accesses keep the compiler to that order).  Wave 0 waits
// for nobody, so the followers' waits end; they are bounded all the same.
// The arithmetic of the factor -- fma(-L[c][j], L[i][j], .) for j ascending, the scaled column -- and of the inverse
// (the correctly rounded 1 / L[j][j], the forward substitution's FMAs in row order) is that of the one-wave form: same
// bits (tools/chol_bits.py).
constexpr unsigned long long SWEEP_PENDING = 0x7ff8dead0000beefULL;
// the volatile accesses name the LDS address space themselves (address-space inference leaves volatile ones generic:
// a flat store with system scope and a wait behind it)
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

template <int L>
__device__ __forceinline__ double row16_bcast(double v) {        // v of lane L of the own row of 16, after an s_nop 1
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
  return r;
}
// The hazard recogniser does not look into inline assembly (a DPP read of a VGPR needs two wait states after the VALU
// write of it): the statements are volatile -- kept in program order -- and carry an s_nop where the order does not
// give the distance.
template <int L>
__device__ __forceinline__ void fmac_neg_bcast(double &acc, double src, double mul) {   // acc -= src[lane L of the row] * mul
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(L));
}

// updates number U0 .. U1-1 of pivot column J: d[c] -= L[c][J] d[J] for the columns c = J + 2 + U
template <int J, int U0, int U1>
__device__ __forceinline__ void sweep_updates(double (&d)[16]) {
  if constexpr (U0 < U1) {
    fmac_neg_bcast<J + 2 + U0>(d[J + 2 + U0], d[J], d[J]);
    sweep_updates<J, U0 + 1, U1>(d);
  }
}

// column J of the diagonal block is final (d[J], lanes >= J of a row; zeros before): to LDS, then its flag.  Every lane
// stores, at a constant offset from a per-lane base: lanes 0 .. 15 their element of the column (the zeros above the
// diagonal on zeros), lane 0 the flag, everybody else into dump slots -- no address arithmetic between the pivots.
struct SweepOut {
  lds_f64 *col;            // lt + lane (lanes < 16): element `lane` of column J goes to lt[16 J + lane] / dump slots
  lds_f64 *flag;           // the panel's 16 flags (lane 0) / 16 dump slots
};
template <int J>
__device__ __forceinline__ void sweep_publish(const SweepOut &o, double v, double rinv) {
  *(volatile lds_f64 *)(o.col + 16 * J) = v;
  *(volatile lds_f64 *)(o.flag + J) = rinv;
}

// Pivot J + 1 of the sweep while the updates of pivot column J (scaled already) are issued: column J + 1 is updated
// first, then the steps of its pivot go out one by one with some of pivot J's remaining updates behind each.  The wave
// issues one instruction per ~6 ticks whatever it is, an s_nop included, so the wait states the hazards ask for (two
// between the write of a VGPR and a DPP read of it, one behind the transcendental) are made of updates while there are
// any.
template <int J>
__device__ __forceinline__ void sweep_pivot(double (&d)[16], const SweepOut &o) {
  constexpr int NU = (J + 2 <= 15) ? 14 - J : 0;      // updates of pivot J to the columns J + 2 .. 15
  constexpr int NA = NU < 2 ? NU : 2;                 // between the update of column J + 1 and its broadcast
  constexpr int NB1 = NU > NA ? 1 : 0;                // between the estimate and its first use
  constexpr int NR = NU - NA - NB1;                   // the rest: seven gaps
#define GAP(g) sweep_updates<J, NA + NB1 + (g) * NR / 7, NA + NB1 + ((g) + 1) * NR / 7>(d)
  if constexpr (J + 1 < 16) {
    double piv, y, a, h;
    fmac_neg_bcast<J + 1>(d[J + 1], d[J], d[J]);
    sweep_updates<J, 0, NA>(d);
    if constexpr (NA == 0) asm volatile("s_nop 1");
    if constexpr (NA == 1) asm volatile("s_nop 0");
    // (v_rsq_f64_dpp assembles, and the compiler's DPP combiner emits it, but on this part it returns NaN: the
    // transcendental unit does not take the DPP operand -- measured; broadcast and estimate stay two instructions)
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(piv) : "v"(d[J + 1]), "n"(J + 1));
    asm volatile("v_rsq_f64 %0, %1" : "=v"(y) : "v"(piv));
    sweep_updates<J, NA, NA + NB1>(d);
    if constexpr (NB1 == 0) asm volatile("s_nop 0");
    asm volatile("v_mul_f64 %0, %1, -%2" : "=v"(a) : "v"(y), "v"(piv));
    asm volatile("v_mul_f64 %0, %1, 0.5" : "=v"(h) : "v"(y));
    GAP(0);
    asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(a) : "v"(y));
    GAP(1);
    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(y) : "v"(h), "v"(a));
    GAP(2);
    asm volatile("v_mul_f64 %0, %1, -%2" : "=v"(a) : "v"(y), "v"(piv));
    asm volatile("v_mul_f64 %0, %1, 0.5" : "=v"(h) : "v"(y));
    GAP(3);
    asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(a) : "v"(y));
    GAP(4);
    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(y) : "v"(h), "v"(a));
    GAP(5);
    asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[J + 1]) : "v"(y));       // lanes >= J + 1 of a row: column J + 1 of L
    sweep_publish<J + 1>(o, d[J + 1], y);     // (two LDS stores: the wait states before the DPP reads of the scaled column)
    GAP(6);
    sweep_pivot<J + 1>(d, o);
  }
#undef GAP
}

// a[N0 .. N1) are computed -- in registers -- here: the compiler may not sink their producers behind a later wait (it
// left all the arithmetic of a follower to the end)
template <int N0, int N1>
__device__ __forceinline__ void pin_values(double (&a)[16]) {
  if constexpr (N1 - N0 >= 4) {
    asm volatile("" : "+v"(a[N0]), "+v"(a[N0 + 1]), "+v"(a[N0 + 2]), "+v"(a[N0 + 3]));
    pin_values<N0 + 4, N1>(a);
  } else if constexpr (N1 - N0 >= 1) {
    asm volatile("" : "+v"(a[N0]));
    pin_values<N0 + 1, N1>(a);
  }
}

// A follower's request for column J: the flag first, then the broadcast reads of L[J+1 .. 15][J] right behind it, all in
// flight together.  LDS serves a wave's reads in order: if the flag read finds the value, the column read behind it finds
// the column.  Issued one column AHEAD (the reads of column J + 1 travel while column J is applied), so a follower that
// keeps up never sees the latency of LDS; one that finds the flag pending asks again.
template <int J, bool DIAG>
__device__ __forceinline__ void follow_fetch(const double *lt, const double *flags, unsigned long long &fl, double (&l)[16]) {
  if constexpr (J < 16) {
    fl = *(volatile lds_u64 *)(lds_u64 *)(flags + J);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = J + (DIAG ? 0 : 1); c < 16; ++c) l[c] = lt[16 * J + c];      // uniform addresses, consecutive: broadcast reads
    asm volatile("" ::: "memory");
  }
}

// column J of a follower: x[J] = (INV: the forward substitution's value, zero above the diagonal of the inverse)
// x[J] rinv_J, then x[c] -= x[J] L[c][J] for the later c
template <int J, bool INV>
__device__ __forceinline__ void follow_step(const double *D, const double *flags, double (&x)[16], int c,
                                            unsigned long long fl, double (&l)[16], int &expired) {
  if constexpr (J < 16) {
    int polls = 0;
    while (fl == SWEEP_PENDING && ++polls < (1 << 22)) follow_fetch<J, INV>(D, flags, fl, l);     // bounded: wave 0 waits for nobody
    // (an expired wait cannot happen while wave 0 runs; were it to, the last column's would expire too: reported as
    // info = -1, never silently)
    if constexpr (J == 15) { if (fl == SWEEP_PENDING) expired = 1; }
    // rows below: scaled by the sweep's own 1 / sqrt(pivot) (the flag's value r0).  Inverse: by the reciprocal of the
    // diagonal element L_jj = RN(pivot r0), which the stand-alone pass (tile_inverse_diag16) and the one-wave form take
    // from an IEEE division -- ~25 instructions on this wave's path per column (measured: the wave ends 900 ticks
    // behind wave 0).  r0 is within a few ulp of 1 / L_jj, so ONE Newton step with fused residual,
    // RN(r0 + r0 RN(1 - L_jj r0)), has a relative error of ~2^-100 before its final rounding: the correctly rounded
    // reciprocal unless 1 / L_jj lies within 2^-100 of a rounding boundary (a ~2^-47 chance per element; tools/chol_bits.py:
    // the digests of factors, LML and gradients equal those of the division form).
    const double r0 = __longlong_as_double((long long)fl);
    const double rinv = INV ? fma(fma(-l[J], r0, 1.0), r0, r0) : r0;
    unsigned long long fl2 = 0;
    double l2[16];
    follow_fetch<J + 1, INV>(D, flags, fl2, l2);
    const double xj = (!INV || J >= c) ? x[J] * rinv : 0.0;
    x[J] = xj;
#pragma unroll
    for (int k = J + 1; k < 16; ++k) x[k] = fma(-xj, l[k], x[k]);
    pin_values<J + 1, 16>(x);
    follow_step<J + 1, INV>(D, flags, x, c, fl2, l2, expired);
  }
}

// tile (ti, tj) of the trailing block [T0, 64)^2 takes the rank-16 update of the panel at column J0: four k-steps; a
// diagonal tile writes its lower part only (the column sweep relies on zeros above)
__device__ __forceinline__ void trailing_tile(double (*D)[NB + 1], int J0, int T0, int ti, int tj, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  d4t acc = d4t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[T0 + 16 * ti + lr][J0 + 4 * ks + lk], D[T0 + 16 * tj + lr][J0 + 4 * ks + lk], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lk + 4 * r, col = lr;
    if (ti != tj || col <= row) D[T0 + 16 * ti + row][T0 + 16 * tj + col] -= acc[r];
  }
}

// One 16-column panel of the 64 x 64 block (columns J0 .. J0+15): the sweep (above; the diagonal block's inverse goes
// to X, flags = 16 slots holding SWEEP_PENDING), then all threads apply the rank-16 update to the trailing lower
// triangle.  J0 is a run-time value and the four panels of a block run through ONE copy of this code (tile_factor's
// loop is not unrolled): fully unrolled per panel the block's factorisation was ~75 KB of straight-line code, every
// instruction of it executed once -- more than the 64 KB instruction cache, i.e. fetched from L2 by every block.
__device__ __forceinline__ void panel_step(double (*D)[NB + 1], double (*X)[NB + 1], double *flags, double *dump, double *lt,
                                           const double *lt_prev, int J0, int tid, int blk, int *info) {
  constexpr int PB = 16;
  const int MB = NB - J0 - PB;             // rows below the diagonal block
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double (*Dp)[NB + 1] = (double (*)[NB + 1])&D[J0][J0];       // the panel's corner
  if (wave == 0) {
    double dg[PB];
#pragma unroll
    for (int c = 0; c < PB; ++c) dg[c] = Dp[lane & 15][c];
    SweepOut o;
    o.col = (lds_f64 *)(lane < PB ? lt + lane : dump + lane);             // (a dump slot per lane: no bank conflicts)
    o.flag = (lds_f64 *)(lane == 0 ? flags : dump + 320 + lane);
    {
      const double piv = row16_bcast<0>(dg[0]);
      double y = __builtin_amdgcn_rsq(piv);
      y = fma(0.5 * y, fma(-piv * y, y, 1.0), y);
      y = fma(0.5 * y, fma(-piv * y, y, 1.0), y);
      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dg[0]) : "v"(y));
      sweep_publish<0>(o, dg[0], y);
      asm volatile("s_nop 1" : "+v"(dg[0]));
    }
    sweep_pivot<0>(dg, o);
    // a pivot that is not positive leaves a 1 / sqrt that is not a positive number (NaN: the estimate of a negative
    // number, or the Newton step on the infinite estimate of zero) -- looked for here, off the pivots' path
    const double rj = flags[lane & 15];
    const unsigned long long notpos = __ballot(!(rj > 0.0)) & 0xffffull;
    if (notpos && lane == 0 && info && *info == 0) *info = blk * NB + J0 + __builtin_ctzll(notpos) + 1;
    if (J0 == 0) POTRF_WAVE_STAMP(10);
  } else if (wave == 1) {
    if (MB > 0) {
      const bool act = lane < MB;
      double row[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) row[c] = act ? Dp[PB + (act ? lane : 0)][c] : 0.0;
      unsigned long long fl;
      double l[PB];
      follow_fetch<0, false>(lt, flags, fl, l);
      int expired = 0;
      follow_step<0, false>(lt, flags, row, 0, fl, l, expired);
      if (expired && info) __hip_atomic_store(info, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (act) {
#pragma unroll
        for (int c = 0; c < PB; ++c) Dp[PB + lane][c] = row[c];
      }
      if (J0 == 0) POTRF_WAVE_STAMP(11);
    }
  } else if (wave == 2) {
    // inverse of the diagonal block, lane = column: forward substitution, column by column of L (right-looking): once
    // x[mm] is known every later row takes its term -- independent FMAs; the serial path is one multiply and one FMA
    // per row
    if (lane < PB) {
      const int c = lane;
      double acc[PB];
#pragma unroll
      for (int ii = 0; ii < PB; ++ii) acc[ii] = (ii == c) ? 1.0 : 0.0;
      unsigned long long fl;
      double l[PB];
      follow_fetch<0, true>(lt, flags, fl, l);
      int expired = 0;
      follow_step<0, true>(lt, flags, acc, c, fl, l, expired);
      if (expired && info) __hip_atomic_store(info, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int ii = 0; ii < PB; ++ii) X[J0 + ii][J0 + c] = acc[ii];
    }
    if (J0 == 0) POTRF_WAVE_STAMP(12);
  } else if (J0 > 0) {
    // the previous panel's diagonal block from its column buffer into D (nobody reads it there before the merges)
#pragma unroll
    for (int u = 0; u < 4; ++u) D[J0 - PB + (lane & 15)][J0 - PB + 4 * u + (lane >> 4)] = lt_prev[16 * (4 * u + (lane >> 4)) + (lane & 15)];
    // the part of the previous panel's rank-16 update that this sweep does not read: the tiles right of its first column
    const int Tp = J0, Mp = NB - Tp;
    for (int ti = 1; ti < Mp / 16; ++ti)
      for (int tj = 1; tj <= ti; ++tj) trailing_tile(D, J0 - PB, Tp, ti, tj, lane);
  }
  __syncthreads();
  if (J0 == 0) POTRF_STAMP(8);
  // Rank-16 update of the trailing lower triangle [T0, 64)^2 on the matrix cores, 16 x 16 tiles.  Here only its first
  // tile column -- the next panel, one tile per wave; the rest is left to wave 3, which is idle during the next sweep
  // (above).  A tile still takes the panels' updates in their order.
  const int T0 = J0 + PB, M = NB - T0;
  if (M > 0) {
    if (wave < M / 16) trailing_tile(D, J0, T0, wave, 0, lane);
    __syncthreads();
  }
}

// X21 = -X22 (L21 X11) for every pair of adjacent B x B diagonal blocks of the 64 x 64 factor D / inverse X
template <int B>
__device__ __forceinline__ void merge_level(double (*D)[NB + 1], double (*X)[NB + 1], double (*T)[32 + 1], int tid) {
  constexpr int NPAIR = NB / (2 * B), NT = B / 16;      // 16 x 16 output tiles per side of a pair
  const int lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // T = L21 . X11 on the matrix cores: NPAIR * NT * NT tiles over the four waves
  {
    int t = 0;
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          if ((t++ & 3) != wv) continue;
          const int p0 = pr * 2 * B;
          d4t acc = d4t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < B / 4; ++ks)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[p0 + B + 16 * ti + lr][p0 + 4 * ks + lk],
                                                       X[p0 + 4 * ks + lk][p0 + 16 * tj + lr], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) T[(pr * B + 16 * ti + lk + 4 * r) & 31][16 * tj + lr] = acc[r];
        }
  }
  __syncthreads();
  // X21 = -X22 . T
  {
    int t = 0;
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          if ((t++ & 3) != wv) continue;
          const int p0 = pr * 2 * B;
          d4t acc = d4t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < B / 4; ++ks)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[p0 + B + 16 * ti + lr][p0 + B + 4 * ks + lk],
                                                       T[(pr * B + 4 * ks + lk) & 31][16 * tj + lr], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) X[p0 + B + 16 * ti + lk + 4 * r][p0 + 16 * tj + lr] = -acc[r];
        }
  }
  __syncthreads();
}


// Cholesky factor of the 64 x 64 block in D (lower triangle, zeros above), in place, and the inverses of its four
// 16 x 16 diagonal blocks in X (which must hold zeros on entry); T: scratch (sweep flags); 256 threads
__device__ __forceinline__ void tile_factor(double (*D)[NB + 1], double (*X)[NB + 1], double (*T)[32 + 1], int tid, int blk, int *info) {
  // T: the sweeps' flags [0, 64), two column buffers of 16 x 16 (a finished column of the diagonal block is stored
  // CONTIGUOUSLY for the followers -- one base address and constant offsets instead of a row stride per element: waves
  // 1 / 2 done at 4123 / 4417 -> 4130 / 4113 ticks of a sweep; the block reaches D one sweep later), dump slots from 576
  double *flags = &T[0][0], *ltb = &T[0][0] + NB, *dump = &T[0][0] + 576;
  if (tid < NB) ((unsigned long long *)flags)[tid] = SWEEP_PENDING;
  __syncthreads();
#pragma nounroll
  for (int J0 = 0; J0 < NB; J0 += 16) {        // one copy of the code for the four panels (see panel_step)
    panel_step(D, X, flags + J0, dump, ltb + 256 * ((J0 >> 4) & 1), ltb + 256 * (((J0 >> 4) & 1) ^ 1), J0, tid, blk, info);
    if (J0 == 0) POTRF_STAMP(2);
  }
  D[NB - 16 + (tid & 15)][NB - 16 + (tid >> 4)] = ltb[256 + 16 * (tid >> 4) + (tid & 15)];      // the last diagonal block
  __syncthreads();
#ifdef GPEMU_POTRF_STAMPS
  if (tid < 64) g_potrf_flags[tid] = flags[tid];
#endif
}

// the inverses of the four 16 x 16 diagonal blocks of a given factor D (the factor path gets them from the sweep)
__device__ __forceinline__ void tile_inverse_diag16(double (*D)[NB + 1], double (*X)[NB + 1], double (*T)[32 + 1], int tid) {
  constexpr int PB = 16;
  const int lane = tid & 63, wave = tid >> 6;
  // wave w, lane = column.  Forward substitution as in the sweep's wave 2.  The reciprocals of the diagonal: one
  // division per lane, handed round through the scratch T (same wave: in order).
  if (lane < PB) {
    const int b0 = wave * PB, c = lane;
    double *rdiag = &T[0][0] + b0;
    rdiag[c] = 1.0 / D[b0 + c][b0 + c];
    double acc[PB], x[PB];
#pragma unroll
    for (int ii = 0; ii < PB; ++ii) acc[ii] = (ii == c) ? 1.0 : 0.0;
#pragma unroll
    for (int mm = 0; mm < PB; ++mm) {
      x[mm] = (mm >= c) ? acc[mm] * rdiag[mm] : 0.0;
#pragma unroll
      for (int ii = mm + 1; ii < PB; ++ii) acc[ii] = fma(-D[b0 + ii][b0 + mm], x[mm], acc[ii]);   // uniform address: broadcast
    }
#pragma unroll
    for (int ii = 0; ii < PB; ++ii) X[b0 + ii][b0 + c] = x[ii];
  }
  __syncthreads();
}

// X = D^-1 for the lower-triangular 64 x 64 factor D, given the inverses of the four 16 x 16 diagonal blocks in X (zeros
// elsewhere); T is the scratch of the merges
__device__ __forceinline__ void tile_inverse(double (*D)[NB + 1], double (*X)[NB + 1], double (*T)[32 + 1], int tid) {
  POTRF_STAMP(5);
  // merge pairs of inverted diagonal blocks of size b into blocks of size 2b (b = 16, then 32); the loops run
  // over the full b (X holds zeros above the diagonal, so the triangular structure needs no bounds) and are
  // unrolled, which lets the LDS reads of one output pipeline instead of waiting on a data-dependent trip count
  merge_level<16>(D, X, T, tid);
  POTRF_STAMP(9);
  merge_level<32>(D, X, T, tid);
}

// blockIdx.x: diagonal block (64 apart), blockIdx.y: problem of a batch (strides batchA, batchD; info per problem)
__global__ __launch_bounds__(256) void potrf_diag_kernel(double *A, int64_t lda, double *Dinv, int do_factor,
                                                         int block_index, int *info, int64_t batchA, int64_t batchD) {
  POTRF_STAMP(0);
  A += (int64_t)blockIdx.y * batchA;
  Dinv += (int64_t)blockIdx.y * batchD;
  info += blockIdx.y;
  __shared__ double D[NB][NB + 1];     // the factor
  __shared__ double X[NB][NB + 1];     // its inverse
  __shared__ double T[32][32 + 1];     // L21 X11 of the merge in flight
  const int tid = threadIdx.x;
  double *Ab = A + ((int64_t)blockIdx.x * NB) * lda + (int64_t)blockIdx.x * NB;
  double *Db = Dinv + (int64_t)blockIdx.x * NB * NB;
  {
    // all 16 loads of a thread in flight together (the block was written by the previous launch on other CUs: every
    // load is a trip to memory; issued one by one between the LDS stores they cost 4.7 us)
    double v[NB * NB / 256];
#pragma unroll
    for (int u = 0; u < NB * NB / 256; ++u) {
      const int idx = tid + 256 * u, r = idx >> 6, c = idx & 63;
      v[u] = (c <= r) ? Ab[(int64_t)r * lda + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NB * NB / 256; ++u) {
      const int idx = tid + 256 * u, r = idx >> 6, c = idx & 63;
      D[r][c] = v[u];
      X[r][c] = 0.0;
    }
  }
  __syncthreads();
  POTRF_STAMP(1);
  if (do_factor) {
    tile_factor(D, X, T, tid, block_index + (int)blockIdx.x, info);
    POTRF_STAMP(3);
    for (int idx = tid; idx < NB * NB; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      Ab[(int64_t)r * lda + c] = D[r][c];    // zeros above the diagonal
    }
  }
  POTRF_STAMP(4);
  if (!do_factor) tile_inverse_diag16(D, X, T, tid);
  tile_inverse(D, X, T, tid);
  POTRF_STAMP(6);
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    Db[idx] = X[r][c];
  }
  POTRF_STAMP(7);
}

// ---- one launch per 256-wide panel ------------------------------------------------------------------
// The four 64-wide steps of a panel (diagonal factor, solve of the rows below, update of the rest of the panel) used
// to be three dependent launches each -- 12 launch boundaries and as many round trips through memory on the serial
// path of the factorisation.  Here one workgroup owns one 64-row strip of the panel [rows rb*64 .., columns of the
// panel] for the whole panel, keeps it in registers (MFMA accumulator layout) and walks the steps s = 0 .. 3:
//     wait for Dinv_s              (published by the workgroup of diagonal row s: "head" s)
//     L_rs = A_rs Dinv_s^T         -> memory (in place)
//     A_rc -= L_rs L_cs^T          for the later columns c of the panel, L_cs published by head c
// and a head, when it arrives at its own diagonal block, factors and inverts it in LDS and publishes the inverse.
// Publication is a release store of the panel's tag to a flag, consumption an acquire load (agent scope: the strips
// live on different XCDs).  Progress: a workgroup only ever waits for heads, heads only for heads of smaller row index.
// With the heads at workgroups 0 .. 3 of their problem whoever is waited for was DISPATCHED before the waiter (a 1-D
// grid is dispatched in index order: observed behaviour of this part, not a HIP guarantee).  With the one-XCD placement
// (heads_one_xcd: heads at workgroups 0, 8, 16, 24, see the kernel) that is no longer so: workgroups 1 .. 23 of a problem
// may sit on a CU polling for a head that has not been dispatched yet.  The invariant is then a CAPACITY one: at any
// time at most 24 workgroups per concurrent panel launch can be blocked on an undispatched head (those of the problem
// at the launch's dispatch frontier; every earlier problem has all its heads on the chip and drains by itself), so the
// frontier always finds a slot as long as the device holds more than 24 x (concurrent panel launches) workgroups of
// this kernel at once.  The host enables the placement only with a margin on that: resident workgroups (runtime
// occupancy x CUs) >= 64 x (fit handles alive in this process), chol_heads_placement(); otherwise heads stay first.
// Every wait is bounded all the same (info = -1 on expiry, ~1 s) so that a fault -- or a device shared with something
// this rule does not see -- cannot hang the GPU: the evaluation fails with GPEMU_ERR_STATE and the handle stays usable.
// The arithmetic (operand order of every MFMA chain, C - acc for the updates) is that of the three-launch path: the
// factor has the same bits, whichever placement.
constexpr int CHOL_Q = 4;                       // blocks per panel
constexpr int CHOL_FLAGS = 4 + 4 * 4;           // per problem: D ready [4], L_cs ready [c][s]
constexpr int CHOL_WAIT_POLLS = 1 << 20;       // ~1 s

// Publication protocol.  The published blocks (Dinv_s, L_cs: 32 KiB each) are written and read with agent-scope
// relaxed atomics -- stores that go through to the coherence point, loads that do not trust this XCD's L2 -- instead
// of ordinary accesses bracketed by agent-scope release / acquire fences: those fences write back, respectively
// invalidate, the WHOLE L2 of the XCD, which the ~10 strips per XCD keep full of their own freshly written blocks
// (measured: a head's step 18 -> 23 us with 79 strips in flight).  The flag follows when every thread's stores have
// been acknowledged (workgroup-scope release + barrier).
__device__ __forceinline__ void publish_store(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double published_load(const double *p) {
  return __hip_atomic_load(const_cast<double *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool HEAD>
__device__ __forceinline__ void chol_wait(int *flag, int tag, int *info) {
  if (threadIdx.x == 0) {
    int polls = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      __builtin_amdgcn_s_sleep(HEAD ? 4 : 12);
      // give up: once anywhere in this problem, then everybody at once (the results are void, the launch must end)
      if (++polls > CHOL_WAIT_POLLS || ((polls & 1023) == 0 && __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
        __hip_atomic_store(info, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void chol_post(int *flag, int tag) {
  // this thread's publish_stores are acknowledged (gfx9 counts stores in vmcnt; a workgroup-scope release fence does not
  // wait for them: the waves of a workgroup share their L1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out = rows [16 wave, 16 wave + 16) of U V^T (64 x 64 x 64), four 16 x 16 tiles per wave, k ascending.
// V_LOWER: V is lower triangular (an inverted diagonal block): column tile nt only has terms k < 16 (nt + 1) -- the
// k-steps beyond would add exact zeros and are skipped (40 of 64 MFMAs remain; same bits).
template <bool V_LOWER>
__device__ __forceinline__ void strip_product(const double (*U)[NB + 1], const double (*V)[NB + 1], int wave, int lane,
                                              d4t (&out)[4]) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) out[nt] = d4t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < NB / 4; ++ks) {
    const double a = U[16 * wave + lr][4 * ks + lk];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (!V_LOWER || ks < 4 * (nt + 1))
        out[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, V[16 * nt + lr][4 * ks + lk], out[nt], 0, 0, 0);
  }
}

// grid (rows of the panel = nblk - jb0, problems); npb = blocks of this panel (<= 4)
// fault != 0 (tests only, GPEMU_CHOL_FAULT): the second head of the first panel never publishes its inverse -- the waits
// for it must expire and the evaluation end with GPEMU_ERR_STATE.
__global__ __launch_bounds__(256) void chol_panel_kernel(double *A, int64_t Np, double *Dinv, int jb0, int npb, int *info,
                                                         int *flags, int tag, int64_t batchA, int64_t batchD, int fault,
                                                         int heads_one_xcd) {
  A += (int64_t)blockIdx.y * batchA;
  Dinv += (int64_t)blockIdx.y * batchD;
  info += blockIdx.y;
  flags += (int64_t)blockIdx.y * CHOL_FLAGS;
  __shared__ double U[NB][NB + 1];     // this strip's current column block (rows wave-private) / the factor of a head
  __shared__ double V[NB][NB + 1];     // Dinv_s or L_cs / the inverse of a head
  __shared__ double T[32][32 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  // Row of this workgroup.  Workgroups go to the XCDs round-robin in launch order; with heads_one_xcd (host: at least 32
  // rows and room on the device, see the progress argument above) the first 32 are dealt so that the four heads (rows
  // 0 .. 3) are workgroups 0, 8, 16, 24 -- all on one XCD: a head takes Dinv and L of the head before it from that XCD's L2
  // instead of across the fabric.
  const int bx = blockIdx.x;
  const int q = (heads_one_xcd && gridDim.x >= 32 && bx < 32) ? (bx & 7) * 4 + (bx >> 3) : bx, rb = jb0 + q;
  PANEL_STAMP(0);
  const bool head = q < npb;
  const int ncol = head ? q + 1 : npb;           // column blocks of the panel on or under the diagonal in this row
  double *Arow = A + ((int64_t)rb * NB) * Np + (int64_t)jb0 * NB;
  // acc[c][nt][reg]: row 16 wave + lk + 4 reg, column 64 c + 16 nt + lr of the strip
  d4t acc[CHOL_Q][4];
#pragma unroll
  for (int c = 0; c < CHOL_Q; ++c)
    if (c < ncol) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[c][nt][r] = Arow[(int64_t)(16 * wave + lk + 4 * r) * Np + 64 * c + 16 * nt + lr];
    }
  auto load_V = [&](const double *src, int64_t ld) {     // a 64 x 64 block from memory, all loads of a thread in flight
    double v[NB * NB / 256];
#pragma unroll
    for (int u = 0; u < NB * NB / 256; ++u) {
      const int idx = tid + 256 * u;
      v[u] = published_load(src + (int64_t)(idx >> 6) * ld + (idx & 63));
    }
#pragma unroll
    for (int u = 0; u < NB * NB / 256; ++u) {
      const int idx = tid + 256 * u;
      V[idx >> 6][idx & 63] = v[u];
    }
  };
  PANEL_STAMP(1);                                // strip loaded (issued)
  const int nsolve = head ? q : npb;             // column blocks left of the diagonal: solved against the heads' inverses
#pragma unroll
  for (int s = 0; s < CHOL_Q; ++s) {
    if (s >= nsolve) continue;                   // (no break: the loop must unroll, acc lives in registers)
    // L_rs = A_rs Dinv_s^T
    if (head) chol_wait<true>(flags + s, tag, info); else chol_wait<false>(flags + s, tag, info);
    PANEL_STAMP(2 + 2 * s);                      // Dinv_s seen
    load_V(Dinv + (int64_t)(jb0 + s) * NB * NB, NB);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) U[16 * wave + lk + 4 * r][16 * nt + lr] = acc[s][nt][r];
    __syncthreads();
    if (head && s == q - 1) PANEL_STAMP(11);     // Dinv_s in LDS
    d4t l[4];
    strip_product<true>(U, V, wave, lane, l);
    if (head && s == q - 1) PANEL_STAMP(12);     // L_qs formed
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double *dst = Arow + (int64_t)(16 * wave + lk + 4 * r) * Np + 64 * s + 16 * nt + lr;
        if (head) publish_store(dst, l[nt][r]); else *dst = l[nt][r];
        U[16 * wave + lk + 4 * r][16 * nt + lr] = l[nt][r];      // own rows of U: read by this wave only
      }
    // L_qs for the rows below (barrier inside).  In a head's last step (s = q - 1) the only update is the one of its
    // own diagonal block, which is what the next publication of Dinv waits for: it goes first, the stores of L_qs
    // complete underneath it and the post follows the update (its readers have until Dinv_q arrives)
    const bool post_late = head && s == q - 1;
    if (head && !post_late) chol_post(flags + 4 + 4 * q + s, tag);
    else __syncthreads();                                        // everyone is done with V = Dinv_s
    // A_rc -= L_rs L_cs^T for the later columns
#pragma unroll
    for (int c = s + 1; c < CHOL_Q; ++c) {
      if (c >= ncol) continue;
      if (head && c == q) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) V[16 * wave + lk + 4 * r][16 * nt + lr] = l[nt][r];    // L_cs is this row's own
      } else {
        if (head) chol_wait<true>(flags + 4 + 4 * c + s, tag, info); else chol_wait<false>(flags + 4 + 4 * c + s, tag, info);
        load_V(A + ((int64_t)(jb0 + c) * NB) * Np + (int64_t)(jb0 + s) * NB, Np);
      }
      __syncthreads();
      if (head && c == q && s == q - 1) PANEL_STAMP(13);         // L_qs stored, the own update's operands in LDS
      d4t t[4];
      strip_product<false>(U, V, wave, lane, t);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[c][nt][r] -= t[nt][r];
      __syncthreads();                                           // before V is refilled
      if (head && c == q && s == q - 1) PANEL_STAMP(14);         // own diagonal block updated
    }
    if (post_late) chol_post(flags + 4 + 4 * q + s, tag);
    PANEL_STAMP(3 + 2 * s);                      // step s done
  }
  if (head) {
    // the diagonal block of this head, now with the updates of the steps before it: factor, invert, publish
#pragma unroll
    for (int c = 0; c < CHOL_Q; ++c)
      if (c == q) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + lk + 4 * r, col = 16 * nt + lr;
            U[row][col] = (col <= row) ? acc[c][nt][r] : 0.0;
            V[row][col] = 0.0;
          }
      }
    __syncthreads();
    tile_factor(U, V, T, tid, rb, info);
    tile_inverse(U, V, T, tid);
    double *Db = Dinv + (int64_t)rb * NB * NB;
    for (int idx = tid; idx < NB * NB; idx += 256) publish_store(Db + idx, V[idx >> 6][idx & 63]);
    if (!(fault && jb0 == 0 && q == 1)) chol_post(flags + q, tag);
    // the factor itself is not read before the next launch: after the publication
    for (int idx = tid; idx < NB * NB; idx += 256) Arow[(int64_t)(idx >> 6) * Np + 64 * q + (idx & 63)] = U[idx >> 6][idx & 63];
  }
  PANEL_STAMP(10);
}

// W[ib*64 + r][ib*64 + c] = Dinv[ib][r][c] for every 64 x 64 diagonal block ib = blockIdx.x
__global__ __launch_bounds__(256) void scatter_diag_blocks_kernel(const double *__restrict__ Dinv, double *__restrict__ W,
                                                                  int64_t Np) {
  const int64_t ib = blockIdx.x;
  Dinv += (int64_t)blockIdx.y * Np * NB;      // blockIdx.y: problem of a batch
  W += (int64_t)blockIdx.y * Np * Np;
  const double *src = Dinv + ib * NB * NB;
  double *dst = W + (ib * NB) * Np + ib * NB;
  for (int idx = threadIdx.x; idx < NB * NB; idx += 256) dst[(int64_t)(idx >> 6) * Np + (idx & 63)] = src[idx];
}

// In-place lower Cholesky of the Np x Np matrix A (Np multiple of 64); Dinv receives the inverted
// diagonal blocks [Np/64][64][64].
// nb > 1: a batch of matrices A + z Np^2 (Dinv + z Np 64, dinfo + z), every launch serving all of them.
// Two levels: the 64-wide steps (diagonal factor, panel solve) only update the rest of their own PANEL of CHOL_Q blocks;
// the trailing matrix beyond the panel gets ONE rank-(64 CHOL_Q) update per panel -- a quarter of the passes over it of a
// rank-64 update per step (the update is HBM bound at K = 64: 8 FLOP per byte moved).
// fit handles alive in this process: each may have a panel launch in flight (estimators.fit_gps runs up to three at once)
static std::atomic<int> g_live_fit_handles{0};

// May the heads of a panel sit at workgroups 0, 8, 16, 24 (one XCD) instead of 0 .. 3?  Only where the device holds,
// with a margin, more workgroups of chol_panel_kernel than can be blocked on undispatched heads (24 per concurrent
// panel launch): resident = runtime occupancy x CUs >= 64 x live handles.  GPEMU_CHOL_HEADS_ONE_XCD=0 / 1 forces it
// (tests), GPEMU_CHOL_CAPACITY overrides the resident figure (tests: a small partition).  Read per call.
static int chol_heads_placement(hipStream_t st) {
  (void)st;
  if (const char *e = getenv("GPEMU_CHOL_HEADS_ONE_XCD")) return atoi(e) != 0;
  static int resident = -1;                       // per process: one device kind per process in every supported setup
  if (resident < 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_panel_kernel, 256, 0) == hipSuccess &&
        hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      resident = per_cu * prop.multiProcessorCount;
    else { (void)hipGetLastError(); resident = 0; }
  }
  int cap = resident;
  if (const char *e = getenv("GPEMU_CHOL_CAPACITY")) cap = atoi(e);
  return cap >= 64 * std::max(1, g_live_fit_handles.load()) ? 1 : 0;
}

int device_cholesky_blocked(double *A, int64_t Np, double *Dinv, int *dinfo, hipStream_t st, int nb, const CholOverlap *ov) {
  const int nblk = (int)(Np / NB);
  constexpr int chol_q = 4;
  // A[rows r0 ..][cols c0 .. c1) -= A[rows r0 ..][k0 .. k1) . A[rows c0 .. c1)[k0 .. k1)^T, lower tiles only
  auto update = [&](int64_t r0, int64_t c0, int64_t c1, int64_t k0, int64_t k1, hipStream_t st) -> int {
    const int M = (int)(Np - r0), N = (int)(c1 - c0);
    if (M <= 0 || N <= 0) return GPEMU_OK;
    GemmArgs u;
    u.A = A + r0 * Np + k0; u.lda = Np;
    u.B = A + c0 * Np + k0; u.ldb = Np;
    u.C = A + r0 * Np + c0; u.ldc = Np;
    u.M = M; u.N = N; u.K = (int)(k1 - k0); u.alpha = -1.0; u.beta = 1.0; u.lower_only = (r0 == c0) ? 1 : 0;
    u.strideA = Np * Np; u.strideB = Np * Np; u.strideC = Np * Np;
    return launch_gemm(u, false, false, nb, st);
  };
  bool side_pending = false;
  // an early return (a failed launch, a HIP error) with a side update in flight: the caller may reuse or free A at once,
  // so the side stream is drained first; the regular exit clears the flag after making `st` wait for it
  struct SideDrain {
    const CholOverlap *ov; const bool &pending;
    ~SideDrain() { if (pending && ov && ov->side) (void)hipStreamSynchronize(ov->side); }
  } side_drain{ov, side_pending};
  // one launch per panel (chol_panel_kernel) where the caller provides its flags and the strips of all problems fit on
  // the chip a few times over; GPEMU_CHOL_PANEL=0: the three-launch steps (same bits)
  const int panel_on = getenv("GPEMU_CHOL_PANEL") ? atoi(getenv("GPEMU_CHOL_PANEL")) : 1;      // read per call (tests)
  constexpr int panel_max_wg = 320;
  const bool fused = panel_on && chol_q == CHOL_Q && ov && ov->flags && (int64_t)nblk * nb <= panel_max_wg;
  const int fault = getenv("GPEMU_CHOL_FAULT") ? atoi(getenv("GPEMU_CHOL_FAULT")) : 0;     // tests: a head that never publishes
  if (fused) GP_HIP(hipMemsetAsync(ov->flags, 0, sizeof(int) * (size_t)CHOL_FLAGS * nb, st));
  const int heads_one_xcd = fused ? chol_heads_placement(st) : 0;
  for (int jb0 = 0; jb0 < nblk; jb0 += chol_q) {
    const int jb1 = std::min(nblk, jb0 + chol_q);             // the panel: blocks [jb0, jb1)
    if (fused) {
      hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)(nblk - jb0), (unsigned)nb), dim3(256), 0, st, A, Np, Dinv, jb0,
                         jb1 - jb0, dinfo, ov->flags, jb0 / chol_q + 1, Np * Np, Np * NB, fault, heads_one_xcd);
      GP_HIP(hipGetLastError());
    }
    for (int jb = jb0; jb < jb1 && !fused; ++jb) {
      const int64_t j0 = (int64_t)jb * NB;
      hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, (unsigned)nb), dim3(256), 0, st, A + j0 * Np + j0, Np,
                         Dinv + (int64_t)jb * NB * NB, 1, jb, dinfo, Np * Np, Np * NB);
      GP_HIP(hipGetLastError());
      const int M = (int)(Np - j0 - NB);
      if (M <= 0) break;
      double *A21 = A + (j0 + NB) * Np + j0;
      GemmArgs g;   // panel = A21 . inv(L11)^T, in place: a workgroup owns whole 64-wide rows of the panel and has
      g.A = A21; g.lda = Np;   // read them completely (K = 64) before it stores
      g.B = Dinv + (int64_t)jb * NB * NB; g.ldb = NB;
      g.C = A21; g.ldc = Np;
      g.M = M; g.N = NB; g.K = NB;
      g.strideA = Np * Np; g.strideB = Np * NB; g.strideC = Np * Np;
      int rc = launch_gemm(g, false, false, nb, st);
      if (rc != GPEMU_OK) return rc;
      // the rest of this panel's columns: blocks (jb, jb1), rows from block jb + 1 down
      rc = update(j0 + NB, j0 + NB, (int64_t)jb1 * NB, j0, j0 + NB, st);
      if (rc != GPEMU_OK) return rc;
    }
    // everything beyond the panel, with the whole panel at once
    const int64_t t0 = (int64_t)jb1 * NB, k0 = (int64_t)jb0 * NB;
    if (t0 >= Np) break;
    const int64_t t1 = std::min<int64_t>(Np, t0 + (int64_t)chol_q * NB);       // the next panel's columns: [t0, t1)
    // (worth its two cross-stream events -- ~12 us per panel -- only while the side update is long: measured per panel
    // -43 us with 60 tile rows left, break-even at ~38: profiles/r03_chol_lookahead.txt)
    constexpr int la_min = 40;
    if (!ov || !ov->side || !fused || Np - t1 < (int64_t)la_min * NB) {
      if (side_pending) GP_HIP(hipStreamWaitEvent(st, ov->rest_done, 0));
      side_pending = false;
      const int rc = update(t0, t0, Np, k0, t0, st);
      if (rc != GPEMU_OK) return rc;
      continue;
    }
    // Look-ahead.  Only the NEXT panel's columns are on the serial path: its panel kernel is a chain of four diagonal
    // blocks on at most one CU per strip, which leaves most of the chip idle.  The columns beyond it, [t1, Np), take this panel's update on
    // the side stream while the next panel is factored on `st`:
    //   side stream   reads this panel's columns (final), adds into columns >= t1 -- in its own stream order
    //   st            next panel = columns [t0, t1) only; its update of them adds to what EARLIER side updates left
    //                 there, so it waits for the previous panel's side update (`rest_done`), not for this one's
    GP_HIP(hipEventRecord(ov->panel_done, st));
    if (side_pending) GP_HIP(hipStreamWaitEvent(st, ov->rest_done, 0));        // the side update of the panel before
    side_pending = false;
    int rc = update(t0, t0, t1, k0, t0, st);
    if (rc != GPEMU_OK) return rc;
    if (t1 < Np) {
      GP_HIP(hipStreamWaitEvent(ov->side, ov->panel_done, 0));
      rc = update(t1, t1, Np, k0, t0, ov->side);
      if (rc != GPEMU_OK) return rc;
      GP_HIP(hipEventRecord(ov->rest_done, ov->side));
      side_pending = true;
    }
  }
  if (side_pending) GP_HIP(hipStreamWaitEvent(st, ov->rest_done, 0));
  side_pending = false;
  return GPEMU_OK;
}

// W = L^-1 (lower, Np x Np, zero above the diagonal) from L (lower triangle of `L`, ld Np) and the
// inverted 64 x 64 diagonal blocks.  Bottom-up merging: blocks of size b are already inverted on the
// diagonal of W; a pair (W11, W22) becomes inv([[L11,0],[L21,L22]]) by  W21 = -W22 (L21 W11),
// two GEMMs per level, batched over all pairs (blockIdx.z), so the whole inverse takes
// 2 log2(Np/64) (+ ragged-tail) launches of large MFMA GEMMs.  T is an Np x Np scratch.
int device_trtri_blocked(const double *L, int64_t Np, const double *Dinv, double *W, double *T, hipStream_t st, int nb,
                         bool zero_upper) {
  const int nblk = (int)(Np / NB);
  // what lies above the block diagonal is never written below: zero it for readers of the full square
  if (zero_upper) GP_HIP(hipMemsetAsync(W, 0, sizeof(double) * (size_t)(Np * Np) * nb, st));
  // diagonal blocks: Dinv [nblk][64][64] -> W, one launch
  hipLaunchKernelGGL(scatter_diag_blocks_kernel, dim3((unsigned)nblk, (unsigned)nb), dim3(256), 0, st, Dinv, W, Np);
  GP_HIP(hipGetLastError());
  for (int64_t b = NB; b < Np; b *= 2) {
    // pairs start at p0 = 2 b t; first block [p0, p0 + b), second [p0 + b, min(p0 + 2b, Np))
    const int64_t nfull = Np / (2 * b);                      // pairs whose second block is complete
    const int64_t rem = Np - nfull * 2 * b;                  // leftover rows after the full pairs
    auto merge = [&](int64_t p0, int64_t b2, int batch) -> int {
      // T21 = L21 . W11   (b2 x b) = (b2 x b) (b x b)
      GemmArgs g;
      g.A = L + (p0 + b) * Np + p0; g.lda = Np; g.strideA = 2 * b * Np + 2 * b;
      g.B = W + p0 * Np + p0; g.ldb = Np; g.strideB = 2 * b * Np + 2 * b;
      g.C = T + (p0 + b) * Np + p0; g.ldc = Np; g.strideC = 2 * b * Np + 2 * b;
      g.M = (int)b2; g.N = (int)b; g.K = (int)b;
      g.k_from_n = 1;                                          // W11 is lower triangular: W11[k][n] = 0 for k < n
      g.batch1 = batch; g.stride2A = Np * Np; g.stride2B = Np * Np; g.stride2C = Np * Np;   // x nb matrices
      int rc = launch_gemm(g, false, true, batch * nb, st);
      if (rc != GPEMU_OK) return rc;
      // W21 = -W22 . T21  (b2 x b) = (b2 x b2) (b2 x b)
      GemmArgs h;
      h.A = W + (p0 + b) * Np + (p0 + b); h.lda = Np; h.strideA = 2 * b * Np + 2 * b;
      h.B = T + (p0 + b) * Np + p0; h.ldb = Np; h.strideB = 2 * b * Np + 2 * b;
      h.C = W + (p0 + b) * Np + p0; h.ldc = Np; h.strideC = 2 * b * Np + 2 * b;
      h.M = (int)b2; h.N = (int)b; h.K = (int)b2; h.alpha = -1.0;
      h.k_to_m = 1;                                            // W22 is lower triangular: W22[m][k] = 0 for k > m
      h.batch1 = batch; h.stride2A = Np * Np; h.stride2B = Np * Np; h.stride2C = Np * Np;
      return launch_gemm(h, false, true, batch * nb, st);
    };
    if (nfull > 0) {
      int rc = merge(0, b, (int)nfull);
      if (rc != GPEMU_OK) return rc;
    }
    if (rem > b) {                                           // a ragged pair: (b, rem - b)
      int rc = merge(nfull * 2 * b, rem - b, 1);
      if (rc != GPEMU_OK) return rc;
    }
  }
  return GPEMU_OK;
}

// ---- small vector kernels ------------------------------------------------------------------------
// out[i] = sum_j M[i][j] v[j] (trans = 0, one wave per row) or sum_j M[j][i] v[j] (trans = 1)
// lower = 1: M is lower triangular in 64 x 64 blocks; what lies above the block diagonal is not read (it need not even be
// initialised) -- the skipped terms are exact zeros, so the sums keep their bits.
__global__ void gemv_kernel(const double *__restrict__ Mx, int64_t ld, const double *__restrict__ v,
                            double *__restrict__ out, int n, int trans, int lower) {
  Mx += (int64_t)blockIdx.y * ld * ld;        // blockIdx.y: problem of a batch (square matrices ld x ld, vectors of ld)
  v += (int64_t)blockIdx.y * ld;
  out += (int64_t)blockIdx.y * ld;
  if (!trans) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    double s = 0.0;
    const int jend = lower ? min(n, (row / NB + 1) * NB) : n;
    for (int j = lane; j < jend; j += 64) s = fma(Mx[(int64_t)row * ld + j], v[j], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s;
  } else {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n) return;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s = fma(Mx[(int64_t)j * ld + col], v[j], s);
    out[col] = s;
  }
}

// out[col] = sum_j M[j][col] v[j] in two deterministic passes: 64-row chunks (grid.y) into part[chunk][col],
// then a column sum over the chunks
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(const double *__restrict__ Mx, int64_t ld,
                                                             const double *__restrict__ v, double *__restrict__ part,
                                                             int n, int lower) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int j0 = blockIdx.y * 64;
  if (col >= n) return;
  if (lower && (int)blockIdx.y < col / NB) {      // rows above this column's diagonal block: zeros, not read
    part[(int64_t)blockIdx.z * ld * ld + (int64_t)blockIdx.y * n + col] = 0.0;
    return;
  }
  Mx += (int64_t)blockIdx.z * ld * ld;        // blockIdx.z: problem of a batch
  v += (int64_t)blockIdx.z * ld;
  part += (int64_t)blockIdx.z * ld * ld;
  double s0 = 0.0, s1 = 0.0;
  const int j1 = (j0 + 64 < n) ? j0 + 64 : n;
  int j = j0;
  for (; j + 1 < j1; j += 2) {
    s0 = fma(Mx[(int64_t)j * ld + col], v[j], s0);
    s1 = fma(Mx[(int64_t)(j + 1) * ld + col], v[j + 1], s1);
  }
  if (j < j1) s0 = fma(Mx[(int64_t)j * ld + col], v[j], s0);
  part[(int64_t)blockIdx.y * n + col] = s0 + s1;
}

__global__ __launch_bounds__(256) void colsum_kernel(const double *__restrict__ part, int nchunk, int n,
                                                     double *__restrict__ out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= n) return;
  part += (int64_t)blockIdx.y * n * n;        // blockIdx.y: problem of a batch (n = Np)
  out += (int64_t)blockIdx.y * n;
  double s = 0.0;
  for (int c = 0; c < nchunk; ++c) s += part[(int64_t)c * n + col];
  out[col] = s;
}

// scal[0] = y.alpha ; scal[1] = sum_{i<N} log L_ii   (one workgroup)
__global__ __launch_bounds__(1024) void lml_terms_kernel(const double *__restrict__ y, const double *__restrict__ alpha,
                                                         const double *__restrict__ L, int64_t ld, int N,
                                                         double *__restrict__ scal) {
  __shared__ double part[2][16];
  y += (int64_t)blockIdx.x * ld;              // blockIdx.x: problem of a batch
  alpha += (int64_t)blockIdx.x * ld;
  L += (int64_t)blockIdx.x * ld * ld;
  scal += blockIdx.x * 4;
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    a = fma(y[i], alpha[i], a);
    b += log(L[(int64_t)i * ld + i]);
  }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = a; part[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { sa += part[0][w]; sb += part[1][w]; }
    scal[0] = sa; scal[1] = sb;
  }
}

// ---- gradient contraction --------------------------------------------------------------------------
// gpart[block][t] = 1/2 sum over this block's (j, l) of (alpha_j alpha_l - Kinv_jl) dK_jl/dtheta_t
// theta order: log l_1..l_d, (log const), (log noise)   (skl kernels.py:733-760, Sum :861-866)
constexpr int NTH_MAX = DPAD + 2;
constexpr int GRAD_ROWS = 16;
__global__ __launch_bounds__(256) void lml_grad_kernel(const double *__restrict__ X, const double *__restrict__ hp,
                                                       const double *__restrict__ alpha,
                                                       const double *__restrict__ Kinv, int64_t ld,
                                                       double *__restrict__ gpart, int N, int d, int kind,
                                                       int has_const, int has_noise) {
  __shared__ double red[NTH_MAX][4];
  // 1-D grid over the (row group, 256-column block) pairs that reach under the diagonal only (half of the full grid's
  // workgroups would start to find nothing to do): row groups 16 b .. 16 b + 15 have b + 1 blocks, so position
  // t = (b + 1)(8 b + r) + x  <->  row group 16 b + r, block x
  int b = (int)((sqrt(1.0 + 0.5 * (double)blockIdx.x) - 1.0) * 0.5);
  while (8 * b * (b + 1) > (int)blockIdx.x) --b;
  while (8 * (b + 1) * (b + 2) <= (int)blockIdx.x) ++b;
  const int rem = (int)blockIdx.x - 8 * b * (b + 1);
  const int by = 16 * b + rem / (b + 1), bx = rem % (b + 1);
  const int l = bx * blockDim.x + threadIdx.x;
  const int j0 = by * GRAD_ROWS;                       // this workgroup's rows j0 .. j0 + GRAD_ROWS - 1 (one reduction for all)
  hp += (int64_t)blockIdx.z * (DPAD + 2);              // blockIdx.z: problem of a batch
  alpha += (int64_t)blockIdx.z * ld;
  Kinv += (int64_t)blockIdx.z * ld * ld;
  gpart += (int64_t)blockIdx.z * gridDim.x * NTH_MAX;
  double acc[NTH_MAX];
#pragma unroll
  for (int t = 0; t < NTH_MAX; ++t) acc[t] = 0.0;
  double xl[DPAD], il2[DPAD];
#pragma unroll
  for (int dd = 0; dd < DPAD; ++dd) {
    xl[dd] = (l < N) ? X[l * DPAD + dd] : 0.0;
    il2[dd] = 1.0 / (hp[dd] * hp[dd]);      // once per thread; a division per pair and dimension was 4/5 of this kernel
  }
  const double al = (l < N) ? alpha[l] : 0.0;
  // the summand is symmetric in (j, l): the lower triangle counts twice, Kinv is only read (and only valid) there
  for (int j = j0; j < j0 + GRAD_ROWS && j < N; ++j) {
    if (l > j) continue;
    const double wgt = (l < j ? 2.0 : 1.0) * (alpha[j] * al - Kinv[(int64_t)j * ld + l]);
    double D[DPAD], r2 = 0.0;
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) {
      double df = X[j * DPAD + dd] - xl[dd];
      D[dd] = (df * df) * il2[dd];             // (x - x')^2 / l^2   (skl kernels.py:1574, 1748)
      r2 += D[dd];
    }
    double f;  // dK_base/dlog l_dd = f * D[dd]
    if (kind == 0) {
      f = (j == l) ? 1.0 : exp(-0.5 * r2);                       // K_gradient = D * K
    } else if (kind == 1) {
      double r = sqrt(r2);
      f = (r > 0.0) ? exp(-r) / r : 0.0;                          // K * D / sqrt(sum D), 0 where r == 0
    } else if (kind == 2) {
      f = 3.0 * exp(-sqrt(3.0 * r2));                             // 3 D exp(-sqrt(3 sum D))
    } else {
      double tmp = sqrt(5.0 * r2);
      f = 5.0 / 3.0 * (tmp + 1.0) * exp(-tmp);                    // 5/3 D (tmp + 1) exp(-tmp)
    }
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) acc[dd] += 0.5 * wgt * f * D[dd];
    if (has_const) acc[d] += 0.5 * wgt * hp[DPAD];
    if (has_noise && j == l) acc[d + has_const] += 0.5 * wgt * hp[DPAD + 1];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < NTH_MAX; ++t) {
    double s = acc[t];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) red[t][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < NTH_MAX) {
    const int t = threadIdx.x;
    gpart[(int64_t)blockIdx.x * NTH_MAX + t] = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
  }
}

// two-stage deterministic sum of the partial gradients: GR_BLOCKS workgroups each sum a contiguous slice ...
constexpr int GR_BLOCKS = 128;
__global__ __launch_bounds__(256) void grad_reduce_stage1_kernel(const double *__restrict__ gpart, int nparts,
                                                                 double *__restrict__ stage, int nth) {
  __shared__ double part[4];
  gpart += (int64_t)blockIdx.y * nparts * NTH_MAX;     // blockIdx.y: problem of a batch
  stage += (int64_t)blockIdx.y * GR_BLOCKS * NTH_MAX;
  const int per = (nparts + GR_BLOCKS - 1) / GR_BLOCKS;
  const int i0 = blockIdx.x * per, i1 = (i0 + per < nparts) ? i0 + per : nparts;
  for (int t = 0; t < nth; ++t) {
    double s = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) s += gpart[(int64_t)i * NTH_MAX + t];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) stage[blockIdx.x * NTH_MAX + t] = (part[0] + part[1]) + (part[2] + part[3]);
  }
}

// ... and one workgroup sums the GR_BLOCKS slices
__global__ __launch_bounds__(64) void grad_reduce_kernel(const double *__restrict__ stage, double *__restrict__ grad, int nth) {
  stage += (int64_t)blockIdx.x * GR_BLOCKS * NTH_MAX;  // blockIdx.x: problem of a batch
  grad += blockIdx.x * NTH_MAX;
  for (int t = 0; t < nth; ++t) {
    double s = 0.0;
    for (int i = threadIdx.x; i < GR_BLOCKS; i += 64) s += stage[i * NTH_MAX + t];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) grad[t] = s;
  }
}

// transpose the top-left N x N of W (ld Np) into Wt (ld Npad): Wt[j][i] = W[i][j]
__global__ void transpose_kernel(const double *__restrict__ W, int64_t ldw, double *__restrict__ Wt, int64_t ldt, int n) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int i = by + r, j = bx + threadIdx.x;
    tile[r][threadIdx.x] = (i < n && j < n) ? W[(int64_t)i * ldw + j] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int j = bx + r, i = by + threadIdx.x;
    if (i < n && j < n) Wt[(int64_t)j * ldt + i] = tile[threadIdx.x][r];
  }
}

int launch_transpose(const double *W, int64_t ldw, double *Wt, int64_t ldt, int n, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32)), dim3(32, 8), 0, st,
                     W, ldw, Wt, ldt, n);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

// copy an N x N lower-triangular host-layout factor (ld N) into the padded Np x Np buffer (identity tail)
__global__ void pad_lower_kernel(const double *__restrict__ L, int N, double *__restrict__ A, int Np) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= Np) return;
  double v = 0.0;
  if (i < N && j < N) v = (j <= i) ? L[(int64_t)i * N + j] : 0.0;
  else if (i == j) v = 1.0;
  A[(int64_t)i * Np + j] = v;
}

// W = L^-1 of one N x N lower factor (device pointer, ld N) written transposed into Wt (ld Npad).
// scratch: A [Np*Np], Dinv [Np*64], W [Np*Np], T [Np*Np]
int device_invert_factor_to_Wt(const double *dL, int64_t N, double *Wt, int64_t Npad, double *A, double *Dinv,
                               double *W, double *T, hipStream_t st) {
  const int64_t Np = round_up(N, NB);
  hipLaunchKernelGGL(pad_lower_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)Np), dim3(256), 0, st, dL, (int)N, A, (int)Np);
  hipLaunchKernelGGL(potrf_diag_kernel, dim3((unsigned)(Np / NB)), dim3(256), 0, st, A, Np, Dinv, 0, 0, (int *)nullptr,
                     (int64_t)0, (int64_t)0);
  GP_HIP(hipGetLastError());
  int rc = device_trtri_blocked(A, Np, Dinv, W, T, st);
  if (rc != GPEMU_OK) return rc;
  return launch_transpose(W, Np, Wt, Npad, (int)N, st);
}

}  // namespace gpemu

// ================================================================================================
struct gpemu_fit {
  int device = 0;
  int64_t N = 0, d = 0, Np = 0;
  int kind = 0, has_const = 0, has_noise = 0;
  double jitter = 0.0;
  hipStream_t stream = nullptr;
  gpemu::CholOverlap overlap;      // side stream + events of the Cholesky look-ahead
  double *X = nullptr, *hp = nullptr, *K = nullptr, *Dinv = nullptr, *W = nullptr,
         *T = nullptr, *Kinv = nullptr, *y = nullptr, *v = nullptr, *alpha = nullptr, *gpart = nullptr,
         *scal = nullptr, *grad = nullptr, *gstage = nullptr;
  int *info = nullptr;
  int n_gparts = 0;
  int cap = 0;              // problems the workspace holds (fit_reserve)
  bool counted = false;     // in g_live_fit_handles (the panel kernel's head placement rule)
};

using namespace gpemu;
#define GP_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != GPEMU_OK) return rc__; \
  } while (0)

static int kind_of(int kernel_kind, double nu) {
  if (kernel_kind == GPEMU_KERNEL_RBF) return 0;
  return nu == 0.5 ? 1 : (nu == 1.5 ? 2 : 3);
}

// workspace for `nb` problems evaluated together (one set of matrices each)
static int fit_reserve(gpemu_fit *f, int nb) {
  if (nb <= f->cap) return GPEMU_OK;
  GP_HIP(hipStreamSynchronize(f->stream));
  double **ptrs[] = {&f->hp, &f->K, &f->Dinv, &f->W, &f->T, &f->Kinv, &f->y, &f->v, &f->alpha, &f->gpart, &f->scal,
                     &f->grad, &f->gstage};
  for (double **p : ptrs) { (void)hipFree(*p); *p = nullptr; }
  (void)hipFree(f->info);
  f->info = nullptr;
  f->cap = 0;
  const int64_t Np = f->Np;
  hipError_t e = hipSuccess;
  auto A = [&](double **p, int64_t n) { if (e == hipSuccess) e = hipMalloc((void **)p, sizeof(double) * (size_t)(n > 0 ? n : 1) * nb); };
  A(&f->hp, DPAD + 2); A(&f->K, Np * Np); A(&f->Dinv, Np * NB); A(&f->W, Np * Np); A(&f->T, Np * Np);
  A(&f->Kinv, Np * Np); A(&f->y, Np); A(&f->v, Np); A(&f->alpha, Np);
  A(&f->gpart, (int64_t)f->n_gparts * NTH_MAX); A(&f->scal, 4); A(&f->grad, NTH_MAX);
  A(&f->gstage, (int64_t)GR_BLOCKS * NTH_MAX);
  if (e == hipSuccess) e = hipMalloc((void **)&f->info, sizeof(int) * nb);
  (void)hipFree(f->overlap.flags);
  f->overlap.flags = nullptr;
  if (e == hipSuccess) e = hipMalloc((void **)&f->overlap.flags, sizeof(int) * (size_t)CHOL_FLAGS * nb);
  if (e != hipSuccess) { set_error("fit workspace for %d problems: %s", nb, hipGetErrorString(e)); return GPEMU_ERR_HIP; }
  f->cap = nb;
  return GPEMU_OK;
}

// K, Cholesky, W, alpha, lml terms for nb problems (theta_z, y_z) at once -- every launch of the chain serves all of
// them (blockIdx carries the problem index, the GEMMs run batched) --; optional K^-1 + gradient.  info[z] > 0: the
// kernel matrix of problem z is not positive definite (its lml / grad are then meaningless).
static int fit_eval_batch(gpemu_fit *f, int nb, const double *ys, const double *thetas, int64_t n_theta, bool want_grad,
                          double *lml, double *grad, int *info_out) {
  const int64_t N = f->N, Np = f->Np, d = f->d;
  const int nth = (int)(d + f->has_const + f->has_noise);
  GP_ARG(n_theta == nth, "n_theta must be d (+1 constant) (+1 noise)");
  GP_ARG(nb >= 1, "empty batch");
  GP_TRY(fit_reserve(f, nb));
  hipStream_t st = f->stream;
  std::vector<double> hp((size_t)nb * (DPAD + 2)), hy((size_t)nb * Np, 0.0);
  for (int z = 0; z < nb; ++z) {
    const double *theta = thetas + (size_t)z * nth;
    double *h = hp.data() + (size_t)z * (DPAD + 2);
    for (int i = 0; i < DPAD; ++i) h[i] = i < d ? std::exp(theta[i]) : 1.0;
    h[DPAD] = f->has_const ? std::exp(theta[d]) : 0.0;
    h[DPAD + 1] = f->has_noise ? std::exp(theta[d + f->has_const]) : 0.0;
    for (int64_t i = 0; i < N; ++i) hy[(size_t)z * Np + i] = ys[(size_t)z * N + i];
  }
  GP_HIP(hipMemcpyAsync(f->hp, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice, st));
  GP_HIP(hipMemcpyAsync(f->y, hy.data(), sizeof(double) * hy.size(), hipMemcpyHostToDevice, st));
  GP_HIP(hipMemsetAsync(f->info, 0, sizeof(int) * nb, st));
  hipLaunchKernelGGL(kmat_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)((Np + KMAT_ROWS - 1) / KMAT_ROWS), (unsigned)nb), dim3(256), 0, st, f->X,
                     f->hp, f->K, (int)N, (int)Np, f->kind, f->jitter);
  GP_HIP(hipGetLastError());
  // look-ahead (with the one-launch panels only; it gained nothing over the three-launch steps, whose narrow kernels
  // slowed down by what the overlap won: profiles/r03_chol_lookahead.txt); GPEMU_CHOL_LOOKAHEAD=0 switches it off
  const char *la = getenv("GPEMU_CHOL_LOOKAHEAD");
  const bool lookahead = f->overlap.side && !(la && atoi(la) == 0);
  CholOverlap ov = f->overlap;
  if (!lookahead) ov.side = nullptr;
  GP_TRY(device_cholesky_blocked(f->K, Np, f->Dinv, f->info, st, nb, &ov));
  // every reader of W below keeps to the 64 x 64 blocks on and under the diagonal: no zero fill of the rest
  GP_TRY(device_trtri_blocked(f->K, Np, f->Dinv, f->W, f->T, st, nb, false));
  // alpha = W^T (W y)
  hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)((Np + 3) / 4), (unsigned)nb), dim3(256), 0, st, f->W, Np, f->y, f->v,
                     (int)Np, 0, 1);
  {
    const int nchunk = (int)((Np + 63) / 64);                 // T (Np x Np) is free again after the inverse
    hipLaunchKernelGGL(gemv_t_partial_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)nchunk, (unsigned)nb), dim3(256),
                       0, st, f->W, Np, f->v, f->T, (int)Np, 1);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)nb), dim3(256), 0, st, f->T, nchunk,
                       (int)Np, f->alpha);
  }
  hipLaunchKernelGGL(lml_terms_kernel, dim3((unsigned)nb), dim3(1024), 0, st, f->y, f->alpha, f->K, Np, (int)N, f->scal);
  GP_HIP(hipGetLastError());
  if (want_grad) {
    GemmArgs g;  // K^-1 = W^T W: symmetric, so only the tiles on and below the diagonal; W is lower triangular, so
    g.A = f->W; g.lda = Np; g.B = f->W; g.ldb = Np; g.C = f->Kinv; g.ldc = Np;   // (W^T W)[a][b] = sum_{i >= max(a, b)}:
    g.M = (int)Np; g.N = (int)Np; g.K = (int)Np;                                  // N^3/3 FLOP instead of 2 N^3
    g.lower_only = 1; g.k_from_m = 1;
    g.strideA = Np * Np; g.strideB = Np * Np; g.strideC = Np * Np;
    GP_TRY(launch_gemm(g, true, true, nb, st));
    // (row group, column block) pairs under the diagonal: row group y has y / 16 + 1 blocks of 256 columns
    const int ngroups = (int)((N + GRAD_ROWS - 1) / GRAD_ROWS);
    int npairs = 0;
    for (int y = 0; y < ngroups; ++y) npairs += y / 16 + 1;
    dim3 grid((unsigned)npairs, 1, (unsigned)nb);
    hipLaunchKernelGGL(lml_grad_kernel, grid, dim3(256), 0, st, f->X, f->hp, f->alpha, f->Kinv, Np, f->gpart, (int)N,
                       (int)d, f->kind, f->has_const, f->has_noise);
    hipLaunchKernelGGL(grad_reduce_stage1_kernel, dim3(GR_BLOCKS, (unsigned)nb), dim3(256), 0, st, f->gpart,
                       npairs, f->gstage, nth);
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)nb), dim3(64), 0, st, f->gstage, f->grad, nth);
    GP_HIP(hipGetLastError());
  }
  std::vector<double> hs((size_t)nb * 4), hg((size_t)nb * NTH_MAX);
  std::vector<int> info((size_t)nb, 0);
  GP_HIP(hipMemcpyAsync(hs.data(), f->scal, sizeof(double) * hs.size(), hipMemcpyDeviceToHost, st));
  GP_HIP(hipMemcpyAsync(info.data(), f->info, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
  if (want_grad) GP_HIP(hipMemcpyAsync(hg.data(), f->grad, sizeof(double) * hg.size(), hipMemcpyDeviceToHost, st));
  GP_HIP(hipStreamSynchronize(st));
  int first_bad = 0;
  for (int z = 0; z < nb; ++z) {
    if (info_out) info_out[z] = info[z];
    if (info[z] < 0) {
      set_error("fit: a workgroup of the panel factorisation waited for its neighbour beyond the poll bound");
      return GPEMU_ERR_STATE;
    }
    if (info[z] != 0 && first_bad == 0) first_bad = info[z];
    if (lml) lml[z] = -0.5 * hs[(size_t)z * 4] - hs[(size_t)z * 4 + 1] - 0.5 * (double)N * std::log(2.0 * M_PI);
    if (want_grad && grad)
      for (int t = 0; t < nth; ++t) grad[(size_t)z * nth + t] = hg[(size_t)z * NTH_MAX + t];
  }
  if (first_bad != 0 && !info_out) {
    set_error("kernel matrix is not positive definite (pivot %d): the kernel is not returning a positive "
              "definite matrix; try increasing alpha", first_bad);
    return first_bad;   // sklearn raises LinAlgError here (skl _gpr.py:350-358)
  }
  return GPEMU_OK;
}

static int fit_eval(gpemu_fit *f, const double *y, const double *theta, int64_t n_theta, bool want_grad,
                    double *lml, double *grad) {
  return fit_eval_batch(f, 1, y, theta, n_theta, want_grad, lml, grad, nullptr);
}

extern "C" {

int gpemu_fit_create(gpemu_fit **out, int device, int64_t N, int64_t d, const double *X, int kernel_kind,
                     double nu, int has_const, int has_noise, double jitter) {
  GP_ARG(out && X, "null pointer");
  *out = nullptr;
  GP_ARG(N > 0 && d > 0 && d <= DPAD, "N > 0 and 0 < d <= 8 required");
  GP_ARG(kernel_kind == GPEMU_KERNEL_RBF || kernel_kind == GPEMU_KERNEL_MATERN, "kernel_kind");
  if (kernel_kind == GPEMU_KERNEL_MATERN) GP_ARG(nu == 0.5 || nu == 1.5 || nu == 2.5, "Matern nu must be 0.5, 1.5 or 2.5");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device available: libgpemu has no CPU implementation");
    return GPEMU_ERR_NO_DEVICE;
  }
  GP_ARG(device >= 0 && device < ndev, "device");
  GP_HIP(hipSetDevice(device));
  gpemu_fit *f = new gpemu_fit();
  f->device = device; f->N = N; f->d = d; f->Np = round_up(N, NB);
  f->kind = kind_of(kernel_kind, nu); f->has_const = has_const ? 1 : 0; f->has_noise = has_noise ? 1 : 0;
  f->jitter = jitter;
  const int64_t Np = f->Np;
  f->n_gparts = (int)(((N + 255) / 256) * N);
  // the serial chain of the factorisation runs on `stream`; the look-ahead updates fill the rest of the chip from a
  // stream of lower priority, so that a waiting step of the chain is dispatched first
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  hipError_t e = hipStreamCreateWithPriority(&f->stream, hipStreamNonBlocking, prio_greatest);
  if (e == hipSuccess) {
    // The side stream may not use the first GPEMU_CHOL_RESERVE (default 4) CUs of every XCD: its GEMM workgroups take a
    // quarter of a CU's LDS each and are replaced one by one as they finish, so without a reserve the 75 KiB diagonal-block
    // kernel of the serial chain finds no CU with room until the whole side grid has drained (measured: 18 -> 100 us).
    // hipExtStreamCreateWithCUMask: bit i = CU i / 8 of XCD i % 8 (profiles/r03_cu_mask_probe.txt).
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    const int ncu = prop.multiProcessorCount;
    constexpr int reserve = 4;
    if (e == hipSuccess && reserve > 0 && ncu % 8 == 0 && 8 * reserve < ncu) {
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
      for (int i = 8 * reserve; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
      e = hipExtStreamCreateWithCUMask(&f->overlap.side, (uint32_t)mask.size(), mask.data());
    } else if (e == hipSuccess) {
      e = hipStreamCreateWithPriority(&f->overlap.side, hipStreamNonBlocking, prio_least);
    }
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&f->overlap.panel_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&f->overlap.rest_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void **)&f->X, sizeof(double) * (size_t)(Np * DPAD));
  if (e == hipSuccess && fit_reserve(f, 1) != GPEMU_OK) e = hipErrorOutOfMemory;
  std::vector<double> hX((size_t)(Np * DPAD), 0.0);
  for (int64_t i = 0; i < N; ++i)
    for (int64_t dd = 0; dd < d; ++dd) hX[i * DPAD + dd] = X[i * d + dd];
  if (e == hipSuccess) e = hipMemcpy(f->X, hX.data(), sizeof(double) * hX.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error("fit_create: %s", hipGetErrorString(e));
    gpemu_fit_destroy(f);
    return GPEMU_ERR_HIP;
  }
  f->counted = true;
  g_live_fit_handles.fetch_add(1);
  *out = f;
  return GPEMU_OK;
}

int gpemu_fit_destroy(gpemu_fit *f) {
  if (!f) return GPEMU_OK;
  if (f->counted) g_live_fit_handles.fetch_sub(1);
  (void)hipSetDevice(f->device);
  if (f->stream) (void)hipStreamSynchronize(f->stream);
  if (f->overlap.side) { (void)hipStreamSynchronize(f->overlap.side); (void)hipStreamDestroy(f->overlap.side); }
  if (f->overlap.panel_done) (void)hipEventDestroy(f->overlap.panel_done);
  if (f->overlap.rest_done) (void)hipEventDestroy(f->overlap.rest_done);
  double *ptrs[] = {f->X, f->hp, f->K, f->Dinv, f->W, f->T, f->Kinv, f->y, f->v, f->alpha, f->gpart, f->scal, f->grad,
                    f->gstage};
  for (double *p : ptrs) (void)hipFree(p);
  (void)hipFree(f->info);
  (void)hipFree(f->overlap.flags);
  if (f->stream) (void)hipStreamDestroy(f->stream);
  delete f;
  return GPEMU_OK;
}

int gpemu_fit_lml(gpemu_fit *f, const double *y, const double *theta, int64_t n_theta, double *lml, double *grad) {
  GP_ARG(f && y && theta && lml, "null pointer");
  GP_HIP(hipSetDevice(f->device));
  return fit_eval(f, y, theta, n_theta, grad != nullptr, lml, grad);
}

int gpemu_fit_lml_batch(gpemu_fit *f, int64_t n_problems, const double *ys, const double *thetas, int64_t n_theta,
                        double *lml, double *grad, int32_t *info) {
  GP_ARG(f && ys && thetas && lml && info, "null pointer");
  GP_ARG(n_problems >= 1 && n_problems <= 4096, "n_problems");
  GP_HIP(hipSetDevice(f->device));
  return fit_eval_batch(f, (int)n_problems, ys, thetas, n_theta, grad != nullptr, lml, grad, info);
}

int gpemu_fit_factor(gpemu_fit *f, const double *y, const double *theta, int64_t n_theta, double *L_out,
                     double *alpha_out, double *lml) {
  GP_ARG(f && y && theta, "null pointer");
  GP_HIP(hipSetDevice(f->device));
  double l = 0.0;
  GP_TRY(fit_eval(f, y, theta, n_theta, false, &l, nullptr));
  if (lml) *lml = l;
  const int64_t N = f->N, Np = f->Np;
  if (alpha_out) GP_HIP(hipMemcpy(alpha_out, f->alpha, sizeof(double) * N, hipMemcpyDeviceToHost));
  if (L_out) {
    GP_HIP(hipMemcpy2D(L_out, sizeof(double) * N, f->K, sizeof(double) * Np, sizeof(double) * N, (size_t)N, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i)
      for (int64_t j = i + 1; j < N; ++j) L_out[i * N + j] = 0.0;   // only the lower triangle is the factor
  }
  return GPEMU_OK;
}

int gpemu_kernel_matrix(int device, int64_t N, int64_t d, const double *X, const double *theta, int64_t n_theta,
                        int kernel_kind, double nu, int has_const, int has_noise, double jitter, double *K_out) {
  GP_ARG(X && theta && K_out, "null pointer");
  gpemu_fit *f = nullptr;
  GP_TRY(gpemu_fit_create(&f, device, N, d, X, kernel_kind, nu, has_const, has_noise, jitter));
  int rc = GPEMU_OK;
  const int nth = (int)(d + f->has_const + f->has_noise);
  if (n_theta != nth) { set_error("bad argument: n_theta"); rc = GPEMU_ERR_ARG; }
  if (rc == GPEMU_OK) {
    double hp[DPAD + 2];
    for (int i = 0; i < DPAD; ++i) hp[i] = i < d ? std::exp(theta[i]) : 1.0;
    hp[DPAD] = f->has_const ? std::exp(theta[d]) : 0.0;
    hp[DPAD + 1] = f->has_noise ? std::exp(theta[d + f->has_const]) : 0.0;
    hipError_t e = hipMemcpy(f->hp, hp, sizeof(hp), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kmat_kernel, dim3((unsigned)((f->Np + 255) / 256), (unsigned)((f->Np + KMAT_ROWS - 1) / KMAT_ROWS)), dim3(256), 0, f->stream,
                       f->X, f->hp, f->K, (int)N, (int)f->Np, f->kind, jitter);
    if (e == hipSuccess) e = hipStreamSynchronize(f->stream);
    if (e == hipSuccess)
      e = hipMemcpy2D(K_out, sizeof(double) * N, f->K, sizeof(double) * f->Np, sizeof(double) * N, (size_t)N, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("kernel_matrix: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  }
  gpemu_fit_destroy(f);
  return rc;
}

int gpemu_cholesky(int device, int64_t N, double *A_inout) {
  GP_ARG(A_inout && N > 0, "A / N");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device available: libgpemu has no CPU implementation");
    return GPEMU_ERR_NO_DEVICE;
  }
  GP_ARG(device >= 0 && device < ndev, "device");
  GP_HIP(hipSetDevice(device));
  const int64_t Np = round_up(N, NB);
  double *A = nullptr, *Dinv = nullptr;
  int *dinfo = nullptr;
  hipError_t e = hipMalloc((void **)&A, sizeof(double) * Np * Np);
  if (e == hipSuccess) e = hipMalloc((void **)&Dinv, sizeof(double) * Np * NB);
  if (e == hipSuccess) e = hipMalloc((void **)&dinfo, sizeof(int));
  std::vector<double> h((size_t)(Np * Np), 0.0);
  for (int64_t i = 0; i < Np; ++i)
    for (int64_t j = 0; j <= i; ++j) h[i * Np + j] = (i < N && j < N) ? A_inout[i * N + j] : (i == j ? 1.0 : 0.0);
  int rc = GPEMU_OK, info = 0;
  if (e == hipSuccess) e = hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(dinfo, 0, sizeof(int));
  if (e == hipSuccess) rc = device_cholesky_blocked(A, Np, Dinv, dinfo, nullptr);
  if (e == hipSuccess && rc == GPEMU_OK) e = hipDeviceSynchronize();
  if (e == hipSuccess && rc == GPEMU_OK) e = hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == GPEMU_OK) e = hipMemcpy(h.data(), A, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
  (void)hipFree(A); (void)hipFree(Dinv); (void)hipFree(dinfo);
  if (e != hipSuccess) { set_error("cholesky: %s", hipGetErrorString(e)); return GPEMU_ERR_HIP; }
  if (rc != GPEMU_OK) return rc;
  if (info < 0) { set_error("cholesky: a wait inside the factorisation kernels expired"); return GPEMU_ERR_STATE; }
  if (info != 0) { set_error("matrix is not positive definite (pivot %d)", info); return info; }
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = 0; j < N; ++j) A_inout[i * N + j] = (j <= i) ? h[i * Np + j] : 0.0;
  return GPEMU_OK;
}

}  // extern "C"
