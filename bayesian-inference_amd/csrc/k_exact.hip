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

// Matrix-core form of the covariance writer (k <= 16, F even).  The VALU forms above spend 10 fp64 FMAs and nine LDS
// reads per output element in every lane: fine when all 256 CUs share an HBM-bound job, but only ~30 GB/s per CU, so
// the writer cannot be confined to the few CUs an HBM-write-bound kernel really needs (a store-only kernel reaches
// 5.2 TB/s on 64 CUs, profiles/r03_cu_mask_probe.txt) while the GP stage of the next chunk uses the rest.  Here
//     cov_b = U diag(var_b) U^T + C,   U = diag(s) S_k^T  (F x k),   C = (C_unexpl / n_div) o (s s^T)
// is a rank-k product on v_mfma_f64_16x16x4_f64: the B fragments (U for the wave's columns) and the accumulator inputs
// (C) stay in registers across the workgroup's samples, the A fragment of a sample is U scaled by its variances: three
// multiplies per lane and sample, no LDS.  A wave owns 16 rows x 128 columns; a 32-column block is computed as TWO MFMA
// tiles holding its even and its odd columns, so a lane ends up with (row, column 2c) and (row, column 2c + 1).
// Written straight from the accumulator layout a wave-instruction would cover 256 bytes in each of 4 rows, and rows of
// 8 F bytes are not 128-byte aligned: pieces of 256 or 512 bytes reach 3.3 TB/s on this part where 1 KiB contiguous
// per wave-instruction reaches 5.4 (profiles/r03_cu_mask_probe.txt).  So the workgroup's 16 x F block of a sample --
// contiguous in memory and line aligned when F <= 512 -- is staged in LDS in its memory order and stored flat: every
// wave-instruction writes 1 KiB of whole cache lines.
typedef double d4x __attribute__((ext_vector_type(4)));

// GP mean / variance of a sample and PC either from arrays [B][k] or -- arrays null -- summed on the fly from the GP
// stage's partial sums exactly as reduce_mean_var_kernel does (same order, same clipping: skl _gpr.py:479-494, ref:
// emulation.py:499), so that emulation.predict needs no reduction launch
struct GpParts {
  const double *mean, *var;                 // [B][k] or null
  const double *mean_part, *vsq_part, *kdiag;
  int nchunk, nrb;
};
__device__ __forceinline__ double gp_var_of(const GpParts &g, int64_t b, int p, int k) {
  if (g.var) return g.var[b * k + p];
  double vs = 0.0;
  for (int r = 0; r < g.nrb; ++r) vs += g.vsq_part[(b * k + p) * g.nrb + r];
  double v = g.kdiag[p] - vs;
  if (v < 0.0) v = 0.0;
  const double sd = sqrt(v);
  return sd * sd;
}
__device__ __forceinline__ double gp_mean_of(const GpParts &g, int64_t b, int p, int k) {
  if (g.mean) return g.mean[b * k + p];
  double mu = 0.0;
  for (int c = 0; c < g.nchunk; ++c) mu += g.mean_part[(b * k + p) * g.nchunk + c];
  return mu;
}

constexpr int PM_NB = 16;      // samples per work item (PM_NB * 16 <= 512 threads stage its variances)

// Persistent: one 512-thread workgroup per CU of the stream it runs on (the grid is the worker count), each walking a
// contiguous range of (row block, group of PM_NB samples) items, row block major, so its constants change once or
// twice.  The LDS block is double buffered and the loop is software pipelined: a wave issues the MFMAs of sample i,
// then -- while the matrix pipe works -- reads sample i - 1's finished block from the other buffer and stores it to
// memory, then writes its accumulators into this sample's buffer: ONE barrier per sample, and the matrix pipe, the LDS
// and the store path of a CU are busy at the same time.
template <int KS, int CBW, int DBG = 0>   // k-steps of 4: k <= 4 KS; CBW 32-column blocks per wave (2: 8 waves, 1: 16 waves)
__global__ __launch_bounds__(1024 / CBW) void predict_cov_mfma_kernel(
    GpParts gp, const double *__restrict__ comp, const double *__restrict__ sscale,
    const double *__restrict__ cun, double *__restrict__ cov, int64_t B, int F, int k, double inv_ndiv, int ngroups,
    int total_items) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tile_sz = 16 * F;
  double *vstage = smem + 2 * tile_sz;           // [PM_NB][4 KS]: the item's variances (0 beyond k)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  constexpr int NT = 1024 / CBW;                 // threads
  const int gbase = wave * 32 * CBW;             // a wave owns 16 rows x 32 CBW columns (F <= 512: one column pass)
  const int t0 = (int)((int64_t)total_items * blockIdx.x / gridDim.x);
  const int t1 = (int)((int64_t)total_items * (blockIdx.x + 1) / gridDim.x);
  // lane (q, c): A[i = c][kk = q], B[kk = q][n = c]; D[reg] = (row q + 4 reg, column c)
  double ua[KS], ub[CBW][2][KS];
  d4x ci[CBW][2];
  double dbg_acc = 0.0;
  int cur_rb = -1, buf = 0, pend_n = 0;          // pend_n = 0: nothing staged yet
  double *pend_dst = cov;
  double vnext = 0.0;                            // thread i: entry i of the next item's [PM_NB][4 KS] variances
  if (t0 < t1 && (int)threadIdx.x < PM_NB * 4 * KS) {
    const int grp0 = t0 % ngroups, ib = threadIdx.x / (4 * KS), p = threadIdx.x - ib * 4 * KS;
    const int64_t b = (int64_t)grp0 * PM_NB + ib;
    vnext = (b < B && p < k) ? gp_var_of(gp, b, p, k) : 0.0;
  }
#pragma unroll 1
  for (int t = t0; t < t1; ++t) {
    const int rb = t / ngroups, grp = t - rb * ngroups;
    const int f0 = rb * 16;
    if (rb != cur_rb) {
      cur_rb = rb;
      // all loads first (they are independent), the arithmetic afterwards
      const int fa = f0 + c;
      double ra[KS], rbv[CBW][2][KS], rc[CBW][2][4], sf[4], sg[CBW][2];
      const double sfa = fa < F ? sscale[fa] : 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int p = 4 * s + q;
        ra[s] = (fa < F && p < k) ? comp[(int64_t)p * F + fa] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) sf[i] = (f0 + q + 4 * i < F) ? sscale[f0 + q + 4 * i] : 0.0;
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int g = gbase + 32 * cb + 2 * c + tt;
          sg[cb][tt] = g < F ? sscale[g] : 0.0;
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const int p = 4 * s + q;
            rbv[cb][tt][s] = (g < F && p < k) ? comp[(int64_t)p * F + g] : 0.0;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int f = f0 + q + 4 * i;
            rc[cb][tt][i] = (f < F && g < F) ? cun[(int64_t)f * F + g] : 0.0;
          }
        }
#pragma unroll
      for (int s = 0; s < KS; ++s) ua[s] = ra[s] * sfa;
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
          for (int s = 0; s < KS; ++s) ub[cb][tt][s] = rbv[cb][tt][s] * sg[cb][tt];
#pragma unroll
          for (int i = 0; i < 4; ++i) ci[cb][tt][i] = rc[cb][tt][i] * inv_ndiv * (sf[i] * sg[cb][tt]);
        }
    }
    // this item's variances, fetched while the previous item was being processed (a global load here would sit on the
    // critical path of every item); the previous item's have been consumed: every wave has passed its last barrier
    const int64_t bb0 = (int64_t)grp * PM_NB;
    if ((int)threadIdx.x < PM_NB * 4 * KS) vstage[threadIdx.x] = vnext;
    __syncthreads();
    if (t + 1 < t1 && (int)threadIdx.x < PM_NB * 4 * KS) {
      const int grp1 = (t + 1) % ngroups, ib = threadIdx.x / (4 * KS), p = threadIdx.x - ib * 4 * KS;
      const int64_t b = (int64_t)grp1 * PM_NB + ib;
      vnext = (b < B && p < k) ? gp_var_of(gp, b, p, k) : 0.0;
    }
    const int nflat = (F - f0 < 16 ? F - f0 : 16) * F;           // doubles of the block that exist
#pragma unroll 1
    for (int ib = 0; ib < PM_NB; ++ib) {
      const int64_t b = bb0 + ib;
      if (b >= B) break;
      double a[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) a[s] = ua[s] * vstage[ib * 4 * KS + 4 * s + q];
      // One instruction stream per wave, issued in order: an MFMA occupies the SIMD's matrix pipe for 64 cycles, so
      // everything else is placed BETWEEN the MFMAs (the scheduling barriers keep the compiler from regrouping):
      // the LDS reads of the previous sample's block, its 1 KiB-per-instruction global stores, and the LDS writes
      // of this sample's first column block all issue in the shadow of the matrix pipe.
      const double *src = smem + (buf ^ 1) * tile_sz;
      double *tile = smem + buf * tile_sz;
      constexpr int NP = 8 * CBW / 2;        // 16-byte pieces per thread and block: 8 (512 threads) or 4 (1024)
      d2x st8[4];                            // four 16-byte pieces in flight between LDS and memory
      d4x d[CBW][2];
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) { d[cb][0] = ci[cb][0]; d[cb][1] = ci[cb][1]; }
      const int e0 = 2 * (int)threadIdx.x;
#define PM_MFMA(CB, T, S) do { if (DBG == 4) d[CB][T][0] += a[S] * ub[CB][T][S]; else d[CB][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[S], ub[CB][T][S], d[CB][T], 0, 0, 0); } while (0)
#define PM_READ(J) st8[(J) & 3] = *reinterpret_cast<const d2x *>(src + e0 + 2 * NT * (J))   /* unconditional: stays inside the LDS allocation */
#define PM_STORE(J) do { if (e0 + 2 * NT * (J) < pend_n) *reinterpret_cast<d2x *>(pend_dst + e0 + 2 * NT * (J)) = st8[(J) & 3]; } while (0)
#define PM_WRITE(CB, I) do { const int g_ = gbase + 32 * (CB) + 2 * c; if (DBG == 3) { dbg_acc += d[CB][0][I] + d[CB][1][I]; } else if (g_ < F) *reinterpret_cast<d2x *>(tile + (q + 4 * (I)) * F + g_) = d2x{d[CB][0][I], d[CB][1][I]}; } while (0)
#define PM_FENCE() __builtin_amdgcn_sched_barrier(0)
      if (DBG == 1) {
#pragma unroll
        for (int s = 0; s < KS; ++s) { d[0][0][0] += a[s]; d[CBW - 1][1][0] += ub[0][0][s] + ub[CBW - 1][1][s]; }
#pragma unroll
        for (int h = 0; h < NP / 4; ++h) {
#pragma unroll
          for (int j = 0; j < 4; ++j) PM_READ(4 * h + j);
#pragma unroll
          for (int j = 0; j < 4; ++j) PM_STORE(4 * h + j);
        }
      } else {
        // MFMA mi of 2 CBW KS (column block 0 first); fillers: pieces 4h..4h+3 are read after MFMAs 4h, 4h + 1 and
        // stored after 4h + 2, 4h + 3; column block 0's LDS writes (CBW = 2) after MFMAs WR0.. (its last MFMA is
        // 2 KS - 1)
        constexpr int NM = 2 * CBW * KS;
        constexpr int WR0 = (2 * KS + 1 > NP) ? 2 * KS + 1 : NP;
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) {
          const int cb = mi / (2 * KS), s = (mi % (2 * KS)) / 2, tt = mi % 2;
          PM_MFMA(cb, tt, s);
          PM_FENCE();
          if (mi < NP) {
            const int h = mi / 4, r = mi % 4;      // half h: reads at r = 0, 1, stores at r = 2, 3
            if (r < 2) { PM_READ(4 * h + 2 * r); PM_READ(4 * h + 2 * r + 1); }
            else { PM_STORE(4 * h + 2 * (r - 2)); PM_STORE(4 * h + 2 * (r - 2) + 1); }
          } else if (CBW == 2 && mi >= WR0 && mi < WR0 + 4) { PM_WRITE(0, mi - WR0); }
          PM_FENCE();
        }
        // what did not fit between the MFMAs (few k-steps), in the same order
#pragma unroll
        for (int sl = 0; sl < NP; ++sl) {
          if (sl >= NM) {
            const int h = sl / 4, r = sl % 4;
            if (r < 2) { PM_READ(4 * h + 2 * r); PM_READ(4 * h + 2 * r + 1); }
            else { PM_STORE(4 * h + 2 * (r - 2)); PM_STORE(4 * h + 2 * (r - 2) + 1); }
          }
        }
        if (CBW == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (WR0 + i >= NM) PM_WRITE(0, i);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) PM_WRITE(CBW - 1, i);
#undef PM_MFMA
#undef PM_READ
#undef PM_STORE
#undef PM_WRITE
#undef PM_FENCE
      if (DBG != 3) __syncthreads();
      pend_dst = cov + (int64_t)b * F * F + (int64_t)f0 * F;
      pend_n = DBG >= 2 ? 0 : nflat;
      buf ^= 1;
    }
  }
  {
    const double *src = smem + (buf ^ 1) * tile_sz;
    for (int e = 2 * (int)threadIdx.x; e < pend_n; e += 2 * NT)
      *reinterpret_cast<d2x *>(pend_dst + e) = *reinterpret_cast<const d2x *>(src + e);
  }
  if (DBG >= 2 && smem[lane] + dbg_acc == -1.2345e300) pend_dst[0] = 0.0;
}

// central value (ref: emulation.py:508-509) for the matrix-core writer: cv[b][g] = (sum_p mean[b][p] S[p][g]) s_g + mean_g.
// One workgroup per CV_NB samples: their k GP means are summed from the partial sums ONCE (CV_NB k threads, contiguous
// reads) into LDS, then every thread forms its features' values from them.  Round 3's form had every thread of every
// workgroup re-sum all k x nchunk partials of its samples: 33 us for 4 MB of output.
constexpr int CV_NB = 4;
__global__ __launch_bounds__(256) void central_value_kernel(GpParts gp, const double *__restrict__ comp,
                                                            const double *__restrict__ smean, const double *__restrict__ sscale,
                                                            double *__restrict__ cv, int64_t B, int F, int k) {
  __shared__ double s_mu[CV_NB][16];
  const int64_t b0 = (int64_t)blockIdx.x * CV_NB;
  if ((int)threadIdx.x < CV_NB * 16) {
    const int ib = threadIdx.x >> 4, p = threadIdx.x & 15;
    s_mu[ib][p] = (b0 + ib < B && p < k) ? gp_mean_of(gp, b0 + ib, p, k) : 0.0;
  }
  // this thread's features: their components, scale and mean (independent of the sums above: loaded meanwhile)
  double cg[2][16], sg[2], mg[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int g = threadIdx.x + 256 * t;
    sg[t] = g < F ? sscale[g] : 0.0;
    mg[t] = g < F ? smean[g] : 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) cg[t][p] = (g < F && p < k) ? comp[(int64_t)p * F + g] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int ib = 0; ib < CV_NB; ++ib) {
    if (b0 + ib >= B) break;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int g = threadIdx.x + 256 * t;
      double sacc = 0.0;
#pragma unroll
      for (int p = 0; p < 16; ++p)
        if (p < k) sacc = fma(s_mu[ib][p], cg[t][p], sacc);
      if (g < F) cv[(b0 + ib) * F + g] = sacc * sg[t] + mg[t];
    }
  }
}

int launch_predict_full(gpemu_model *m, int64_t B, double n_div, double *dcv, double *dcov, hipStream_t st,
                        const double *dmean, const double *dvar) {
  const int F = (int)m->F, k = (int)m->k;
  dim3 grid((unsigned)((F + PF_TG - 1) / PF_TG), (unsigned)((F + PF_TF - 1) / PF_TF),
            (unsigned)((B + PF_NB - 1) / PF_NB));
  size_t shm = sizeof(double) * (size_t)(k * (PF_TF + PF_TG) + k);
  // 16-byte stores need every row start of dcov 16-byte aligned: F even and an aligned base
  const bool pair = F % 2 == 0 && (reinterpret_cast<uintptr_t>(dcov) & 15) == 0;
  const size_t shm_rows = sizeof(double) * (size_t)(k * (PR_ROWS + PR_COLS) + k * PF_NB);
  const bool no_mfma = getenv("GPEMU_PREDICT_VALU") != nullptr;   // the VALU writer, for comparison (read per call)
  const GpParts parts{dmean, dvar, m->ws.mean_part, m->ws.vsq_part, m->kdiag, m->ws.cur_nchunk, m->ws.cur_nrb};
  if (pair && k <= 16 && F <= 512 && !no_mfma) {
    const int nrow = (F + 15) / 16, ngroups = (int)((B + PM_NB - 1) / PM_NB), total = nrow * ngroups;
    const int ncu = m->num_cu;
    const int workers = std::min(ncu, total);
    const size_t shm_pm = sizeof(double) * (2 * 16 * (size_t)F + PM_NB * 16);
    // 16 waves x 32 columns (CBW = 2, 8 waves x 64 columns, measured within 1 % on the whole chip: not instantiated)
#define GP_LAUNCH_PM2(KSV, CB, DB)                                                                                 \
  do {                                                                                                             \
    static bool attr_set = false;                                                                                  \
    if (!attr_set) {                                                                                               \
      GP_HIP(hipFuncSetAttribute((const void *)predict_cov_mfma_kernel<KSV, CB, DB>,                               \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));                        \
      attr_set = true;                                                                                             \
    }                                                                                                              \
    hipLaunchKernelGGL((predict_cov_mfma_kernel<KSV, CB, DB>), dim3((unsigned)workers), dim3(1024 / CB), shm_pm,   \
                       st, parts, m->comp, m->sscale, m->cunexpl, dcov, B, F, k, 1.0 / n_div, ngroups, total);     \
  } while (0)
#define GP_LAUNCH_PM(KSV, DB) GP_LAUNCH_PM2(KSV, 1, DB)
    // the central values first (a few us, their inputs -- the GP stage's partial sums -- still in cache), then the writer
    hipLaunchKernelGGL(central_value_kernel, dim3((unsigned)((B + CV_NB - 1) / CV_NB)), dim3(256), 0, st, parts, m->comp,
                       m->smean, m->sscale, dcv, B, F, k);
    if (k <= 4) GP_LAUNCH_PM(1, 0);
    else if (k <= 8) GP_LAUNCH_PM(2, 0);
    else if (k <= 12) GP_LAUNCH_PM(3, 0);
    else GP_LAUNCH_PM(4, 0);
#undef GP_LAUNCH_PM2
#undef GP_LAUNCH_PM
    GP_HIP(hipGetLastError());
    return GPEMU_OK;
  }
  if (!dmean || !dvar) {         // the VALU forms read arrays
    const int rc_red = launch_reduce_mean_var(m, B, m->ws.mean, m->ws.var, st);
    if (rc_red != GPEMU_OK) return rc_red;
    dmean = m->ws.mean;
    dvar = m->ws.var;
  }
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
