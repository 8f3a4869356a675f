// GP predictive mean / variance for a batch of parameter points, all PCs of one emulation group.
//
// Replaces  ref: emulation.py:494-499  ->  skl _gpr.py:441-494  (GaussianProcessRegressor.predict
// with return_std=True, k times):
//     K_* = kernel_(X, X_train);  mean = K_* . alpha_;  V = L_^-1 K_*^T;  var = diag - sum_i V_i^2
// Device form: W = L_^-1 is formed once at model creation, so V = W K_*^T is a lower-triangular
// GEMM on the f64 matrix cores (v_mfma_f64_16x16x4_f64) with the column sum of squares fused into
// the epilogue -- V is never written to memory.
//
//   kstar_kernel      K_*^T[p][j][b] (HBM/L2 workspace) + partial means   (VALU f64, HBM-write)
//   trmm_vsq_*        sum_i (W_p K_*^T)[i][b]^2 per 64-row block           (MFMA f64)
//       _dma_kernel         B > 256: persistent, LDS-direct loads, XCD-aware LPT schedule (the default)
//       _persistent_kernel  same schedule, register-staged loads (GPEMU_TRMM_NO_DMA=1)
//       _kernel             one item per workgroup (GPEMU_TRMM_SIMPLE=1)
//       _smallb_kernel      B <= 128: 32 x 64 items, K split inside the workgroup
//   reduce_kernel     sums the partials, var = kdiag - vsq, clip, std^2
#include <algorithm>

#include "internal.h"
#include "predict_dev.h"

namespace gpemu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// base kernel value from the squared scaled distance r2 (skl kernels.py:1564-1565, 1715-1733)
template <int KIND>  // 0 rbf, 1 matern 0.5, 2 matern 1.5, 3 matern 2.5
__device__ __forceinline__ double base_kernel(double r2) {
  if (KIND == 0) return exp(-0.5 * r2);
  double r = sqrt(r2);
  if (KIND == 1) return exp(-r);
  if (KIND == 2) {
    double t = r * 1.7320508075688772;  // math.sqrt(3)
    return (1.0 + t) * exp(-t);
  }
  double t = r * 2.23606797749979;  // math.sqrt(5)
  return (1.0 + t + t * t / 3.0) * exp(-t);
}

// grid (Bcap/64, Npad/JCHUNK, k), block 256: lane <-> query b, wave <-> 32 training rows.
template <int KIND, int RPW>   // RPW = training rows per wave (32: large batches; 8: small batches, 4x the workgroups)
__global__ __launch_bounds__(256) void kstar_kernel(
    double *__restrict__ Xq, const double *__restrict__ Xs, const double *__restrict__ inv_ls,
    const double *__restrict__ constv, const double *__restrict__ alpha, double *__restrict__ KS,
    double *__restrict__ mean_part, int64_t N, int64_t Npad, int64_t Bcap, int has_const, ProposeArgs pa) {
  __shared__ double s_tab[32];
  __shared__ __attribute__((aligned(16))) double s_xs[4 * RPW * DPAD];   // this workgroup's training rows (scaled)
  __shared__ double s_al[4 * RPW];                                       // and their alpha
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = blockIdx.z;
  const int chunk = blockIdx.y;
  const int nchunk = gridDim.y;
  const int64_t b = (int64_t)blockIdx.x * 64 + lane;
  const int64_t jb0 = (int64_t)chunk * (4 * RPW);

  // issue the loads of the workgroup's rows first (contiguous in memory): their latency overlaps with the
  // dependent index -> position chain of the proposal below
  constexpr int NPAIR = 4 * RPW * DPAD / 2;            // 16-byte pairs to stage
  constexpr int PER_T = (NPAIR + 255) / 256;
  d2 stage[PER_T];
  const d2 *xsrc = reinterpret_cast<const d2 *>(Xs + ((int64_t)p * Npad + jb0) * DPAD);
#pragma unroll
  for (int t = 0; t < PER_T; ++t) {
    const int idx = threadIdx.x + 256 * t;
    stage[t] = (idx < NPAIR) ? xsrc[idx] : d2{0.0, 0.0};
  }
  double al_stage = 0.0;
  if (threadIdx.x < 4 * RPW) al_stage = alpha[(int64_t)p * Npad + jb0 + threadIdx.x];

  double xq[DPAD];
  if (pa.enabled) {
    // stretch-move proposal for column b (every workgroup recomputes it; one of them stores it)
    double z = 1.0;
    int w = 0, j = 0;
    const bool live = b < pa.n;
    if (live) { w = pa.idx_s[b]; j = pa.partner[b]; z = pa.zz[b]; }
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) {
      double v = 0.0;
      if (live && dd < pa.d) {
        const double cj = pa.X[(int64_t)j * DPAD + dd], sw = pa.X[(int64_t)w * DPAD + dd];
        v = cj - (cj - sw) * z;
      }
      xq[dd] = v;
    }
    if (chunk == 0 && p == 0 && wave == 0) {
#pragma unroll
      for (int dd = 0; dd < DPAD; ++dd) Xq[b * DPAD + dd] = xq[dd];
      if (live) pa.factors[b] = (pa.d - 1.0) * log(z);
    }
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) xq[dd] = xq[dd] * inv_ls[p * DPAD + dd];
  } else if (pa.raw) {
    // caller rows [n][d]: padded on the fly, the padded row stored once
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) xq[dd] = (b < pa.n && dd < pa.d) ? pa.raw[b * pa.d + dd] : 0.0;
    if (chunk == 0 && p == 0 && wave == 0) {
#pragma unroll
      for (int dd = 0; dd < DPAD; ++dd) Xq[b * DPAD + dd] = xq[dd];
    }
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) xq[dd] = xq[dd] * inv_ls[p * DPAD + dd];
  } else {
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) xq[dd] = Xq[b * DPAD + dd] * inv_ls[p * DPAD + dd];
  }
  const double c = has_const ? constv[p] : 0.0;

  if (threadIdx.x < 32) s_tab[threadIdx.x] = c_exp2_32[threadIdx.x];
#pragma unroll
  for (int t = 0; t < PER_T; ++t) {
    const int idx = threadIdx.x + 256 * t;
    if (idx < NPAIR) reinterpret_cast<d2 *>(s_xs)[idx] = stage[t];
  }
  if (threadIdx.x < 4 * RPW) s_al[threadIdx.x] = al_stage;
  __syncthreads();

  const int64_t jbase = jb0 + wave * RPW;
  const double *xs = s_xs + wave * RPW * DPAD;       // wave-uniform LDS addresses: broadcast reads
  const double *al = s_al + wave * RPW;
  double *ks = KS + ((int64_t)p * Npad + jbase) * Bcap + b;
  double macc = 0.0;
#pragma unroll 4
  for (int jj = 0; jj < RPW; ++jj) {
    double r2 = 0.0;
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) {
      double df = xq[dd] - xs[jj * DPAD + dd];
      r2 = fma(df, df, r2);
    }
    double v = base_kernel_fast<KIND>(r2, s_tab) + c;
    if (jbase + jj >= N) v = 0.0;  // padded training rows contribute nothing
    ks[(int64_t)jj * Bcap] = v;
    macc = fma(al[jj], v, macc);
  }
  __shared__ double red[4][64];
  red[wave][lane] = macc;
  __syncthreads();
  if (wave == 0) {
    double s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    mean_part[(b * gridDim.z + p) * nchunk + chunk] = s;
  }
}

int launch_kstar(gpemu_model *m, int64_t B, double *dXq, hipStream_t st, const ProposeArgs *pa) {
  const ProposeArgs pargs = pa ? *pa : ProposeArgs();
  // only the column tiles that hold real queries; the rows of dXq up to round_up(B, TILE) must be
  // finite (pad_queries_kernel / the sampler's proposal buffer zero them)
  Workspace &w = m->ws;
  const int64_t Bv = m->variant_B > 0 ? m->variant_B : B;   // a chain stacked with others is evaluated as it would be alone
  const bool small = Bv <= 256;                      // few columns: more, shorter workgroups
  static const int big_rpw = getenv("GPEMU_KSTAR_RPW") ? atoi(getenv("GPEMU_KSTAR_RPW")) : 32;
  const int rows_per_wg = small ? 32 : 4 * big_rpw;
  w.cur_nchunk = (int)(m->Npad / rows_per_wg);
  // column blocks of 64 queries: whole 128-column tiles for the triangular GEMM, except that a batch of at most
  // 64 (always served by the small-batch kernel's 64-column items) needs only its first block
  const int64_t ncols = (B <= 64) ? 64 : round_up(B, TILE);
  dim3 grid((unsigned)(ncols / 64), (unsigned)w.cur_nchunk, (unsigned)m->k), block(256);
  const int pe0 = prof_mark(m, st);
  int kind = 0;
  if (m->kernel_kind == GPEMU_KERNEL_MATERN) kind = (m->nu == 0.5) ? 1 : (m->nu == 1.5 ? 2 : 3);
#define GP_LAUNCH_KSTAR(KD)                                                                              \
  do {                                                                                                   \
    if (small)                                                                                           \
      hipLaunchKernelGGL((kstar_kernel<KD, 8>), grid, block, 0, st, dXq, m->Xs, m->inv_ls, m->constv, m->alpha, \
                         w.KS, w.mean_part, m->N, m->Npad, w.Bcap, m->has_const, pargs);                  \
    else if (big_rpw == 16)                                                                              \
      hipLaunchKernelGGL((kstar_kernel<KD, 16>), grid, block, 0, st, dXq, m->Xs, m->inv_ls, m->constv, m->alpha, \
                         w.KS, w.mean_part, m->N, m->Npad, w.Bcap, m->has_const, pargs);                  \
    else if (big_rpw == 8)                                                                               \
      hipLaunchKernelGGL((kstar_kernel<KD, 8>), grid, block, 0, st, dXq, m->Xs, m->inv_ls, m->constv, m->alpha, \
                         w.KS, w.mean_part, m->N, m->Npad, w.Bcap, m->has_const, pargs);                  \
    else                                                                                                 \
      hipLaunchKernelGGL((kstar_kernel<KD, 32>), grid, block, 0, st, dXq, m->Xs, m->inv_ls, m->constv, m->alpha, \
                         w.KS, w.mean_part, m->N, m->Npad, w.Bcap, m->has_const, pargs);                  \
  } while (0)
  switch (kind) {
    case 0: GP_LAUNCH_KSTAR(0); break;
    case 1: GP_LAUNCH_KSTAR(1); break;
    case 2: GP_LAUNCH_KSTAR(2); break;
    default: GP_LAUNCH_KSTAR(3); break;
  }
#undef GP_LAUNCH_KSTAR
  GP_HIP(hipGetLastError());
  prof_pair(m, 1, pe0, prof_mark(m, st));
  return GPEMU_OK;
}

// ------------------------------------------------------------------------------------------
// Triangular GEMM with fused column sum-of-squares.
//   V[i][b] = sum_{j<=i} Wt[j][i] * KS[j][b]      (per PC p; both operands are k-major)
//   out[b][p][rb] = sum_{i in 64-row block rb} V[i][b]^2
// MFMA v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][n = l&15];
// D[reg] is row (l>>4) + 4*reg, column l&15 (verified by tools/mfma_f64_peak).
//
// One 512-thread workgroup per CU: 8 waves = 2 per SIMD, arranged 2 (M) x 4 (N), 32 x 32 per wave
// (2 x 2 MFMA tiles, 16 accumulator registers) on a 64 x 128 tile with K step 32, LDS double
// buffered (112 KiB, which also pins residency at one workgroup per CU).  Work items (row block, PC,
// column block) are ordered by decreasing K extent, so the hardware dispatcher hands heavy items out
// first and light ones fill the tail.  Measured history (C3, B = 512): 128x128 tiles / 4 waves
// 182 us (accumulators bounced through AGPRs, 1 wave/SIMD); 64x128 / 2 workgroups per CU 123 us;
// this kernel 117 us; of that ~98 us is the bare MFMA stream of the same item schedule.
constexpr int TM = 64;      // rows per item
constexpr int KT = 32;      // K step
constexpr int LSTRA = 80;   // LDS row strides in doubles: (2 * stride) % 64 == 32, so the two 16-lane
constexpr int LSTRB = 144;  // groups of a half-wave hit disjoint 32-bank windows (ds_read_b64)

__global__ __launch_bounds__(512, 2) void trmm_vsq_kernel(const double *__restrict__ Wt,
                                                          const double *__restrict__ KS,
                                                          double *__restrict__ out, int64_t Npad,
                                                          int64_t Bcap, int k, int nrb, int ncb) {
  __shared__ __attribute__((aligned(16))) double sA[2][KT][LSTRA];
  __shared__ __attribute__((aligned(16))) double sB[2][KT][LSTRB];
  __shared__ double red[2][TILE];

  const int ncombo = k * ncb;
  const int rbi = blockIdx.x / ncombo;
  const int combo = blockIdx.x - rbi * ncombo;
  const int p = combo / ncb;
  const int cb = combo - p * ncb;
  const int rb = nrb - 1 - rbi;  // heavy (long K) row blocks first
  const int64_t i0 = (int64_t)rb * TM, b0 = (int64_t)cb * TILE;
  const int ntile = (int)((i0 + TM + KT - 1) / KT);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lk = lane >> 4;
  // staging: A tile 32 x 64 doubles = 1024 x 16 B (2 per thread), B tile 32 x 128 = 2048 x 16 B (4);
  // every wave-load is one contiguous 512 B / 1 KiB row segment
  const int arow = tid >> 5, ac2 = tid & 31;   // rows arow, arow + 16
  const int brow = tid >> 6, bc2 = tid & 63;   // rows brow + 8 r
  const double *pa = Wt + (int64_t)p * Npad * Npad + i0 + (int64_t)arow * Npad + 2 * ac2;
  const double *pb = KS + (int64_t)p * Npad * Bcap + b0 + (int64_t)brow * Bcap + 2 * bc2;
  const int64_t astep = (int64_t)KT * Npad, bstep = (int64_t)KT * Bcap;

  d2 ra[2], rbv[4];
  auto gload = [&]() {
    ra[0] = *reinterpret_cast<const d2 *>(pa);
    ra[1] = *reinterpret_cast<const d2 *>(pa + 16 * Npad);
#pragma unroll
    for (int r = 0; r < 4; ++r) rbv[r] = *reinterpret_cast<const d2 *>(pb + (int64_t)(8 * r) * Bcap);
    pa += astep;
    pb += bstep;
  };
  auto sstore = [&](int buf) {
    *reinterpret_cast<d2 *>(&sA[buf][arow][2 * ac2]) = ra[0];
    *reinterpret_cast<d2 *>(&sA[buf][arow + 16][2 * ac2]) = ra[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<d2 *>(&sB[buf][brow + 8 * r][2 * bc2]) = rbv[r];
  };

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  gload();
  sstore(0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntile) gload();
    // operand fragments of k-step ks+1 are read from LDS before the MFMAs of k-step ks are issued
    double a[2][2], b[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[0][mi] = sA[buf][lk][wm * 32 + mi * 16 + lr];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) b[0][ni] = sB[buf][lk][wn * 32 + ni * 16 + lr];
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < KT / 4) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[nxt][mi] = sA[buf][(ks + 1) * 4 + lk][wm * 32 + mi * 16 + lr];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[nxt][ni] = sB[buf][(ks + 1) * 4 + lk][wn * 32 + ni * 16 + lr];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][mi], b[cur][ni], acc[mi][ni], 0, 0, 0);
    }
    if (t + 1 < ntile) sstore(buf ^ 1);
    __syncthreads();
  }

  // column sums of V^2 over this wave's 32 rows x 32 columns, then over the two row halves
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    double s = 0.0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) s = fma(acc[mi][ni][r], acc[mi][ni][r], s);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lk == 0) red[wm][wn * 32 + ni * 16 + lr] = s;
  }
  __syncthreads();
  if (tid < TILE) out[((b0 + tid) * k + p) * nrb + rb] = red[0][tid] + red[1][tid];
}

// ------------------------------------------------------------------------------------------
// Persistent form of the same tile: one workgroup per CU walks a host-built list of items (LPT
// schedule over the known K extents, so the triangular work is balanced to within one small item)
// as ONE software-pipelined stream of k-tiles:
//   iteration s:  global loads of k-tile s+2 -> registers (two register sets)
//                 MFMAs of k-tile s from LDS buffer s&1, with the LDS stores of k-tile s+1
//                 (loaded during iteration s-1) interleaved between the MFMA groups
//                 one barrier
// so neither the global-load latency nor the LDS store pass (48 KiB per k-tile at ~80 B/clk) nor an
// item's prologue sits on the MFMA critical path.  Items of the first quarter of the rows (short K)
// are split into two 64-column halves to give the schedule small pieces for its tail.
struct TrmmItem {
  int p, rb, col0, half;  // PC, 64-row block, first column, 1 = 64-column item
};
constexpr int TRMM_MAX_ITEMS = 64;  // per worker; the schedule falls back to more workers' worth otherwise
#ifdef GPEMU_TRMM_STAMPS
__device__ unsigned long long g_trmm_stamps[512 * 8];
#endif

__global__ __launch_bounds__(512, 2) void trmm_vsq_persistent_kernel(
    const double *__restrict__ Wt, const double *__restrict__ KS, double *__restrict__ out,
    const TrmmItem *__restrict__ sched, const int *__restrict__ sched_cnt, int max_items,
    int64_t Npad, int64_t Bcap, int k, int nrb) {
  __shared__ __attribute__((aligned(16))) double sA[2][KT][LSTRA];
  __shared__ __attribute__((aligned(16))) double sB[2][KT][LSTRB];
  __shared__ double red[2][TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lk = lane >> 4;
  const int arow = tid >> 5, ac2 = tid & 31;   // A (and half-width B) staging: rows arow, arow + 16
  const int brow = tid >> 6, bc2 = tid & 63;   // full-width B staging: rows brow + 8 r
  const int64_t astep = (int64_t)KT * Npad, bstep = (int64_t)KT * Bcap;

  // this worker's item list, copied to LDS once: reading a descriptor from global memory at an item
  // switch would sit behind the tile loads in flight (vmcnt is in order) and stall the stream
  __shared__ TrmmItem s_items[TRMM_MAX_ITEMS];
  const int nitems = sched_cnt[blockIdx.x];
  if (tid < nitems) s_items[tid] = sched[(int64_t)blockIdx.x * max_items + tid];
  __syncthreads();
  const TrmmItem *my = s_items;

  // ---- load cursor (runs two k-tiles ahead of the compute cursor) ----
  // Every call issues exactly six 16-byte loads and every staged k-tile is stored with exactly six
  // LDS stores, unconditionally: with a fixed number of memory operations per iteration hipcc can
  // count s_waitcnt vmcnt(N) precisely and the stores of k-tile s+1 wait only for ITS loads, not for
  // the loads of k-tile s+2 issued just before.  Past the end of the list the cursor re-reads its
  // last k-tile (harmless), and 64-column items stage the whole 128-column tile they live in.
  int l_item = 0, l_t = 0, l_nt = 1;
  const double *l_pa = Wt + (int64_t)arow * Npad + 2 * ac2;
  const double *l_pb = KS + (int64_t)brow * Bcap + 2 * bc2;
  auto l_open = [&]() {
    const TrmmItem it = my[l_item];
    const int64_t i0 = (int64_t)it.rb * TM;
    l_nt = (int)((i0 + TM + KT - 1) / KT);
    l_pa = Wt + (int64_t)it.p * Npad * Npad + i0 + (int64_t)arow * Npad + 2 * ac2;
    l_pb = KS + (int64_t)it.p * Npad * Bcap + (it.col0 & ~(TILE - 1)) + (int64_t)brow * Bcap + 2 * bc2;
    l_t = 0;
  };
  struct Stage {
    d2 a[2], b[4];
  };
  Stage st0, st1;
  auto gload = [&](Stage &sg) {
    sg.a[0] = *reinterpret_cast<const d2 *>(l_pa);
    sg.a[1] = *reinterpret_cast<const d2 *>(l_pa + 16 * Npad);
#pragma unroll
    for (int r = 0; r < 4; ++r) sg.b[r] = *reinterpret_cast<const d2 *>(l_pb + (int64_t)(8 * r) * Bcap);
    if (l_item < nitems) {
      if (++l_t == l_nt) {
        if (++l_item < nitems) l_open();   // else: stay on the last k-tile
      } else {
        l_pa += astep;
        l_pb += bstep;
      }
    }
  };
  // one of the six 16-byte LDS stores of a staged k-tile (part = 0..5, compile-time after unrolling)
  auto sstore_part = [&](const Stage &sg, int buf, int part) {
    if (part == 0) *reinterpret_cast<d2 *>(&sA[buf][arow][2 * ac2]) = sg.a[0];
    else if (part == 1) *reinterpret_cast<d2 *>(&sA[buf][arow + 16][2 * ac2]) = sg.a[1];
    else *reinterpret_cast<d2 *>(&sB[buf][brow + 8 * (part - 2)][2 * bc2]) = sg.b[part - 2];
  };

#ifdef GPEMU_TRMM_STAMPS
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
  if (tid == 0) g_trmm_stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
  if (nitems == 0) return;
  l_open();
  // prologue: k-tile 0 -> LDS buffer 0, k-tile 1 -> register set st1
  gload(st0);
#pragma unroll
  for (int part = 0; part < 6; ++part) sstore_part(st0, 0, part);
  gload(st1);
  __syncthreads();

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  int c_item = 0, c_t = 0;
  TrmmItem cur = my[0];
  int c_nt = (int)(((int64_t)cur.rb * TM + TM + KT - 1) / KT);

  // one k-tile: loads of k-tile +2 into `snew`, MFMAs from LDS buffer `buf`, LDS stores of `sold`
  // (k-tile +1) into the other buffer; returns true when the worker's last item is finished
  auto step = [&](int buf, Stage &snew, const Stage &sold) -> bool {
    gload(snew);
    if (cur.half) {
      double a[2][2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[0][mi] = sA[buf][lk][wm * 32 + mi * 16 + lr];
      const int hc = (cur.col0 & 64) + wn * 16 + lr;
      b[0] = sB[buf][lk][hc];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (ks + 1 < KT / 4) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) a[nx][mi] = sA[buf][(ks + 1) * 4 + lk][wm * 32 + mi * 16 + lr];
          b[nx] = sB[buf][(ks + 1) * 4 + lk][hc];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the next k-step's LDS reads ahead of these MFMAs
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cu][mi], b[cu], acc[mi][0], 0, 0, 0);
        if (ks >= 1 && ks <= 6) sstore_part(sold, buf ^ 1, ks - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      double a[2][2], b[2][2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[0][mi] = sA[buf][lk][wm * 32 + mi * 16 + lr];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[0][ni] = sB[buf][lk][wn * 32 + ni * 16 + lr];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (ks + 1 < KT / 4) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) a[nx][mi] = sA[buf][(ks + 1) * 4 + lk][wm * 32 + mi * 16 + lr];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) b[nx][ni] = sB[buf][(ks + 1) * 4 + lk][wn * 32 + ni * 16 + lr];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the next k-step's LDS reads ahead of these MFMAs
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cu][mi], b[cu][ni], acc[mi][ni], 0, 0, 0);
        if (ks >= 1 && ks <= 6) sstore_part(sold, buf ^ 1, ks - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (++c_t == c_nt) {
      // item finished: column sums of V^2 over this wave's rows, then over the two row halves
      const int ncols = cur.half ? 64 : TILE;
      const int wcol = cur.half ? wn * 16 : wn * 32;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (ni == 0 || !cur.half) {
          double sq = 0.0;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq = fma(acc[mi][ni][r], acc[mi][ni][r], sq);
          sq += __shfl_xor(sq, 16);
          sq += __shfl_xor(sq, 32);
          if (lk == 0) red[wm][wcol + ni * 16 + lr] = sq;
        }
      }
      __syncthreads();
      if (tid < ncols)
        out[(((int64_t)cur.col0 + tid) * k + cur.p) * nrb + cur.rb] = red[0][tid] + red[1][tid];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};
#ifdef GPEMU_TRMM_STAMPS
      if (tid == 0 && c_item + 1 < 7) g_trmm_stamps[blockIdx.x * 8 + 1 + c_item] = __builtin_amdgcn_s_memrealtime();
      if (tid == 0 && c_item + 1 == nitems) g_trmm_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - clk0;
#endif
      if (++c_item == nitems) return true;
      cur = my[c_item];
      c_nt = (int)(((int64_t)cur.rb * TM + TM + KT - 1) / KT);
      c_t = 0;
      // `red` is rewritten only after at least one more barrier (the next k-tile's)
    }
    return false;
  };
  for (;;) {
    if (step(0, st0, st1)) break;
    if (step(1, st1, st0)) break;
  }
}


// ------------------------------------------------------------------------------------------
// LDS-direct variant of the persistent kernel: the k-tiles are streamed from L2 into LDS by
// global_load_lds_dwordx4 (each wave-load lands as one contiguous 1 KiB chunk, lane l -> bytes 16 l .. 16 l + 15),
// so there is no register staging and no ds_write on the waves' instruction streams.  The LDS tiles are
// unpadded, k-major ([k][64] for W, [k][128] for K_*^T); bank conflicts of the fragment reads are avoided by
// storing the odd k-rows with their 16-double blocks swapped (position = index ^ 16) -- the swizzle is applied
// on the GLOBAL side (which pair a lane fetches), the LDS side of a DMA load is always contiguous.  Two LDS
// buffers as distinct arrays (so that hipcc's waitcnt insertion can tell the buffer being filled from the
// one being read); three buffers: the loads of k-tile s+2 are issued at the start of k-tile s and the barrier
// that ends k-tile s waits (s_waitcnt vmcnt(6), written by hand) only for the loads of k-tile s+1.
typedef const __attribute__((address_space(1))) void *gas_ptr;
typedef __attribute__((address_space(3))) void *las_ptr;

template <bool INTERLEAVE>   // the six loads of k-tile s+2 go out one by one between the MFMA groups of k-tile s
__global__ __launch_bounds__(512, 2) void trmm_vsq_dma_kernel(
    const double *__restrict__ Wt, const double *__restrict__ KS, double *__restrict__ out,
    const TrmmItem *__restrict__ sched, const int *__restrict__ sched_cnt, int max_items,
    int64_t Npad, int64_t Bcap, int k, int nrb, unsigned long long *__restrict__ stamps) {
  constexpr int BUFD = KT * TM + KT * TILE;            // one k-tile: [W tile | K_*^T tile], 48 KiB
  __shared__ __attribute__((aligned(16))) double L0[BUFD];
  __shared__ __attribute__((aligned(16))) double L1[BUFD];
  __shared__ __attribute__((aligned(16))) double L2[BUFD];
  __shared__ double red[2][TILE];
  __shared__ TrmmItem s_items[TRMM_MAX_ITEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lk = lane >> 4;

  if (stamps && tid == 0) stamps[blockIdx.x * 16] = __builtin_amdgcn_s_memrealtime();
  const int nitems = sched_cnt[blockIdx.x];
  if (tid < nitems) s_items[tid] = sched[(int64_t)blockIdx.x * max_items + tid];
  __syncthreads();
  if (nitems == 0) return;
  const TrmmItem *my = s_items;

  // per-lane source offsets (doubles) of the wave's six 1 KiB chunks of a k-tile
  int offA[2], offB[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kk = 2 * (wave * 2 + j) + (lane >> 5);          // chunk = two k-rows of the W tile
    offA[j] = kk * (int)Npad + ((2 * (lane & 31)) ^ ((kk & 1) << 4));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = wave * 4 + r;                              // chunk = one k-row of the K_*^T tile
    offB[r] = kk * (int)Bcap + ((2 * lane) ^ ((kk & 1) << 4));
  }
  const int64_t astep = (int64_t)KT * Npad, bstep = (int64_t)KT * Bcap;

  // ---- load cursor (one k-tile ahead of the compute cursor) ----
  int l_item = 0, l_t = 0, l_nt = 1;
  const double *l_pa = Wt, *l_pb = KS;
  auto l_open = [&]() {
    const TrmmItem it = my[l_item];
    const int64_t i0 = (int64_t)it.rb * TM;
    l_nt = (int)((i0 + TM + KT - 1) / KT);
    l_pa = Wt + (int64_t)it.p * Npad * Npad + i0;
    l_pb = KS + (int64_t)it.p * Npad * Bcap + (it.col0 & ~(TILE - 1));
    l_t = 0;
  };
  // Issued as inline assembly: with the builtin, hipcc's waitcnt insertion treats every later LDS read as a
  // possible reader of the in-flight destination and drains vmcnt to 0, which defeats the two-tile lookahead;
  // the waits for these loads are the hand-written s_waitcnt vmcnt(6) of tile_barrier().
  auto dma1 = [&](const double *src, double *dst_wave_uniform) {
    const unsigned lds_off = (unsigned)(uintptr_t)((las_ptr)dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off"
                 :
                 : "s"(lds_off), "v"(src)
                 : "memory");
  };
  // one of the wave's six loads of a k-tile (part compile-time after unrolling); the cursor moves on after the last
  auto dma_part = [&](double *dA, int part) {
    double *dB = dA + KT * TM;
    if (part < 2) dma1(l_pa + offA[part], dA + (wave * 2 + part) * 128);
    else dma1(l_pb + offB[part - 2], dB + (wave * 4 + part - 2) * 128);
    if (part == 5 && l_item < nitems) {
      if (++l_t == l_nt) {
        if (++l_item < nitems) l_open();   // else: stay on the last k-tile (harmless re-read)
      } else {
        l_pa += astep;
        l_pb += bstep;
      }
    }
  };
  auto dma = [&](double *dA) {
#pragma unroll
    for (int part = 0; part < 6; ++part) dma_part(dA, part);
  };

  // vmcnt(6) lgkmcnt(0): everything but this wave's six newest loads has landed, every LDS read has returned
  auto tile_barrier = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0076);
    __builtin_amdgcn_s_barrier();
  };
  l_open();
  dma(L0);
  dma(L1);
  tile_barrier();
  if (stamps && tid == 0) stamps[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime();

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  int c_item = 0, c_t = 0;
  TrmmItem cur = my[0];
  int c_nt = (int)(((int64_t)cur.rb * TM + TM + KT - 1) / KT);
  const int sw = (lk & 1) << 4;                                 // this lane's k-rows are odd <=> lk odd
  const int ia0 = lk * TM + ((wm * 32 + lr) ^ sw), ia1 = lk * TM + ((wm * 32 + 16 + lr) ^ sw);

  auto step = [&](const double *cA, double *nA) -> bool {
    const double *cB = cA + KT * TM;
    if (!INTERLEAVE) dma(nA);                                   // k-tile +2 -> the buffer k-tile -1 was read from
    if (cur.half) {
      const int ib = lk * TILE + (((cur.col0 & 64) + wn * 16 + lr) ^ sw);
      double a[2][2], b[2];
      a[0][0] = cA[ia0];
      a[0][1] = cA[ia1];
      b[0] = cB[ib];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (ks + 1 < KT / 4) {
          a[nx][0] = cA[(ks + 1) * 4 * TM + ia0];
          a[nx][1] = cA[(ks + 1) * 4 * TM + ia1];
          b[nx] = cB[(ks + 1) * 4 * TILE + ib];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cu][mi], b[cu], acc[mi][0], 0, 0, 0);
        if (INTERLEAVE && ks < 6) dma_part(nA, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const int ib0 = lk * TILE + ((wn * 32 + lr) ^ sw), ib1 = lk * TILE + ((wn * 32 + 16 + lr) ^ sw);
      double a[2][2], b[2][2];
      a[0][0] = cA[ia0];
      a[0][1] = cA[ia1];
      b[0][0] = cB[ib0];
      b[0][1] = cB[ib1];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (ks + 1 < KT / 4) {
          a[nx][0] = cA[(ks + 1) * 4 * TM + ia0];
          a[nx][1] = cA[(ks + 1) * 4 * TM + ia1];
          b[nx][0] = cB[(ks + 1) * 4 * TILE + ib0];
          b[nx][1] = cB[(ks + 1) * 4 * TILE + ib1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cu][mi], b[cu][ni], acc[mi][ni], 0, 0, 0);
        if (INTERLEAVE && ks < 6) dma_part(nA, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    tile_barrier();
    if (++c_t == c_nt) {
      const int ncols = cur.half ? 64 : TILE;
      const int wcol = cur.half ? wn * 16 : wn * 32;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (ni == 0 || !cur.half) {
          double sq = 0.0;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq = fma(acc[mi][ni][r], acc[mi][ni][r], sq);
          sq += __shfl_xor(sq, 16);
          sq += __shfl_xor(sq, 32);
          if (lk == 0) red[wm][wcol + ni * 16 + lr] = sq;
        }
      }
      __syncthreads();
      if (tid < ncols)
        out[(((int64_t)cur.col0 + tid) * k + cur.p) * nrb + cur.rb] = red[0][tid] + red[1][tid];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};
      if (stamps && tid == 0 && c_item < 13) stamps[blockIdx.x * 16 + 2 + c_item] = __builtin_amdgcn_s_memrealtime();
      if (++c_item == nitems) return true;
      cur = my[c_item];
      c_nt = (int)(((int64_t)cur.rb * TM + TM + KT - 1) / KT);
      c_t = 0;
    }
    return false;
  };
  for (;;) {
    if (step(L0, L2)) break;
    if (step(L1, L0)) break;
    if (step(L2, L1)) break;
  }
}

// ------------------------------------------------------------------------------------------
// Work-queue form of trmm_vsq_dma_kernel<true> (round 3).  The static LPT schedule balances the k-tile COUNT per worker
// to +-1, but the workers do not run equally fast -- the XCDs' mean finish times differ by 1.6 us and single workers by
// more (in-kernel stamps: finish 84.4 ... 91.3 us around a median of 87.5, profiles/r03_trmm_balance.txt) -- and the
// launch ends with its slowest worker.  Here every XCD's items sit in ONE list in LPT order (the same items, the same
// XCD placement) and its workers draw from it: the first item of a worker is its index in the list, every further one
// comes from a per-XCD counter.  The fetch costs the k-tile pipeline nothing: an atomic add and, one k-tile later, a load
// of the item, issued by lane 0 of wave 0 right after a k-tile barrier, i.e. BEFORE that k-tile's six LDS-direct loads --
// vmcnt counts in order, so the hand-placed s_waitcnt vmcnt(6) of the next barrier ("all but my six newest") covers them;
// items land in an 8-entry LDS ring, up to five ahead of the compute cursor, so neither the load cursor (two k-tiles
// ahead) nor the compute cursor ever waits for one.  Every item is computed exactly once by whoever
// draws it and writes its own output slot: the results do not depend on the assignment (chains stay bit-identical).
typedef int i4q __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 1) void trmm_vsq_dyn_kernel(   // one workgroup per CU (144 KiB of LDS): no need to squeeze into 128 VGPRs
    const double *__restrict__ Wt, const double *__restrict__ KS, double *__restrict__ out,
    const TrmmItem *__restrict__ qitems, const int *__restrict__ qn, unsigned int *__restrict__ qcnt, int qset, int qmax,
    int workers_per_xcd, int64_t Npad, int64_t Bcap, int k, int nrb, unsigned long long *__restrict__ stamps) {
  constexpr bool INTERLEAVE = true;
  constexpr int BUFD = KT * TM + KT * TILE;            // one k-tile: [W tile | K_*^T tile], 48 KiB
  __shared__ __attribute__((aligned(16))) double L0[BUFD];
  __shared__ __attribute__((aligned(16))) double L1[BUFD];
  __shared__ __attribute__((aligned(16))) double L2[BUFD];
  __shared__ double red[2][TILE];
  __shared__ TrmmItem s_items[8];           // ring: item j of this worker at j & 7; p < 0 = the list is exhausted

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lk = lane >> 4;

  if (stamps && tid == 0) stamps[blockIdx.x * 16] = __builtin_amdgcn_s_memrealtime();
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3;       // workgroups are dispatched round-robin over the XCDs
  const int nq = qn[xcd];
  const TrmmItem *qlist = qitems + (int64_t)xcd * qmax;
  unsigned int *const my_cnt = qcnt + ((int64_t)qset * 8 + xcd) * 32;
  if (blockIdx.x == 0 && tid < 8) qcnt[((int64_t)(qset ^ 1) * 8 + tid) * 32] = (unsigned)workers_per_xcd;   // the next launch's set
  // ring entry j carries its index in the upper bits of `half` (half | j << 1): an entry that has not arrived yet is
  // told from what the slot held eight items ago
  if (tid == 0) s_items[0] = (wx < nq) ? qlist[wx] : TrmmItem{-1, 0, 0, 0};
  if (tid >= 1 && tid < 8) s_items[tid] = TrmmItem{-1, 0, 0, -2};
  __syncthreads();
  if (s_items[0].p < 0) return;
  // item j of this worker: s_items[j & 7] (valid once the fetch pipeline below has written it and a barrier has passed)
  struct Ring { const TrmmItem *r; __device__ const TrmmItem &operator[](int j) const { return r[j & 7]; } } my{s_items};
  // fetch pipeline (wave 0): at most one atomic and one item load in flight, one stage per k-tile
  int f_wr = 1, fst = 0, q_tiles = 0;
  unsigned int at_val = 0;
  i4q ld_val = i4q{0, 0, 0, 0};

  // per-lane source offsets (doubles) of the wave's six 1 KiB chunks of a k-tile
  int offA[2], offB[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kk = 2 * (wave * 2 + j) + (lane >> 5);          // chunk = two k-rows of the W tile
    offA[j] = kk * (int)Npad + ((2 * (lane & 31)) ^ ((kk & 1) << 4));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = wave * 4 + r;                              // chunk = one k-row of the K_*^T tile
    offB[r] = kk * (int)Bcap + ((2 * lane) ^ ((kk & 1) << 4));
  }
  const int64_t astep = (int64_t)KT * Npad, bstep = (int64_t)KT * Bcap;

  // ---- load cursor (one k-tile ahead of the compute cursor) ----
  int l_item = 0, l_t = 0, l_nt = 1;
  bool l_end = false;
  const double *l_pa = Wt, *l_pb = KS;
  auto fetch_tick = [&]() __attribute__((always_inline)) {
    {
      // Fetch pipeline, one stage per k-tile.  Its two memory operations are issued HERE, i.e. before this k-tile's six
      // LDS-direct loads in wave 0's instruction stream: vmcnt counts in order, so the s_waitcnt vmcnt(6) of the NEXT
      // k-tile barrier -- "everything but my six newest loads" -- covers them without a cycle of extra waiting.  The
      // empty asm pins the uses of last k-tile's results behind the barrier.
      asm volatile("" : "+v"(at_val), "+v"(ld_val));
      // fst: bit 0 = an atomic is in flight, bit 1 = an item load is in flight, bit 2 = the list is exhausted (one
      // integer: as separate variables the compiler keeps them in scratch memory, whose reloads wait for vmcnt(0))
      int st = fst;
      if (st & 2) {
        if (lane == 0) s_items[f_wr & 7] = TrmmItem{ld_val[0], ld_val[1], ld_val[2], ld_val[3] | (f_wr << 1)};
        ++f_wr;
        q_tiles += 2 * (__builtin_amdgcn_readfirstlane(ld_val[1]) + 1);      // k-tiles of the item: 2 (rb + 1)
        st &= ~2;
      }
      if (st & 1) {
        st &= ~1;
        const int idx = __builtin_amdgcn_readfirstlane((int)at_val);
        if (idx < nq) {
          const TrmmItem *src = qlist + idx;
          if (lane == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld_val) : "v"(src) : "memory");
          st |= 2;
        } else {
          if (lane == 0) s_items[f_wr & 7] = TrmmItem{-1, 0, 0, f_wr << 1};
          ++f_wr;
          st |= 4;
        }
      }
      // Draw the next item only when it is about to be needed: the load cursor's remaining k-tiles plus those of the
      // items already waiting for it have shrunk to the fetch latency and a margin.  (Drawing as far ahead as the ring
      // allows hands out the whole list in the first microseconds, first come first served: 121 us instead of 95.)
      if (!(st & 7) && (l_nt - l_t) + q_tiles <= 7) {
        if (lane == 0) {
          const unsigned one = 1u;
          asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(at_val) : "v"(my_cnt), "v"(one) : "memory");
        }
        st |= 1;
      }
      fst = st;
    }
  };
  auto l_open = [&]() __attribute__((always_inline)) {
    const TrmmItem it = my[l_item];
    const int64_t i0 = (int64_t)it.rb * TM;
    l_nt = (int)((i0 + TM + KT - 1) / KT);
    l_pa = Wt + (int64_t)it.p * Npad * Npad + i0;
    l_pb = KS + (int64_t)it.p * Npad * Bcap + (it.col0 & ~(TILE - 1));
    l_t = 0;
  };
  // Issued as inline assembly: with the builtin, hipcc's waitcnt insertion treats every later LDS read as a
  // possible reader of the in-flight destination and drains vmcnt to 0, which defeats the two-tile lookahead;
  // the waits for these loads are the hand-written s_waitcnt vmcnt(6) of tile_barrier().
  auto dma1 = [&](const double *src, double *dst_wave_uniform) {
    const unsigned lds_off = (unsigned)(uintptr_t)((las_ptr)dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off"
                 :
                 : "s"(lds_off), "v"(src)
                 : "memory");
  };
  // one of the wave's six loads of a k-tile (part compile-time after unrolling); the cursor moves on after the last
  auto dma_part = [&](double *dA, int part) __attribute__((always_inline)) {
    double *dB = dA + KT * TM;
    if (part < 2) dma1(l_pa + offA[part], dA + (wave * 2 + part) * 128);
    else dma1(l_pb + offB[part - 2], dB + (wave * 4 + part - 2) * 128);
    if (part == 5 && !l_end) {
      if (++l_t == l_nt) {
        // Never expected (the fetch runs >= 4 k-tiles ahead of this point), but a late entry must not be mistaken for
        // the slot's old content: drain the fetch pipeline synchronously until the entry is there.
        while ((my[l_item + 1].half >> 1) != l_item + 1) {
          __syncthreads();
          if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fetch_tick();
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): wave 0's ring write
          __syncthreads();
        }
        if (my[l_item + 1].p >= 0) { ++l_item; l_open(); q_tiles -= l_nt; }
        else l_end = true;                 // stay on the last k-tile (harmless re-read)
      } else {
        l_pa += astep;
        l_pb += bstep;
      }
    }
  };
  auto dma = [&](double *dA) {
#pragma unroll
    for (int part = 0; part < 6; ++part) dma_part(dA, part);
  };

  // vmcnt(6) lgkmcnt(0): everything but this wave's six newest loads has landed, every LDS read has returned
  int c_item = 0;
  auto tile_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0x0076);
    __builtin_amdgcn_s_barrier();
    if (wave == 0) fetch_tick();
  };
  l_open();
  dma(L0);
  dma(L1);
  tile_barrier();
  if (stamps && tid == 0) stamps[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime();

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  int c_t = 0;
  TrmmItem cur = my[0];
  cur.half &= 1;
  int c_nt = (int)(((int64_t)cur.rb * TM + TM + KT - 1) / KT);
  const int sw = (lk & 1) << 4;                                 // this lane's k-rows are odd <=> lk odd
  const int ia0 = lk * TM + ((wm * 32 + lr) ^ sw), ia1 = lk * TM + ((wm * 32 + 16 + lr) ^ sw);

  auto step = [&](const double *cA, double *nA) __attribute__((always_inline)) -> bool {
    const double *cB = cA + KT * TM;
    if (!INTERLEAVE) dma(nA);                                   // k-tile +2 -> the buffer k-tile -1 was read from
    if (cur.half) {
      const int ib = lk * TILE + (((cur.col0 & 64) + wn * 16 + lr) ^ sw);
      double a[2][2], b[2];
      a[0][0] = cA[ia0];
      a[0][1] = cA[ia1];
      b[0] = cB[ib];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (ks + 1 < KT / 4) {
          a[nx][0] = cA[(ks + 1) * 4 * TM + ia0];
          a[nx][1] = cA[(ks + 1) * 4 * TM + ia1];
          b[nx] = cB[(ks + 1) * 4 * TILE + ib];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cu][mi], b[cu], acc[mi][0], 0, 0, 0);
        if (INTERLEAVE && ks < 6) dma_part(nA, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const int ib0 = lk * TILE + ((wn * 32 + lr) ^ sw), ib1 = lk * TILE + ((wn * 32 + 16 + lr) ^ sw);
      double a[2][2], b[2][2];
      a[0][0] = cA[ia0];
      a[0][1] = cA[ia1];
      b[0][0] = cB[ib0];
      b[0][1] = cB[ib1];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (ks + 1 < KT / 4) {
          a[nx][0] = cA[(ks + 1) * 4 * TM + ia0];
          a[nx][1] = cA[(ks + 1) * 4 * TM + ia1];
          b[nx][0] = cB[(ks + 1) * 4 * TILE + ib0];
          b[nx][1] = cB[(ks + 1) * 4 * TILE + ib1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cu][mi], b[cu][ni], acc[mi][ni], 0, 0, 0);
        if (INTERLEAVE && ks < 6) dma_part(nA, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    tile_barrier();
    if (++c_t == c_nt) {
      const int ncols = cur.half ? 64 : TILE;
      const int wcol = cur.half ? wn * 16 : wn * 32;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (ni == 0 || !cur.half) {
          double sq = 0.0;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq = fma(acc[mi][ni][r], acc[mi][ni][r], sq);
          sq += __shfl_xor(sq, 16);
          sq += __shfl_xor(sq, 32);
          if (lk == 0) red[wm][wcol + ni * 16 + lr] = sq;
        }
      }
      __syncthreads();
      if (tid < ncols)
        out[(((int64_t)cur.col0 + tid) * k + cur.p) * nrb + cur.rb] = red[0][tid] + red[1][tid];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};
      if (stamps && tid == 0 && c_item < 13) stamps[blockIdx.x * 16 + 2 + c_item] = __builtin_amdgcn_s_memrealtime();
      ++c_item;
      if (my[c_item].p < 0) {
        if (stamps && tid == 0) stamps[blockIdx.x * 16 + 14] = (unsigned long long)c_item;   // items this worker drew
        return true;
      }
      cur = my[c_item];
      cur.half &= 1;
      c_nt = (int)(((int64_t)cur.rb * TM + TM + KT - 1) / KT);
      c_t = 0;
    }
    return false;
  };
  for (;;) {
    if (step(L0, L2)) break;
    if (step(L1, L0)) break;
    if (step(L2, L1)) break;
  }
}

// ------------------------------------------------------------------------------------------
// Small-batch form (B <= 128 per launch: a rank's slice of the proposing half on a multi-GPU run).
// With few columns the 64 x 128 items are too few and too long (the longest, full-K item alone takes
// ~80 us), so here an item is 32 rows x 64 columns and its K range is split four ways INSIDE the
// workgroup: 8 waves = 4 K-slices x 2 column halves, each wave 32 x 32.  Slice s takes the k-tiles
// t = s, s+4, ...; after the K loop the four partial V tiles are summed through LDS, squared and
// column-summed.  Partials are per 32-row block (nrb = Npad / 32).
constexpr int SB_TM = 32, SB_TN = 64, SB_KT = 16, SB_SL = 4;
constexpr int SB_STRA = 48;   // (2 * 48) % 64 == 32
constexpr int SB_STRB = 80;

__global__ __launch_bounds__(512, 2) void trmm_vsq_smallb_kernel(const double *__restrict__ Wt,
                                                                 const double *__restrict__ KS,
                                                                 double *__restrict__ out, int64_t Npad,
                                                                 int64_t Bcap, int k, int nrb, int ncb) {
  // staging: [buf][slice][k][m or n]; reused as the reduction buffer [slice][32][64] at the end
  __shared__ __attribute__((aligned(16))) double sA[2][SB_SL][SB_KT][SB_STRA];
  __shared__ __attribute__((aligned(16))) double sB[2][SB_SL][SB_KT][SB_STRB];
  const int ncombo = k * ncb;
  const int rbi = blockIdx.x / ncombo;
  const int combo = blockIdx.x - rbi * ncombo;
  const int p = combo / ncb;
  const int cb = combo - p * ncb;
  const int rb = nrb - 1 - rbi;   // long K first
  const int64_t i0 = (int64_t)rb * SB_TM, b0 = (int64_t)cb * SB_TN;
  const int ntile = (int)((i0 + SB_TM + SB_KT - 1) / SB_KT);
  const int nround = (ntile + SB_SL - 1) / SB_SL;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, ksl = wave >> 1;
  const int lr = lane & 15, lk = lane >> 4;
  // staging per round: A 4 x (16 x 32) doubles = 1024 d2 (2 per thread), B 4 x (16 x 64) = 2048 d2 (4 per thread)
  //   A: idx = tid + 512 r: slice = idx >> 8, row = (idx >> 4) & 15, c2 = idx & 15
  //   B: idx = tid + 512 r: slice = idx >> 9, row = (idx >> 5) & 15, c2 = idx & 31
  const double *Ab = Wt + (int64_t)p * Npad * Npad + i0;
  const double *Bb = KS + (int64_t)p * Npad * Bcap + b0;
  d2 ra[2], rbv[4];
  auto gload = [&](int round) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int idx = tid + 512 * r, sl = idx >> 8, row = (idx >> 4) & 15, c2 = idx & 15;
      int t = round * SB_SL + sl;
      if (t >= ntile) t = ntile - 1;   // harmless re-read; the slice skips its MFMAs
      ra[r] = *reinterpret_cast<const d2 *>(Ab + (int64_t)(t * SB_KT + row) * Npad + 2 * c2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + 512 * r, sl = idx >> 9, row = (idx >> 5) & 15, c2 = idx & 31;
      int t = round * SB_SL + sl;
      if (t >= ntile) t = ntile - 1;
      rbv[r] = *reinterpret_cast<const d2 *>(Bb + (int64_t)(t * SB_KT + row) * Bcap + 2 * c2);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int idx = tid + 512 * r, sl = idx >> 8, row = (idx >> 4) & 15, c2 = idx & 15;
      *reinterpret_cast<d2 *>(&sA[buf][sl][row][2 * c2]) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + 512 * r, sl = idx >> 9, row = (idx >> 5) & 15, c2 = idx & 31;
      *reinterpret_cast<d2 *>(&sB[buf][sl][row][2 * c2]) = rbv[r];
    }
  };

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  gload(0);
  sstore(0);
  __syncthreads();
  for (int round = 0; round < nround; ++round) {
    const int buf = round & 1;
    if (round + 1 < nround) gload(round + 1);
    if (round * SB_SL + ksl < ntile) {
#pragma unroll
      for (int ks = 0; ks < SB_KT / 4; ++ks) {
        double a[2], b[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi] = sA[buf][ksl][ks * 4 + lk][mi * 16 + lr];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni] = sB[buf][ksl][ks * 4 + lk][wn * 32 + ni * 16 + lr];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    if (round + 1 < nround) sstore(buf ^ 1);
    __syncthreads();
  }
  // sum the four K-slices: V[row][col] through LDS (reuse sB: 4 x 32 x 64 doubles = 64 KiB <= sizeof(sB))
  double *red = &sB[0][0][0][0];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[(ksl * SB_TM + mi * 16 + lk + 4 * r) * SB_TN + wn * 32 + ni * 16 + lr] = acc[mi][ni][r];
  __syncthreads();
  // 512 threads: thread -> (column c = tid & 63, row group g = tid >> 6 of 4 rows)
  {
    const int c = tid & 63, g = tid >> 6;
    double sq = 0.0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = g * 4 + rr;
      const double v = (red[(0 * SB_TM + row) * SB_TN + c] + red[(1 * SB_TM + row) * SB_TN + c]) +
                       (red[(2 * SB_TM + row) * SB_TN + c] + red[(3 * SB_TM + row) * SB_TN + c]);
      sq = fma(v, v, sq);
    }
    double *colsum = &sA[0][0][0][0];   // [8][64]
    colsum[g * 64 + c] = sq;
    __syncthreads();
    if (tid < 64) {
      double tot = 0.0;
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) tot += colsum[gg * 64 + tid];
      out[((b0 + tid) * k + p) * nrb + rb] = tot;
    }
  }
}

// host side: LPT schedule of the items of one launch shape (cached per model and column-tile count)
static void build_trmm_schedule(gpemu_model *m, int ncb, std::vector<TrmmItem> &flat, std::vector<int> &cnt,
                                int &max_items, int &nworkers, std::vector<std::vector<TrmmItem>> *queues = nullptr) {
  const int nrb = (int)m->vsq_nrb, k = (int)m->k;
  struct It { double cost; TrmmItem it; };
  std::vector<It> items;
  // row blocks with short K are issued as two 64-column halves; with one or two column blocks (B <= 256) there
  // are too few items for 256 workers unless every row block is
  static const int split_all_ncb = getenv("GPEMU_TRMM_SPLIT_ALL_NCB") ? atoi(getenv("GPEMU_TRMM_SPLIT_ALL_NCB")) : 2;
  const int split_below = (ncb <= split_all_ncb) ? nrb : nrb / 4;
  // per-item cost of the epilogue, in k-tiles.  Measured (in-kernel stamps, profiles/r03_trmm_balance.txt): workers with
  // 2 / 3 / 5 items finish at 85.3 / 87.6 / 88.5 us, but raising it to 0.4-0.9 (3 / 4 items everywhere) leaves the slowest
  // worker at 91 us: the spread is the XCDs' (means 86.6-88.2 us), not the cost model's
  static const double ov = getenv("GPEMU_TRMM_OV") ? atof(getenv("GPEMU_TRMM_OV")) : 0.15;
  for (int rb = 0; rb < nrb; ++rb) {
    const double nt = (double)(((int64_t)rb * TM + TM + KT - 1) / KT);
    for (int p = 0; p < k; ++p)
      for (int cb = 0; cb < ncb; ++cb) {
        if (rb < split_below) {
          items.push_back({0.5 * nt + ov, TrmmItem{p, rb, cb * TILE, 1}});
          items.push_back({0.5 * nt + ov, TrmmItem{p, rb, cb * TILE + 64, 1}});
        } else {
          items.push_back({nt + ov, TrmmItem{p, rb, cb * TILE, 0}});
        }
      }
  }
  std::stable_sort(items.begin(), items.end(), [](const It &a, const It &b) { return a.cost > b.cost; });
  const int ncu = (m->worker_limit > 0 && m->worker_limit < m->num_cu) ? m->worker_limit : m->num_cu;
  nworkers = ncu < (int)items.size() ? ncu : (int)items.size();
  std::vector<std::vector<TrmmItem>> per(nworkers);
  std::vector<double> load(nworkers, 0.0);
  // XCD-aware placement.  Workgroups are dispatched round-robin over the 8 XCDs (worker w runs on XCD w % 8),
  // each with its own 4 MiB L2.  All items of one (PC, column block) group read the same K_*^T block and the
  // groups of one PC read the same W_p; every group costs the same, so when the groups divide evenly they are
  // dealt to the XCDs in (PC, column block) order -- an XCD then works on at most two or three PCs, and the
  // column-block items of one (PC, row block), equally long, start together and stream the same W rows through
  // that L2 -- and LPT runs within each XCD's workers.  Otherwise: plain LPT over all workers.
  const int nxcd = 8, ngroups = k * ncb;
  static const bool xcd_aware = getenv("GPEMU_TRMM_NO_XCD") == nullptr;
  if (xcd_aware && nworkers == ncu && nworkers % nxcd == 0 && ngroups % nxcd == 0) {
    const int gper = ngroups / nxcd;
    if (queues) {
      // the work-queue form: every XCD's items as one list in LPT order.  Usable when a worker's first item -- its
      // index in the list -- is long enough for the fetch pipeline to deliver the second one (a few k-tiles).
      queues->assign(nxcd, {});
      for (const It &x : items) (*queues)[(x.it.p * ncb + x.it.col0 / TILE) / gper].push_back(x.it);
      const int wpx = nworkers / nxcd;
      bool ok = true;
      for (const auto &q : *queues) {
        if ((int)q.size() < wpx) { ok = false; break; }
        for (int w = 0; w < wpx; ++w)
          if (((int64_t)q[w].rb * TM + TM + KT - 1) / KT < 8) ok = false;
      }
      if (!ok) queues->clear();
    }
    for (const It &x : items) {
      const int g = x.it.p * ncb + x.it.col0 / TILE;
      const int xcd = g / gper;
      int best = xcd;
      for (int w = xcd; w < nworkers; w += nxcd)
        if (load[w] < load[best]) best = w;
      per[best].push_back(x.it);
      load[best] += x.cost;
    }
  } else {
    for (const It &x : items) {  // longest processing time first onto the least loaded worker
      int best = 0;
      for (int w = 1; w < nworkers; ++w)
        if (load[w] < load[best]) best = w;
      per[best].push_back(x.it);
      load[best] += x.cost;
    }
  }
  max_items = 1;
  for (auto &v : per) max_items = v.size() > (size_t)max_items ? (int)v.size() : max_items;
  flat.assign((size_t)nworkers * max_items, TrmmItem{0, 0, 0, 0});
  cnt.assign(nworkers, 0);
  for (int w = 0; w < nworkers; ++w) {
    cnt[w] = (int)per[w].size();
    for (size_t i = 0; i < per[w].size(); ++i) flat[(size_t)w * max_items + i] = per[w][i];
  }
}

int launch_trmm_vsq(gpemu_model *m, int64_t B, hipStream_t st) {
  Workspace &w = m->ws;
  static const int smallb_max = getenv("GPEMU_SMALLB_MAX") ? atoi(getenv("GPEMU_SMALLB_MAX")) : 128;
  static const bool old_smallb = getenv("GPEMU_TRMM_OLD_SMALLB") != nullptr;
  const int64_t Bv = m->variant_B > 0 ? m->variant_B : B;   // a chain stacked with others is evaluated as it would be alone
  // More than 512 columns (emulation.predict on a large batch): one launch per 512 columns.  K_*^T of 1024 columns is
  // 82 MB, and with W_p it no longer streams through the XCDs' L2s the way the 512-column schedule is built for: one
  // launch of 1024 columns takes 228 us, two of 512 take 2 x 93 us.  Only when the pieces are equally wide (one
  // cached schedule); the column partials of a piece land where the single launch would put them.
  static const int64_t max_cols = getenv("GPEMU_TRMM_MAX_COLS") ? atoll(getenv("GPEMU_TRMM_MAX_COLS")) : 512;
  if (m->variant_B == 0 && max_cols >= 256 && B > max_cols) {
    const int64_t n = (B + max_cols - 1) / max_cols, per = round_up((B + n - 1) / n, TILE), last = B - (n - 1) * per;
    if (last > 0 && round_up(last, TILE) == per) {
      double *const KS0 = w.KS, *const V0 = w.vsq_part;
      int rc = GPEMU_OK;
      for (int64_t c0 = 0; c0 < B && rc == GPEMU_OK; c0 += per) {
        w.KS = KS0 + c0;
        w.vsq_part = V0 + c0 * m->k * m->vsq_nrb;
        rc = launch_trmm_vsq(m, std::min(per, B - c0), st);
      }
      w.KS = KS0;
      w.vsq_part = V0;
      return rc;
    }
  }
  if (Bv <= smallb_max && !old_smallb) {  // small batch: persistent 32 x 32 items, operands straight into registers
    const int rc = launch_trmm_vsq_small(m, B, st);
    if (rc != GPEMU_ERR_UNSUPPORTED) return rc;
    if (m->variant_B > 0) {
      set_error("stacked chains: the small-batch triangular GEMM cannot take %lld columns", (long long)B);
      return rc;
    }
  }
  if (Bv <= smallb_max && m->variant_B == 0) {   // general small-batch form: 32 x 64 items with the K range split inside the workgroup (at 129..256
                           // rows the persistent kernel with every item halved is faster: 69 us vs 81 us)
    const int nrb32 = (int)(m->Npad / SB_TM);
    const int ncb64 = (int)(round_up(B, SB_TN) / SB_TN);
    w.cur_nrb = nrb32;
    const int pe0s = prof_mark(m, st);
    hipLaunchKernelGGL(trmm_vsq_smallb_kernel, dim3((unsigned)(nrb32 * (int)m->k * ncb64)), dim3(512), 0, st, m->Wt,
                       w.KS, w.vsq_part, m->Npad, w.Bcap, (int)m->k, nrb32, ncb64);
    GP_HIP(hipGetLastError());
    prof_pair(m, 0, pe0s, prof_mark(m, st));
    return GPEMU_OK;
  }
  w.cur_nrb = (int)m->vsq_nrb;
  const int nrb = (int)m->vsq_nrb;
  const int ncb = (int)(round_up(B, TILE) / TILE);  // only the column tiles that hold real queries
  static const bool use_simple = getenv("GPEMU_TRMM_SIMPLE") != nullptr;
  if (use_simple) {
    const int pe0 = prof_mark(m, st);
    hipLaunchKernelGGL(trmm_vsq_kernel, dim3((unsigned)(nrb * (int)m->k * ncb)), dim3(512), 0, st, m->Wt,
                       w.KS, w.vsq_part, m->Npad, w.Bcap, (int)m->k, nrb, ncb);
    GP_HIP(hipGetLastError());
    prof_pair(m, 0, pe0, prof_mark(m, st));
    return GPEMU_OK;
  }
  const int cap = (m->worker_limit > 0 && m->worker_limit < m->num_cu) ? m->worker_limit : m->num_cu;
  if (m->sched_ncb != ncb || m->sched_cap != cap) {
    const gpemu_model::SchedEntry *hit = nullptr;
    for (const auto &e : m->sched_cache)
      if (e.ncb == ncb && e.cap == cap) hit = &e;
    if (!hit) {
      std::vector<TrmmItem> flat;
      std::vector<int> cnt;
      std::vector<std::vector<TrmmItem>> queues;
      int max_items = 0, nworkers = 0;
      build_trmm_schedule(m, ncb, flat, cnt, max_items, nworkers, &queues);
      if (max_items > TRMM_MAX_ITEMS) {
        set_error("triangular GEMM schedule needs %d items per worker (limit %d): batch too large for one launch",
                  max_items, TRMM_MAX_ITEMS);
        return GPEMU_ERR_UNSUPPORTED;
      }
      // schedules are kept (a few KB each): launches in flight keep reading the one they were given
      gpemu_model::SchedEntry e{ncb, cap, nullptr, nullptr, max_items, nworkers};
      GP_HIP(hipMalloc(&e.items, sizeof(TrmmItem) * flat.size()));
      GP_HIP(hipMalloc((void **)&e.cnt, sizeof(int) * cnt.size()));
      GP_HIP(hipMemcpy(e.items, flat.data(), sizeof(TrmmItem) * flat.size(), hipMemcpyHostToDevice));
      GP_HIP(hipMemcpy(e.cnt, cnt.data(), sizeof(int) * cnt.size(), hipMemcpyHostToDevice));
      if (!queues.empty()) {
        size_t qmax = 0;
        for (const auto &q : queues) qmax = std::max(qmax, q.size());
        std::vector<TrmmItem> qflat(8 * qmax, TrmmItem{-1, 0, 0, 0});
        std::vector<int> qn(8);
        for (int x = 0; x < 8; ++x) {
          qn[x] = (int)queues[x].size();
          std::copy(queues[x].begin(), queues[x].end(), qflat.begin() + x * qmax);
        }
        std::vector<unsigned int> qc(2 * 8 * 32, 0u);       // one counter per 128-byte line; both sets start at "workers per XCD"
        for (int t = 0; t < 16; ++t) qc[(size_t)t * 32] = (unsigned)(nworkers / 8);
        GP_HIP(hipMalloc(&e.qitems, sizeof(TrmmItem) * qflat.size()));
        GP_HIP(hipMalloc((void **)&e.qn, sizeof(int) * 8));
        GP_HIP(hipMalloc((void **)&e.qcnt, sizeof(unsigned int) * qc.size()));
        GP_HIP(hipMemcpy(e.qitems, qflat.data(), sizeof(TrmmItem) * qflat.size(), hipMemcpyHostToDevice));
        GP_HIP(hipMemcpy(e.qn, qn.data(), sizeof(int) * 8, hipMemcpyHostToDevice));
        GP_HIP(hipMemcpy(e.qcnt, qc.data(), sizeof(unsigned int) * qc.size(), hipMemcpyHostToDevice));
        e.qmax = (int)qmax;
      }
      m->sched_cache.push_back(e);
      hit = &m->sched_cache.back();
    }
    m->sched_cur = (int)(hit - m->sched_cache.data());
    m->sched_items = hit->items; m->sched_cnt = hit->cnt;
    m->sched_ncb = ncb; m->sched_cap = cap; m->sched_max_items = hit->max_items; m->sched_workers = hit->workers;
  }
  const int pe0 = prof_mark(m, st);
  static const bool use_dma = getenv("GPEMU_TRMM_NO_DMA") == nullptr;   // register-staged variant kept for comparison
  static const bool interleave = getenv("GPEMU_TRMM_DMA_TOP") == nullptr;
  static const char *stamp_path = getenv("GPEMU_TRMM_STAMP_FILE");       // diagnostic: per-worker time stamps
  static unsigned long long *dstamps = nullptr;
  static int stamp_calls = 0;
  if (stamp_path && !dstamps) {
    GP_HIP(hipMalloc((void **)&dstamps, sizeof(unsigned long long) * 16 * 1024));
    GP_HIP(hipMemset(dstamps, 0, sizeof(unsigned long long) * 16 * 1024));
  }
  // the work-queue form (GPEMU_TRMM_DYN=1; a measured negative, see trmm_vsq_dyn_kernel): default = the static LPT lists
  static const bool use_dyn = getenv("GPEMU_TRMM_DYN") && atoi(getenv("GPEMU_TRMM_DYN")) != 0;
  gpemu_model::SchedEntry *se = (m->sched_cur >= 0) ? &m->sched_cache[(size_t)m->sched_cur] : nullptr;
  if (use_dma && interleave && use_dyn && se && se->qitems) {
    // launches on one stream run in order: launch n draws from counter set n & 1 and re-arms the other one
    const int qset = (int)(se->launches++ & 1u);
    hipLaunchKernelGGL(trmm_vsq_dyn_kernel, dim3((unsigned)m->sched_workers), dim3(512), 0, st, m->Wt, w.KS, w.vsq_part,
                       (const TrmmItem *)se->qitems, se->qn, se->qcnt, qset, se->qmax, m->sched_workers / 8, m->Npad, w.Bcap,
                       (int)m->k, nrb, dstamps);
    static const bool dyn_debug = getenv("GPEMU_TRMM_DYN_DEBUG") != nullptr;
    if (dyn_debug && se->launches == 50) {
      GP_HIP(hipStreamSynchronize(st));
      unsigned int hc[2 * 8 * 32];
      int hn[8];
      GP_HIP(hipMemcpy(hc, se->qcnt, sizeof(hc), hipMemcpyDeviceToHost));
      GP_HIP(hipMemcpy(hn, se->qn, sizeof(hn), hipMemcpyDeviceToHost));
      fprintf(stderr, "dyn queue after launch 50 (set %d used): items per XCD", qset);
      for (int x = 0; x < 8; ++x) fprintf(stderr, " %d", hn[x]);
      fprintf(stderr, "; counters used set:");
      for (int x = 0; x < 8; ++x) fprintf(stderr, " %u", hc[(qset * 8 + x) * 32]);
      fprintf(stderr, "; other set:");
      for (int x = 0; x < 8; ++x) fprintf(stderr, " %u", hc[((qset ^ 1) * 8 + x) * 32]);
      fprintf(stderr, "\n");
    }
  } else if (use_dma && interleave)
    hipLaunchKernelGGL(trmm_vsq_dma_kernel<true>, dim3((unsigned)m->sched_workers), dim3(512), 0, st, m->Wt,
                       w.KS, w.vsq_part, (const TrmmItem *)m->sched_items, m->sched_cnt, m->sched_max_items,
                       m->Npad, w.Bcap, (int)m->k, nrb, dstamps);
  else if (use_dma)
    hipLaunchKernelGGL(trmm_vsq_dma_kernel<false>, dim3((unsigned)m->sched_workers), dim3(512), 0, st, m->Wt,
                       w.KS, w.vsq_part, (const TrmmItem *)m->sched_items, m->sched_cnt, m->sched_max_items,
                       m->Npad, w.Bcap, (int)m->k, nrb, dstamps);
  else
    hipLaunchKernelGGL(trmm_vsq_persistent_kernel, dim3((unsigned)m->sched_workers), dim3(512), 0, st, m->Wt,
                       w.KS, w.vsq_part, (const TrmmItem *)m->sched_items, m->sched_cnt, m->sched_max_items,
                       m->Npad, w.Bcap, (int)m->k, nrb);
  GP_HIP(hipGetLastError());
  prof_pair(m, 0, pe0, prof_mark(m, st));
  if (stamp_path && ++stamp_calls == 600) {
    GP_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(16 * 1024);
    GP_HIP(hipMemcpy(h.data(), dstamps, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    std::vector<int> cnt(m->sched_workers);
    GP_HIP(hipMemcpy(cnt.data(), m->sched_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
    if (FILE *f = fopen(stamp_path, "w")) {
      unsigned long long t0 = ~0ull;
      for (int wk = 0; wk < m->sched_workers; ++wk) t0 = std::min(t0, h[wk * 16]);
      for (int wk = 0; wk < m->sched_workers; ++wk) {
        if (h[wk * 16 + 14]) cnt[wk] = (int)h[wk * 16 + 14];          // the work-queue kernel: items the worker drew
        fprintf(f, "%d %d", wk, cnt[wk]);
        for (int i = 0; i < 2 + std::min(cnt[wk], 12); ++i) fprintf(f, " %.2f", (double)(h[wk * 16 + i] - t0) / 100.0);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#ifdef GPEMU_TRMM_STAMPS
  if (getenv("GPEMU_DUMP_STAMPS")) {
    static int dumped = 0;
    if (++dumped == 20) {
      GP_HIP(hipStreamSynchronize(st));
      std::vector<unsigned long long> h(512 * 8);
      GP_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_trmm_stamps), sizeof(unsigned long long) * h.size()));
      std::vector<int> cnt(m->sched_workers);
      GP_HIP(hipMemcpy(cnt.data(), m->sched_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
      FILE *f = fopen(getenv("GPEMU_DUMP_STAMPS"), "w");
      unsigned long long t0 = ~0ull;
      for (int w = 0; w < m->sched_workers; ++w) t0 = h[w * 8] < t0 ? h[w * 8] : t0;
      for (int w = 0; w < m->sched_workers; ++w) {
        fprintf(f, "%d %d", w, cnt[w]);
        for (int i = 0; i < 7; ++i) fprintf(f, " %.2f", i <= cnt[w] ? (double)(h[w * 8 + i] - t0) / 100.0 : -1.0);
        fprintf(f, " %llu", h[w * 8 + 7]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
  return GPEMU_OK;
}

// ------------------------------------------------------------------------------------------
__global__ void reduce_mean_var_kernel(const double *__restrict__ mean_part,
                                       const double *__restrict__ vsq_part,
                                       const double *__restrict__ kdiag, double *__restrict__ mean,
                                       double *__restrict__ var, int64_t B, int64_t Bcap, int k,
                                       int nchunk, int nrb) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * k) return;
  int p = (int)(idx / B);
  int64_t b = idx - (int64_t)p * B;
  double mu = 0.0, vs = 0.0;
  for (int c = 0; c < nchunk; ++c) mu += mean_part[(b * k + p) * nchunk + c];
  for (int r = 0; r < nrb; ++r) vs += vsq_part[(b * k + p) * nrb + r];
  double v = kdiag[p] - vs;
  if (v < 0.0) v = 0.0;     // skl _gpr.py:479-485
  double sd = sqrt(v);      // predict returns std (skl _gpr.py:494) ...
  mean[b * k + p] = mu;
  var[b * k + p] = sd * sd; // ... and the reference squares it again (ref: emulation.py:499)
}

int launch_reduce_mean_var(gpemu_model *m, int64_t B, double *dmean, double *dvar, hipStream_t st) {
  const Workspace &w = m->ws;
  int64_t n = B * m->k;
  hipLaunchKernelGGL(reduce_mean_var_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     w.mean_part, w.vsq_part, m->kdiag, dmean, dvar, B, w.Bcap, (int)m->k,
                     w.cur_nchunk, w.cur_nrb);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
