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
//   trmm_vsq_kernel   sum_i (W_p K_*^T)[i][b]^2 per 128-row block          (MFMA f64)
//   reduce_kernel     sums the partials, var = kdiag - vsq, clip, std^2
#include "internal.h"

namespace gpemu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
__global__ void pad_queries_kernel(const double *__restrict__ X, double *__restrict__ Xq, int64_t B,
                                   int64_t Bcap, int d) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Bcap * DPAD) return;
  int64_t b = idx / DPAD;
  int dd = (int)(idx % DPAD);
  Xq[idx] = (b < B && dd < d) ? X[b * d + dd] : 0.0;
}

int launch_pad_queries(gpemu_model *m, int64_t B, const double *dX, hipStream_t st) {
  int64_t n = m->ws.Bcap * DPAD;
  hipLaunchKernelGGL(pad_queries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dX,
                     m->ws.Xq, B, m->ws.Bcap, (int)m->d);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

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
template <int KIND>
__global__ __launch_bounds__(256) void kstar_kernel(
    const double *__restrict__ Xq, const double *__restrict__ Xs, const double *__restrict__ ls,
    const double *__restrict__ constv, const double *__restrict__ alpha, double *__restrict__ KS,
    double *__restrict__ mean_part, int64_t N, int64_t Npad, int64_t Bcap, int has_const,
    int *__restrict__ work_counter) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *work_counter = 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = blockIdx.z;
  const int chunk = blockIdx.y;
  const int nchunk = gridDim.y;
  const int64_t b = (int64_t)blockIdx.x * 64 + lane;

  double xq[DPAD];
#pragma unroll
  for (int dd = 0; dd < DPAD; ++dd) xq[dd] = Xq[b * DPAD + dd] / ls[p * DPAD + dd];
  const double c = has_const ? constv[p] : 0.0;

  const int64_t jbase = (int64_t)chunk * JCHUNK + wave * (JCHUNK / 4);
  const double *xs = Xs + ((int64_t)p * Npad + jbase) * DPAD;
  const double *al = alpha + (int64_t)p * Npad + jbase;
  double *ks = KS + ((int64_t)p * Npad + jbase) * Bcap + b;
  double macc = 0.0;
#pragma unroll 4
  for (int jj = 0; jj < JCHUNK / 4; ++jj) {
    double r2 = 0.0;
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) {
      double df = xq[dd] - xs[jj * DPAD + dd];  // wave-uniform address -> scalar loads
      r2 = fma(df, df, r2);
    }
    double v = base_kernel<KIND>(r2) + c;
    if (jbase + jj >= N) v = 0.0;  // padded training rows contribute nothing
    ks[(int64_t)jj * Bcap] = v;
    macc = fma(al[jj], v, macc);
  }
  __shared__ double red[4][64];
  red[wave][lane] = macc;
  __syncthreads();
  if (wave == 0) {
    double s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    mean_part[((int64_t)p * nchunk + chunk) * Bcap + b] = s;
  }
}

int launch_kstar(gpemu_model *m, int64_t B, const double *dXq, hipStream_t st) {
  // only the column tiles that hold real queries; the rows of dXq up to round_up(B, TILE) must be
  // finite (pad_queries_kernel / the sampler's proposal buffer zero them)
  const Workspace &w = m->ws;
  dim3 grid((unsigned)(round_up(B, TILE) / 64), (unsigned)(m->Npad / JCHUNK), (unsigned)m->k), block(256);
  const int pe0 = prof_mark(m, st);
  int kind = 0;
  if (m->kernel_kind == GPEMU_KERNEL_MATERN) kind = (m->nu == 0.5) ? 1 : (m->nu == 1.5 ? 2 : 3);
#define GP_LAUNCH_KSTAR(KD)                                                                      \
  hipLaunchKernelGGL(kstar_kernel<KD>, grid, block, 0, st, dXq, m->Xs, m->ls, m->constv, m->alpha, \
                     w.KS, w.mean_part, m->N, m->Npad, w.Bcap, m->has_const, m->work_counter)
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
//   out[p][rb][b] = sum_{i in row block rb} V[i][b]^2
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128, K step 16, LDS double buffered.
// MFMA v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][n = l&15];
// D[reg] is row (l>>4) + 4*reg, column l&15.
constexpr int KT = 16;
constexpr int LSTR = 144;  // LDS row stride in doubles: (2*LSTR) % 64 == 32 -> conflict-free b64 reads

__global__ __launch_bounds__(256) void trmm_vsq_kernel(const double *__restrict__ Wt,
                                                       const double *__restrict__ KS,
                                                       double *__restrict__ out, int64_t Npad,
                                                       int64_t Bcap, int k, int nrb, int ncb) {
  __shared__ __attribute__((aligned(16))) double sA[2][KT][LSTR];
  __shared__ __attribute__((aligned(16))) double sB[2][KT][LSTR];
  __shared__ double red[2][TILE];

  // heavy row blocks first (work per block ~ rb + 1)
  const int bid = blockIdx.x;
  const int rbi = bid / (k * ncb);
  const int rem = bid - rbi * (k * ncb);
  const int p = rem / ncb;
  const int cb = rem - p * ncb;
  const int rb = nrb - 1 - rbi;
  const int64_t i0 = (int64_t)rb * TILE, b0 = (int64_t)cb * TILE;
  const double *A = Wt + (int64_t)p * Npad * Npad + i0;
  const double *Bm = KS + (int64_t)p * Npad * Bcap + b0;
  const int ntile = (int)((i0 + TILE) / KT);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;

  // global -> register staging: 4 x 16 B per operand per thread; one wave-load = one 1 KiB row
  d2 ra[4], rbv[4];
  auto gload = [&](int t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int idx = tid + 256 * r;
      int row = idx >> 6, c2 = idx & 63;
      ra[r] = *reinterpret_cast<const d2 *>(A + (int64_t)(t * KT + row) * Npad + 2 * c2);
      rbv[r] = *reinterpret_cast<const d2 *>(Bm + (int64_t)(t * KT + row) * Bcap + 2 * c2);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int idx = tid + 256 * r;
      int row = idx >> 6, c2 = idx & 63;
      *reinterpret_cast<d2 *>(&sA[buf][row][2 * c2]) = ra[r];
      *reinterpret_cast<d2 *>(&sB[buf][row][2 * c2]) = rbv[r];
    }
  };

  d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  gload(0);
  sstore(0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntile) gload(t + 1);
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {
      double a[4], b[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[mi] = sA[buf][ks * 4 + lk][wm * 64 + mi * 16 + lr];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) b[ni] = sB[buf][ks * 4 + lk][wn * 64 + ni * 16 + lr];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (t + 1 < ntile) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: column sums of V^2 over this wave's 64 rows
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    double s = 0.0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) s = fma(acc[mi][ni][r], acc[mi][ni][r], s);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lk == 0) red[wm][wn * 64 + ni * 16 + lr] = s;
  }
  __syncthreads();
  if (tid < TILE) out[((int64_t)p * nrb + rb) * Bcap + b0 + tid] = red[0][tid] + red[1][tid];
}


// ------------------------------------------------------------------------------------------
// v2: 64-row x 128-column tile, 4 waves (2 x 2, 32 x 64 per wave = 2 x 4 MFMA tiles, 64 accumulator
// registers), two workgroups per CU (2 waves/SIMD).  Smaller work items (<= 4 "units" instead of 8)
// balance the triangular work over 256 CUs, and the accumulators stay in VGPRs for the whole loop.
constexpr int TM2 = 64;
constexpr int LSTRA2 = 80;   // (2*80) % 64 == 32

__global__ __launch_bounds__(256, 2) void trmm_vsq_kernel_v2(const double *__restrict__ Wt,
                                                             const double *__restrict__ KS,
                                                             double *__restrict__ out, int64_t Npad,
                                                             int64_t Bcap, int k, int nrb2, int ncb,
                                                             int nrb_out, int xcd_map) {
  __shared__ __attribute__((aligned(16))) double sA[2][KT][LSTRA2];
  __shared__ __attribute__((aligned(16))) double sB[2][KT][LSTR];
  __shared__ double red[2][TILE];

  // Work-item order: heavy (long K) row blocks first.  XCD affinity: workgroups are dealt round-robin
  // over the 8 XCDs (bid % 8), so give every XCD whole (PC, column-block) combos: the 16 row blocks
  // of a combo stream the same K_* tiles in step and share them through that XCD's L2 (speed only).
  const int ncombo = k * ncb;
  int rbi, combo;
  if (xcd_map) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, cpx = ncombo >> 3;
    rbi = slot / cpx;
    combo = (slot - rbi * cpx) * 8 + xcd;
  } else {
    rbi = blockIdx.x / ncombo;
    combo = blockIdx.x - rbi * ncombo;
  }
  const int p = combo / ncb;
  const int cb = combo - p * ncb;
  const int rb = nrb2 - 1 - rbi;
  const int64_t i0 = (int64_t)rb * TM2, b0 = (int64_t)cb * TILE;
  const double *A = Wt + (int64_t)p * Npad * Npad + i0;
  const double *Bm = KS + (int64_t)p * Npad * Bcap + b0;
  const int ntile = (int)((i0 + TM2) / KT);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;

  // staging: A tile 16 x 64 doubles = 512 x 16 B (2 per thread), B tile 16 x 128 = 1024 x 16 B (4)
  d2 ra[2], rbv[4];
  auto gload = [&](int t) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int idx = tid + 256 * r;
      int row = idx >> 5, c2 = idx & 31;
      ra[r] = *reinterpret_cast<const d2 *>(A + (int64_t)(t * KT + row) * Npad + 2 * c2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int idx = tid + 256 * r;
      int row = idx >> 6, c2 = idx & 63;
      rbv[r] = *reinterpret_cast<const d2 *>(Bm + (int64_t)(t * KT + row) * Bcap + 2 * c2);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int idx = tid + 256 * r;
      int row = idx >> 5, c2 = idx & 31;
      *reinterpret_cast<d2 *>(&sA[buf][row][2 * c2]) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int idx = tid + 256 * r;
      int row = idx >> 6, c2 = idx & 63;
      *reinterpret_cast<d2 *>(&sB[buf][row][2 * c2]) = rbv[r];
    }
  };

  d4 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  gload(0);
  sstore(0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntile) gload(t + 1);
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {
      double a[2], b[4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = sA[buf][ks * 4 + lk][wm * 32 + mi * 16 + lr];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) b[ni] = sB[buf][ks * 4 + lk][wn * 64 + ni * 16 + lr];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (t + 1 < ntile) sstore(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    double s = 0.0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) s = fma(acc[mi][ni][r], acc[mi][ni][r], s);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lk == 0) red[wm][wn * 64 + ni * 16 + lr] = s;
  }
  __syncthreads();
  if (tid < TILE) out[((int64_t)p * nrb_out + rb) * Bcap + b0 + tid] = red[0][tid] + red[1][tid];
}

// ------------------------------------------------------------------------------------------
// v3 = v2's tile (64 x 128, 4 waves, 2 workgroups per CU) made persistent: the grid is 2 x #CU
// workgroups that pull (row block, PC, column block) items from a device counter, heaviest first, so
// the triangular work is balanced whatever the dispatcher's placement (LPT scheduling).  The counter
// is zeroed by kstar_kernel, which always runs before this kernel on the same stream.  Operand
// fragments of k-step ks+1 are read from LDS before the MFMAs of k-step ks are issued.
__global__ __launch_bounds__(256, 2) void trmm_vsq_kernel_v3(const double *__restrict__ Wt,
                                                             const double *__restrict__ KS,
                                                             double *__restrict__ out,
                                                             int *__restrict__ counter, int64_t Npad,
                                                             int64_t Bcap, int k, int nrb2, int ncb,
                                                             int nitems) {
  __shared__ __attribute__((aligned(16))) double sA[2][KT][LSTRA2];
  __shared__ __attribute__((aligned(16))) double sB[2][KT][LSTR];
  __shared__ double red[2][TILE];
  __shared__ int s_item;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int ncombo = k * ncb;
  // staging coordinates (A: 2 x 16 B per thread, B: 4 x 16 B per thread)
  const int arow = tid >> 5, ac2 = tid & 31;   // + 8 rows for the second load
  const int brow = tid >> 6, bc2 = tid & 63;   // + 4 rows per further load

  for (;;) {
    if (tid == 0) s_item = atomicAdd(counter, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= nitems) break;   // uniform
    const int rbi = item / ncombo;
    const int combo = item - rbi * ncombo;
    const int p = combo / ncb;
    const int cb = combo - p * ncb;
    const int rb = nrb2 - 1 - rbi;
    const int64_t i0 = (int64_t)rb * TM2, b0 = (int64_t)cb * TILE;
    const int ntile = (int)((i0 + TM2) / KT);
    const double *pa = Wt + (int64_t)p * Npad * Npad + i0 + (int64_t)arow * Npad + 2 * ac2;
    const double *pb = KS + (int64_t)p * Npad * Bcap + b0 + (int64_t)brow * Bcap + 2 * bc2;
    const int64_t astep = (int64_t)KT * Npad, bstep = (int64_t)KT * Bcap;

    d2 ra[2], rbv[4];
    auto gload = [&]() {
      ra[0] = *reinterpret_cast<const d2 *>(pa);
      ra[1] = *reinterpret_cast<const d2 *>(pa + 8 * Npad);
#pragma unroll
      for (int r = 0; r < 4; ++r) rbv[r] = *reinterpret_cast<const d2 *>(pb + (int64_t)(4 * r) * Bcap);
      pa += astep;
      pb += bstep;
    };
    auto sstore = [&](int buf) {
      *reinterpret_cast<d2 *>(&sA[buf][arow][2 * ac2]) = ra[0];
      *reinterpret_cast<d2 *>(&sA[buf][arow + 8][2 * ac2]) = ra[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<d2 *>(&sB[buf][brow + 4 * r][2 * bc2]) = rbv[r];
    };

    d4 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

    gload();
    sstore(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
      const int buf = t & 1;
      if (t + 1 < ntile) gload();
      double a[2][2], b[2][4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[0][mi] = sA[buf][lk][wm * 32 + mi * 16 + lr];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) b[0][ni] = sB[buf][lk][wn * 64 + ni * 16 + lr];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < KT / 4) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) a[nxt][mi] = sA[buf][(ks + 1) * 4 + lk][wm * 32 + mi * 16 + lr];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) b[nxt][ni] = sB[buf][(ks + 1) * 4 + lk][wn * 64 + ni * 16 + lr];
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][mi], b[cur][ni], acc[mi][ni], 0, 0, 0);
      }
      if (t + 1 < ntile) sstore(buf ^ 1);
      __syncthreads();
    }

#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      double s = 0.0;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) s = fma(acc[mi][ni][r], acc[mi][ni][r], s);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lk == 0) red[wm][wn * 64 + ni * 16 + lr] = s;
    }
    __syncthreads();
    if (tid < TILE) out[((int64_t)p * nrb2 + rb) * Bcap + b0 + tid] = red[0][tid] + red[1][tid];
    // the next item's first __syncthreads (after the counter read) orders the reuse of red / s_item
  }
}

// ------------------------------------------------------------------------------------------
// v4: one 512-thread workgroup per CU (8 waves = 2 per SIMD, 2 (M) x 4 (N), 32 x 32 per wave) on a
// 64 x 128 tile with K step 32.  All 8 waves work on ONE item, so an item runs at the CU's full MFMA
// rate (with two independent workgroups per CU the longest item ran at half rate and set the
// makespan).  112 KiB of LDS pins residency at one workgroup per CU; the hardware dispatcher hands
// the heavy-first ordered items to CUs as they free up.
constexpr int KT4 = 32;

template <int DBG>
__global__ __launch_bounds__(512, 2) void trmm_vsq_kernel_v4(const double *__restrict__ Wt,
                                                             const double *__restrict__ KS,
                                                             double *__restrict__ out, int64_t Npad,
                                                             int64_t Bcap, int k, int nrb2, int ncb) {
  __shared__ __attribute__((aligned(16))) double sA[2][KT4][LSTRA2];
  __shared__ __attribute__((aligned(16))) double sB[2][KT4][LSTR];
  __shared__ double red[2][TILE];

  const int ncombo = k * ncb;
  const int rbi = blockIdx.x / ncombo;
  const int combo = blockIdx.x - rbi * ncombo;
  const int p = combo / ncb;
  const int cb = combo - p * ncb;
  const int rb = nrb2 - 1 - rbi;  // heavy (long K) row blocks first
  const int64_t i0 = (int64_t)rb * TM2, b0 = (int64_t)cb * TILE;
  const int ntile = (int)((i0 + TM2 + KT4 - 1) / KT4);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lk = lane >> 4;
  // staging: A tile 32 x 64 doubles = 1024 x 16 B (2 per thread), B tile 32 x 128 = 2048 x 16 B (4)
  const int arow = tid >> 5, ac2 = tid & 31;   // rows arow, arow + 16
  const int brow = tid >> 6, bc2 = tid & 63;   // rows brow + 8 r
  const double *pa = Wt + (int64_t)p * Npad * Npad + i0 + (int64_t)arow * Npad + 2 * ac2;
  const double *pb = KS + (int64_t)p * Npad * Bcap + b0 + (int64_t)brow * Bcap + 2 * bc2;
  const int64_t astep = (int64_t)KT4 * Npad, bstep = (int64_t)KT4 * Bcap;

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
    if (DBG != 1 && DBG != 3 && t + 1 < ntile) gload();
    double a[2][2], b[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[0][mi] = sA[buf][lk][wm * 32 + mi * 16 + lr];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) b[0][ni] = sB[buf][lk][wn * 32 + ni * 16 + lr];
#pragma unroll
    for (int ks = 0; ks < KT4 / 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (DBG == 3) {
        a[nxt][0] = a[cur][0]; a[nxt][1] = a[cur][1]; b[nxt][0] = b[cur][0]; b[nxt][1] = b[cur][1];
      } else if (ks + 1 < KT4 / 4) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[nxt][mi] = sA[buf][(ks + 1) * 4 + lk][wm * 32 + mi * 16 + lr];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[nxt][ni] = sB[buf][(ks + 1) * 4 + lk][wn * 32 + ni * 16 + lr];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          if (DBG != 2) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][mi], b[cur][ni], acc[mi][ni], 0, 0, 0);
          else acc[mi][ni][0] += a[cur][mi] + b[cur][ni];
    }
    if (DBG != 1 && DBG != 3 && t + 1 < ntile) sstore(buf ^ 1);
    __syncthreads();
  }

  // column sums of V^2: this wave's 32 rows x 32 columns
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
  if (tid < TILE) out[((int64_t)p * nrb2 + rb) * Bcap + b0 + tid] = red[0][tid] + red[1][tid];
}

int launch_trmm_vsq(gpemu_model *m, int64_t B, hipStream_t st) {
  const Workspace &w = m->ws;
  const int nrb = (int)(m->Npad / TILE);
  const int ncb = (int)(round_up(B, TILE) / TILE);  // only the column tiles that hold real queries
  const int nblk = nrb * (int)m->k * ncb;
  const int pe0 = prof_mark(m, st);
  if (m->trmm_variant == 4) {
    const int nrb2 = (int)(m->Npad / TM2);
    static const int dbg = getenv("GPEMU_TRMM_DBG") ? atoi(getenv("GPEMU_TRMM_DBG")) : 0;
    const dim3 g4((unsigned)(nrb2 * (int)m->k * ncb));
    if (dbg == 1)
      hipLaunchKernelGGL(trmm_vsq_kernel_v4<1>, g4, dim3(512), 0, st, m->Wt, w.KS, w.vsq_part, m->Npad, w.Bcap, (int)m->k, nrb2, ncb);
    else if (dbg == 3)
      hipLaunchKernelGGL(trmm_vsq_kernel_v4<3>, g4, dim3(512), 0, st, m->Wt, w.KS, w.vsq_part, m->Npad, w.Bcap, (int)m->k, nrb2, ncb);
    else if (dbg == 2)
      hipLaunchKernelGGL(trmm_vsq_kernel_v4<2>, g4, dim3(512), 0, st, m->Wt, w.KS, w.vsq_part, m->Npad, w.Bcap, (int)m->k, nrb2, ncb);
    else
      hipLaunchKernelGGL(trmm_vsq_kernel_v4<0>, g4, dim3(512), 0, st, m->Wt, w.KS, w.vsq_part, m->Npad, w.Bcap, (int)m->k, nrb2, ncb);
  } else if (m->trmm_variant == 3) {
    const int nrb2 = (int)(m->Npad / TM2);
    const int nitems = nrb2 * (int)m->k * ncb;
    int grid = 2 * m->num_cu;
    if (grid > nitems) grid = nitems;
    hipLaunchKernelGGL(trmm_vsq_kernel_v3, dim3((unsigned)grid), dim3(256), 0, st, m->Wt, w.KS,
                       w.vsq_part, m->work_counter, m->Npad, w.Bcap, (int)m->k, nrb2, ncb, nitems);
  } else if (m->trmm_variant == 2) {
    const int nrb2 = (int)(m->Npad / TM2);
    hipLaunchKernelGGL(trmm_vsq_kernel_v2, dim3((unsigned)(nrb2 * (int)m->k * ncb)), dim3(256), 0, st,
                       m->Wt, w.KS, w.vsq_part, m->Npad, w.Bcap, (int)m->k, nrb2, ncb, nrb2,
                       (((int)m->k * ncb) % 8 == 0 && !getenv("GPEMU_NO_XCD_MAP")) ? 1 : 0);
  } else
  hipLaunchKernelGGL(trmm_vsq_kernel, dim3((unsigned)nblk), dim3(256), 0, st, m->Wt, w.KS,
                     w.vsq_part, m->Npad, w.Bcap, (int)m->k, nrb, ncb);
  GP_HIP(hipGetLastError());
  prof_pair(m, 0, pe0, prof_mark(m, st));
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
  for (int c = 0; c < nchunk; ++c) mu += mean_part[((int64_t)p * nchunk + c) * Bcap + b];
  for (int r = 0; r < nrb; ++r) vs += vsq_part[((int64_t)p * nrb + r) * Bcap + b];
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
                     (int)(m->Npad / JCHUNK), (int)m->vsq_nrb);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
