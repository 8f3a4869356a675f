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
    double *__restrict__ mean_part, int64_t N, int64_t Npad, int64_t Bcap, int has_const) {
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
                     w.KS, w.mean_part, m->N, m->Npad, w.Bcap, m->has_const)
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

int launch_trmm_vsq(gpemu_model *m, int64_t B, hipStream_t st) {
  const Workspace &w = m->ws;
  const int nrb = (int)(m->Npad / TILE);
  const int ncb = (int)(round_up(B, TILE) / TILE);  // only the column tiles that hold real queries
  const int nblk = nrb * (int)m->k * ncb;
  const int pe0 = prof_mark(m, st);
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
                     (int)(m->Npad / JCHUNK), (int)(m->Npad / TILE));
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
