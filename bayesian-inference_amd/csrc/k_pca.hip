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

// per column (thread = column, rows accumulated in order, as numpy reduces axis 0)
__global__ void column_stats_kernel(const double *__restrict__ Y, int N, int F, double *__restrict__ mean,
                                    double *__restrict__ var, double *__restrict__ scale) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double s = 0.0;
  for (int i = 0; i < N; ++i) s += Y[(int64_t)i * F + f];
  const double T = s / N;
  double corr = 0.0, sq = 0.0;
  for (int i = 0; i < N; ++i) {
    const double t = Y[(int64_t)i * F + f] - T;
    corr += t;
    sq += t * t;
  }
  const double unnorm = sq - corr * corr / N;
  const double v = unnorm / N;
  const double eps = 2.220446049250313e-16;
  const double bound = N * eps * v + (N * T * eps) * (N * T * eps);   // skl _data.py:76-89
  double sc = sqrt(v);
  if (v <= bound || sc == 0.0) sc = 1.0;
  mean[f] = T;
  var[f] = v;
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
                                  double *__restrict__ G, int64_t ldg, int transpose_work) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double s = 0.0;
  for (int i = 0; i < N; ++i) s += Ys[(int64_t)i * F + f];
  const double mu = s / N;
  pmean[f] = mu;
  for (int i = 0; i < N; ++i) {
    const double v = Ys[(int64_t)i * F + f] - mu;
    if (!transpose_work) G[(int64_t)f * ldg + i] = v;   // work column = feature f (length N)
    else G[(int64_t)i * ldg + f] = v;                   // work column = design point i (length F)
  }
}

__global__ void identity_kernel(double *V, int n, int64_t ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * ld) return;
  V[idx] = ((idx / ld) == (idx % ld)) ? 1.0 : 0.0;
}

// One round of the tournament: workgroup b rotates columns (p, q) of G (length m) and of V (length n).
__global__ __launch_bounds__(256) void jacobi_round_kernel(double *__restrict__ G, int64_t ldg, int m,
                                                           double *__restrict__ V, int64_t ldv, int n, int npad,
                                                           int round, unsigned long long *__restrict__ offmax) {
  __shared__ double red[3][4];
  __shared__ double cs[2];
  const int i = blockIdx.x;
  const int nm1 = npad - 1;
  int p, q;
  if (i == 0) { p = nm1; q = round % nm1; }
  else { p = (round + i) % nm1; q = (round - i + nm1) % nm1; }
  if (p >= n || q >= n) return;          // padding player sits out
  if (p > q) { const int t = p; p = q; q = t; }
  double *gp = G + (int64_t)p * ldg, *gq = G + (int64_t)q * ldg;
  double a = 0.0, b = 0.0, g = 0.0;
  for (int r = threadIdx.x; r < m; r += 256) {
    const double x = gp[r], y = gq[r];
    a = fma(x, x, a);
    b = fma(y, y, b);
    g = fma(x, y, g);
  }
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off); b += __shfl_xor(b, off); g += __shfl_xor(g, off);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; red[2][threadIdx.x >> 6] = g; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    g = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    double c = 1.0, s = 0.0;
    const double denom = sqrt(a * b);
    if (denom > 0.0 && g != 0.0) {
      const double off = fabs(g) / denom;
      atomicMax(offmax, (unsigned long long)__double_as_longlong(off));   // positive doubles order as integers
      if (off > 1e-300) {
        const double zeta = (b - a) / (2.0 * g);
        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        c = 1.0 / sqrt(1.0 + t * t);
        s = c * t;
      }
    }
    cs[0] = c; cs[1] = s;
  }
  __syncthreads();
  const double c = cs[0], s = cs[1];
  if (s == 0.0) return;
  for (int r = threadIdx.x; r < m; r += 256) {
    const double x = gp[r], y = gq[r];
    gp[r] = c * x - s * y;
    gq[r] = s * x + c * y;
  }
  double *vp = V + (int64_t)p * ldv, *vq = V + (int64_t)q * ldv;
  for (int r = threadIdx.x; r < n; r += 256) {
    const double x = vp[r], y = vq[r];
    vp[r] = c * x - s * y;
    vq[r] = s * x + c * y;
  }
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
  const int npad = (n + 1) & ~1;
  const int64_t ldg = round_up(m, 16), ldv = round_up(n, 16);
  double *dY = nullptr, *dYs = nullptr, *dmean = nullptr, *dvar = nullptr, *dscale = nullptr, *dpm = nullptr,
         *dG = nullptr, *dV = nullptr;
  unsigned long long *doff = nullptr;
  hipError_t e = hipMalloc((void **)&dY, sizeof(double) * N * F);
  auto A = [&](double **p, int64_t cnt) { if (e == hipSuccess) e = hipMalloc((void **)p, sizeof(double) * (size_t)cnt); };
  A(&dYs, N * F); A(&dmean, F); A(&dvar, F); A(&dscale, F); A(&dpm, F); A(&dG, (int64_t)n * ldg); A(&dV, (int64_t)n * ldv);
  if (e == hipSuccess) e = hipMalloc((void **)&doff, sizeof(unsigned long long));
  int rc = GPEMU_OK;
  std::vector<double> hG, hV;
  int sweeps = 0;
  if (e == hipSuccess) e = hipMemcpy(dY, Y, sizeof(double) * N * F, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(column_stats_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, nullptr, dY, (int)N, (int)F,
                       dmean, dvar, dscale);
    hipLaunchKernelGGL(standardise_kernel, dim3((unsigned)((N * F + 255) / 256)), dim3(256), 0, nullptr, dY, (int)N,
                       (int)F, dmean, dscale, dYs);
    e = hipMemsetAsync(dG, 0, sizeof(double) * (size_t)n * ldg, nullptr);
    hipLaunchKernelGGL(pca_centre_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, nullptr, dYs, (int)N, (int)F,
                       dpm, dG, ldg, tw ? 1 : 0);
    hipLaunchKernelGGL(identity_kernel, dim3((unsigned)(((int64_t)n * ldv + 255) / 256)), dim3(256), 0, nullptr, dV, n, ldv);
    const double tol = std::sqrt((double)m) * 2.220446049250313e-16;
    for (sweeps = 0; sweeps < 60 && e == hipSuccess; ++sweeps) {
      e = hipMemsetAsync(doff, 0, sizeof(unsigned long long), nullptr);
      for (int r = 0; r < npad - 1; ++r)
        hipLaunchKernelGGL(jacobi_round_kernel, dim3((unsigned)(npad / 2)), dim3(256), 0, nullptr, dG, ldg, m, dV, ldv,
                           n, npad, r, doff);
      unsigned long long bits = 0;
      if (e == hipSuccess) e = hipMemcpy(&bits, doff, sizeof(bits), hipMemcpyDeviceToHost);
      double off;
      std::memcpy(&off, &bits, sizeof(off));
      if (off <= tol) { ++sweeps; break; }
    }
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (e == hipSuccess) {
    hG.resize((size_t)n * ldg); hV.resize((size_t)n * ldv);
    e = hipMemcpy(hG.data(), dG, sizeof(double) * hG.size(), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hV.data(), dV, sizeof(double) * hV.size(), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(scaler_mean, dmean, sizeof(double) * F, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(scaler_var, dvar, sizeof(double) * F, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(scaler_scale, dscale, sizeof(double) * F, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(pca_mean, dpm, sizeof(double) * F, hipMemcpyDeviceToHost);
  }
  (void)hipFree(dY); (void)hipFree(dYs); (void)hipFree(dmean); (void)hipFree(dvar); (void)hipFree(dscale);
  (void)hipFree(dpm); (void)hipFree(dG); (void)hipFree(dV); (void)hipFree(doff);
  if (e != hipSuccess) { set_error("pca_fit: %s", hipGetErrorString(e)); return GPEMU_ERR_HIP; }
  if (rc != GPEMU_OK) return rc;
  if (n_sweeps) *n_sweeps = sweeps;

  // ---- host assembly: order by singular value, normalise, sign convention -----------------------------
  std::vector<double> sigma((size_t)n);
  for (int j = 0; j < n; ++j) {
    double s2 = 0.0;
    for (int r = 0; r < m; ++r) s2 += hG[(size_t)j * ldg + r] * hG[(size_t)j * ldg + r];
    sigma[j] = std::sqrt(s2);
  }
  std::vector<int> order((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sigma[a] > sigma[b]; });
  double total_var = 0.0;
  for (int j = 0; j < n; ++j) total_var += sigma[j] * sigma[j] / (double)(N - 1);
  std::vector<double> comp((size_t)F);
  for (int64_t c = 0; c < nc; ++c) {
    const int j = order[(size_t)c];
    const double sg = sigma[j];
    // component row (length F): right singular vector of Xc
    if (!tw) for (int64_t f = 0; f < F; ++f) comp[f] = hV[(size_t)j * ldv + f];
    else for (int64_t f = 0; f < F; ++f) comp[f] = sg > 0.0 ? hG[(size_t)j * ldg + f] / sg : 0.0;
    int64_t arg = 0;
    double best = -1.0;
    for (int64_t f = 0; f < F; ++f)
      if (std::fabs(comp[f]) > best) { best = std::fabs(comp[f]); arg = f; }
    const double sign = comp[arg] < 0.0 ? -1.0 : 1.0;     // svd_flip, u_based_decision=False
    for (int64_t f = 0; f < F; ++f) components[c * F + f] = sign * comp[f];
    if (flip_argmax) flip_argmax[c] = arg;
    explained_variance[c] = sg * sg / (double)(N - 1);
    explained_variance_ratio[c] = explained_variance[c] / total_var;
    // Y_pca[:, c] = U[:, c] * S[c]
    if (!tw) for (int64_t i = 0; i < N; ++i) Y_pca[i * nc + c] = sign * hG[(size_t)j * ldg + i];
    else for (int64_t i = 0; i < N; ++i) Y_pca[i * nc + c] = sign * sg * hV[(size_t)j * ldv + i];
  }
  return GPEMU_OK;
}
