// Probe: accuracy of v_rsq_f64 and of 1 / 2 Newton steps on it (the reciprocal square root on the serial path of the
// 64 x 64 Cholesky, k_fit.hip), against 1 / sqrt in long double on the host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double *x, double *o0, double *o1, double *o2, double *o3, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p = x[i];
  double r = __builtin_amdgcn_rsq(p);
  o0[i] = r;
  r = fma(0.5 * r, fma(-p * r, r, 1.0), r);
  o1[i] = r;
  r = fma(0.5 * r, fma(-p * r, r, 1.0), r);
  o2[i] = r;
  // one third-order step: e = 1 - p r^2, r' = r + r e (1/2 + 3/8 e): 4 dependent operations instead of 6
  const double r0 = __builtin_amdgcn_rsq(p);
  const double e = fma(-p * r0, r0, 1.0);
  o3[i] = fma(r0 * e, fma(0.375, e, 0.5), r0);
}

int main() {
  const int n = 1 << 22;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;          // [0, 1)
    const int e = (int)((s >> 3) % 80) - 40;
    h[i] = std::ldexp(1.0 + u, e);
  }
  double *dx, *d0, *d1, *d2, *d3;
  hipMalloc(&dx, 8 * n); hipMalloc(&d0, 8 * n); hipMalloc(&d1, 8 * n); hipMalloc(&d2, 8 * n); hipMalloc(&d3, 8 * n);
  hipMemcpy(dx, h.data(), 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
  std::vector<double> r0(n), r1(n), r2(n), r3(n);
  hipMemcpy(r0.data(), d0, 8 * n, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, 8 * n, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, 8 * n, hipMemcpyDeviceToHost);
  hipMemcpy(r3.data(), d3, 8 * n, hipMemcpyDeviceToHost);
  long double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / sqrtl((long double)h[i]);
    e0 = fmaxl(e0, fabsl((r0[i] - t) / t));
    e1 = fmaxl(e1, fabsl((r1[i] - t) / t));
    e2 = fmaxl(e2, fabsl((r2[i] - t) / t));
    e3 = fmaxl(e3, fabsl((r3[i] - t) / t));
  }
  printf("max relative error over %d arguments: v_rsq_f64 %.3Le   + 1 Newton step %.3Le   + 2 steps %.3Le   one third-order step %.3Le   (2^-53 = 1.11e-16)\n", n, e0, e1, e2, e3);
  return 0;
}
