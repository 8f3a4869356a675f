// Device helpers of the cross-kernel (K_*) evaluation shared by kstar_kernel (k_predict.hip) and the fused
// sampler front kernel (k_front.hip): the table-based exponential and the base kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace gpemu {

// exp(x) for x <= 0 (the only arguments the kernels produce): x = (32 e + j) ln2/32 + r, |r| <= ln2/64,
// exp(x) = 2^e 2^(j/32) (1 + p(r)) with a degree-6 Taylor p and a 32-entry table in LDS.  < 1 ulp (checked
// against 200-bit arithmetic); 19 instructions instead of ~30 for the library exp, no special-case branches:
// arguments below -1000 are clamped (result 0), NaN propagates.
__device__ static const double c_exp2_32[32] = {
    1.0, 1.0218971486541166, 1.0442737824274138, 1.0671404006768237, 1.0905077326652577, 1.1143867425958924,
    1.1387886347566916, 1.1637248587775775, 1.189207115002721, 1.215247359980469, 1.241857812073484,
    1.2690509571917332, 1.2968395546510096, 1.3252366431597413, 1.3542555469368927, 1.383909881963832,
    1.4142135623730951, 1.4451808069770467, 1.4768261459394993, 1.5091644275934228, 1.5422108254079407,
    1.5759808451078865, 1.6104903319492543, 1.645755478153965, 1.681792830507429, 1.718619298122478,
    1.7562521603732995, 1.7947090750031072, 1.8340080864093424, 1.8741676341103, 1.9152065613971474,
    1.9571441241754002};

__device__ __forceinline__ double exp_neg(double x, const double *tab) {
  x = (x < -1000.0) ? -1000.0 : x;
  const double nf = rint(x * 46.16624130844683);             // 32 / ln 2
  double r = fma(nf, -0.02166084939249829, x);               // ln 2 / 32, high and low parts
  r = fma(nf, -7.247021293269686e-19, r);
  const int n = (int)nf;
  double p = fma(r, 1.0 / 720.0, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p *= r;
  const double t = tab[n & 31];
  return ldexp(fma(t, p, t), n >> 5);
}

// base kernel value from the squared scaled distance r2, kstar's version (table-based exponential)
template <int KIND>
__device__ __forceinline__ double base_kernel_fast(double r2, const double *tab) {
  if (KIND == 0) return exp_neg(-0.5 * r2, tab);
  double r = sqrt(r2);
  if (KIND == 1) return exp_neg(-r, tab);
  if (KIND == 2) {
    double t = r * 1.7320508075688772;
    return (1.0 + t) * exp_neg(-t, tab);
  }
  double t = r * 2.23606797749979;
  return (1.0 + t + t * t / 3.0) * exp_neg(-t, tab);
}

}  // namespace gpemu
