// Host-side check of the matrix-core cross-kernel's operands (bayesian-inference_amd/csrc/kstar_host.h), built with g++
// by tests/test_kstar_host.py: emulates the device arithmetic -- q' = fma(q, qsc, qof), the augmented rank-8 product
// accumulated k-step by k-step in fp64 FMAs, the query's -1/2 |q'|^2 added last -- and prints, per kernel family, the
// largest error of the recovered squared scaled distance against the direct long-double evaluation
// (ref: skl kernels.py:1553-1582, 1708-1781: dists = pdist(X / length_scale)).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../bayesian-inference_amd/csrc/kstar_host.h"

int main(int argc, char **argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 203, d = argc > 2 ? atoll(argv[2]) : 6, k = argc > 3 ? atoll(argv[3]) : 3;
  const int64_t Npad = (N + 127) / 128 * 128, B = 37;
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  std::vector<double> X(N * d), ls(k * d), al(k * N), Q(B * d);
  for (auto &v : X) v = -3.0 + 7.0 * rnd();
  for (auto &v : ls) v = 0.3 + 4.0 * rnd();
  for (auto &v : al) v = rnd();
  for (auto &v : Q) v = -3.0 + 7.0 * rnd();
  for (int64_t dd = 0; dd < d; ++dd) Q[5 * d + dd] = X[11 * d + dd];      // a query ON a training point
  for (int kind = 0; kind < 2; ++kind) {
    gpemu::KstarHost h;
    gpemu::build_kstar_operands(N, Npad, d, k, kind, X.data(), ls.data(), al.data(), h);
    const int KS = h.ksteps;
    const double unit = kind == 0 ? (double)(1 << gpemu::KSTAR_TB) / 0.6931471805599453 : 1.0;   // RBF: exponent units
    long double worst = 0, worst_on = 0;
    bool layout_ok = true;
    for (int64_t p = 0; p < k; ++p)
      for (int64_t b = 0; b < B; ++b) {
        std::vector<double> qp(4 * KS);
        double nq = 0.0;
        for (int c = 0; c < 4 * KS; ++c) {
          const double qv = c < d ? Q[b * d + c] : 0.0;
          qp[c] = std::fma(qv, h.qsc[p * 4 * KS + c], h.qof[p * 4 * KS + c]);
          if (c < d) nq = std::fma(qp[c], qp[c], nq);
        }
        for (int64_t j = 0; j < N; ++j) {
          const int64_t jt = j / 16, i = j % 16;
          double acc = 0.0;
          for (int s4 = 0; s4 < KS; ++s4)
            for (int q = 0; q < 4; ++q)
              acc = std::fma(h.Xa[((p * (Npad / 16) + jt) * KS + s4) * 64 + q * 16 + i], qp[4 * s4 + q], acc);
          const double half_r2_units = -(acc - 0.5 * nq);                     // = 1/2 r^2 in the operands' units
          long double r2 = 0;
          for (int64_t dd = 0; dd < d; ++dd) {
            const long double df = ((long double)X[j * d + dd] - (long double)Q[b * d + dd]) / (long double)ls[p * d + dd];
            r2 += df * df;
          }
          const long double err = fabsl(2.0L * half_r2_units / unit - r2);     // absolute error of r^2
          if (b == 5 && j == 11) worst_on = fmaxl(worst_on, err);
          worst = fmaxl(worst, err / fmaxl(1.0L, r2));
          if (h.alf[(p * (Npad / 16) + jt) * 16 + (i % 4) * 4 + i / 4] != al[p * N + j]) layout_ok = false;
        }
      }
    for (int64_t p = 0; p < k && layout_ok; ++p)          // padded rows are all-zero fragments
      for (int64_t j = N; j < Npad; ++j)
        for (int s4 = 0; s4 < KS; ++s4)
          for (int q = 0; q < 4; ++q)
            if (h.Xa[((p * (Npad / 16) + j / 16) * KS + s4) * 64 + q * 16 + j % 16] != 0.0) layout_ok = false;
    printf("kind %d ksteps %d worst_rel_r2 %.3Le at_training_point %.3Le layout %s\n", kind, KS, worst, worst_on, layout_ok ? "ok" : "BAD");
  }
  return 0;
}
