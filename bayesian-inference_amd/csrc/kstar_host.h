// Host side of the matrix-core cross-kernel (predict_dev.h: kstar_mfma_block): the augmented, centred, fragment-ordered
// training operands of one emulation group, built once at model creation.
// ref: emulation.py:497 -> skl kernels.py:1553-1582, 1708-1781 (the distance these operands reproduce).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace gpemu {

constexpr int KSTAR_TB = 6;            // log2 of the exponential's table size (exp2_scaled)

struct KstarHost {
  int ksteps = 2;                      // MFMA k-steps: 4 ksteps >= d + 1
  std::vector<double> Xa;              // [k][Npad/16][ksteps][64]
  std::vector<double> alf;             // [k][Npad/16][16]
  std::vector<double> qsc, qof;        // [k][4 ksteps]: q' = q qsc + qof  (slot d: 0, 1)
  std::vector<double> tab;             // [2^TB]: 2^(j / 2^TB)
};

// X_train [N][d], ls [k][d], alpha [k][N];  kind 0: RBF (operands scaled by sqrt(2^TB / ln 2)), else Matern (unscaled)
inline void build_kstar_operands(int64_t N, int64_t Npad, int64_t d, int64_t k, int kind, const double *X_train,
                                 const double *ls, const double *alpha, KstarHost &out, int tb = KSTAR_TB) {
  const int KS = (d + 1 <= 8) ? 2 : 3;
  const int64_t njt = Npad / 16;
  out.ksteps = KS;
  out.Xa.assign((size_t)(k * njt * KS * 64), 0.0);
  out.alf.assign((size_t)(k * njt * 16), 0.0);
  out.qsc.assign((size_t)(k * 4 * KS), 0.0);
  out.qof.assign((size_t)(k * 4 * KS), 0.0);
  out.tab.resize((size_t)1 << tb);
  for (int j = 0; j < (1 << tb); ++j) out.tab[j] = (double)exp2l((long double)j / (long double)(1 << tb));
  const double s = (kind == 0) ? std::sqrt((double)(1 << tb) / 0.6931471805599453) : 1.0;
  std::vector<double> aug((size_t)(4 * KS));
  for (int64_t p = 0; p < k; ++p) {
    // centre: mid-range of the scaled training coordinates u = X / ls (skl: X / length_scale)
    double cen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t dd = 0; dd < d; ++dd) {
      double lo = INFINITY, hi = -INFINITY;
      for (int64_t j = 0; j < N; ++j) {
        const double u = X_train[j * d + dd] / ls[p * d + dd];
        lo = std::fmin(lo, u);
        hi = std::fmax(hi, u);
      }
      cen[dd] = 0.5 * (lo + hi);
      out.qsc[(size_t)(p * 4 * KS + dd)] = s / ls[p * d + dd];
      out.qof[(size_t)(p * 4 * KS + dd)] = -s * cen[dd];
    }
    out.qof[(size_t)(p * 4 * KS + d)] = 1.0;                      // the query's 1 against the row's -1/2 |x'|^2
    for (int64_t j = 0; j < N; ++j) {
      std::fill(aug.begin(), aug.end(), 0.0);
      long double nx = 0.0L;
      for (int64_t dd = 0; dd < d; ++dd) {
        const double u = X_train[j * d + dd] / ls[p * d + dd];
        const double v = (u - cen[dd]) * s;
        aug[(size_t)dd] = v;
        nx += (long double)v * (long double)v;
      }
      aug[(size_t)d] = (double)(-0.5L * nx);
      const int64_t jt = j / 16, i = j % 16;
      for (int s4 = 0; s4 < KS; ++s4)
        for (int q = 0; q < 4; ++q)
          out.Xa[(size_t)(((p * njt + jt) * KS + s4) * 64 + q * 16 + i)] = aug[(size_t)(4 * s4 + q)];
      out.alf[(size_t)((p * njt + jt) * 16 + (i % 4) * 4 + i / 4)] = alpha[p * N + j];
    }
  }
}

}  // namespace gpemu
