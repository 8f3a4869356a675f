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


// ---- cross-kernel on the matrix cores ---------------------------------------------------------------------------
// K_*[j][b] = k(|x_j - q_b|) needs the squared scaled distance of every (training row, query) pair.  Round 3 formed
// it on the vector ALUs (d subtractions + d FMAs per pair, padded to 8: 16 of the 29 fp64 instructions per element);
// here it is one rank-8 product on v_mfma_f64_16x16x4_f64:
//     -1/2 |x - q|^2 = x.q - 1/2 |x|^2 - 1/2 |q|^2
// with the training side AUGMENTED by its own -1/2 |x|^2 in slot d and the query side by a 1 there, so that two MFMAs
// (K = 8 >= d + 1) give x.q - 1/2 |x|^2 for a 16 x 16 tile of pairs and one add of the query's -1/2 |q|^2 -- a
// per-lane constant, every accumulator register of a lane belongs to the same query -- finishes it.  Both sides are
// CENTRED per PC (the mid-range of the scaled training coordinates) so that the cancellation error is ~d (range / 2 ls)^2
// eps, and for the RBF kernel scaled by sqrt(2^TB / ln 2) so that the product IS the exponent in units of ln 2 / 2^TB:
// the exponential then needs no range-reduction multiply (exp2_scaled).  Layouts (built on the host at model creation,
// gpemu_api.hip: build_kstar_operands): Xa[p][jt][ks][lane] = aug[16 jt + (lane & 15)][4 ks + (lane >> 4)] -- the A
// fragment of a j-tile is one coalesced 512-byte load -- and alf[p][jt][4 q + r] = alpha[16 jt + q + 4 r], the order in
// which the accumulator rows of lane group q = lane >> 4 come.
// ref: emulation.py:497 -> skl kernels.py:1553-1582 (RBF), 1708-1781 (Matern).
typedef double kd4 __attribute__((ext_vector_type(4)));

constexpr double KSTAR_LN2 = 0.6931471805599453;
template <int TB> constexpr double kstar_rbf_scale2() { return (double)(1 << TB) / KSTAR_LN2; }   // s^2: x' = s (x/ls - c)

// exp(y ln2 / 2^TB) for y <= ~0: n = rint(y) through the 1.5 2^52 shifter (the integer is the low word of the sum),
// f = y - n exactly, 2^(n / 2^TB) = 2^(n >> TB) tab[n & (2^TB - 1)], exp(f ln2 / 2^TB) - 1 by a Taylor polynomial whose
// degree follows the table size (|f| <= 1/2: truncation < 0.35 ulp), the power of two added into the exponent field.
// Arguments below -700 (natural units) are clamped: the result stays a normal number.  A NaN comes out as a finite
// value (v_max_f64 drops it): callers inject the NaN of a query on their own (kstar_mfma_block: the column's mean).
template <int TB>
__device__ __forceinline__ kd4 exp2_scaled4(kd4 y, const double *tab) {
  constexpr double H = KSTAR_LN2 / (double)(1 << TB);
  constexpr double SHIFT = 6755399441055744.0;   // 1.5 * 2^52
  // four independent arguments stage by stage: the table reads of all four are in flight together
  kd4 t, f, p, tb, v;
  int n[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    y[r] = fmax(y[r], -700.0 / H);
    t[r] = y[r] + SHIFT;
    n[r] = __double2loint(t[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tb[r] = tab[n[r] & ((1 << TB) - 1)];
#pragma unroll
  for (int r = 0; r < 4; ++r) f[r] = y[r] - (t[r] - SHIFT);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double q;
    if (TB <= 5) {
      q = fma(f[r], H * H * H * H * H * H / 720.0, H * H * H * H * H / 120.0);
      q = fma(q, f[r], H * H * H * H / 24.0);
    } else if (TB <= 7) {
      q = fma(f[r], H * H * H * H * H / 120.0, H * H * H * H / 24.0);
    } else {
      q = H * H * H * H / 24.0;
    }
    q = fma(q, f[r], H * H * H / 6.0);
    q = fma(q, f[r], H * H / 2.0);
    q = fma(q, f[r], H);
    p[r] = q * f[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double w = fma(tb[r], p[r], tb[r]);
    const int hi = __double2hiint(w) + ((n[r] >> TB) << 20);
    v[r] = __hiloint2double(hi, __double2loint(w));
  }
  return v;
}

// Matern-0.5 only: exp(-r) is not flat at r = 0, so a pair closer than ~1e-3.5 of the data's extent (a query ON a
// training point) needs r^2 to better than the product form's ~d eps |x|^2: those few pairs are recomputed from the
// coordinate differences, round 3's arithmetic (row-major scaled rows xs[j][8], query q inv_ls).
struct KstarDirect {
  const double *xs;    // [Npad][8] of this PC
  const double *inv;   // [8]
};
__device__ __forceinline__ double kstar_direct_r2(const KstarDirect &dir, const double *s_q, int64_t row, int col) {
  double r2 = 0.0;
#pragma unroll
  for (int dd = 0; dd < 8; ++dd) {
    const double df = s_q[col * 8 + dd] * dir.inv[dd] - dir.xs[row * 8 + dd];
    r2 = fma(df, df, r2);
  }
  return r2;
}

// kernel values of the four accumulator registers of a lane (one query column, four training rows)
//   KIND 0 (RBF): operands scaled, acc + hq = -1/2 r^2 2^TB / ln2;  hq = -1/2 |q'|^2
//   KIND 1, 2, 3 (Matern 0.5 / 1.5 / 2.5): operands unscaled, r^2 = max(-2 acc + |q'|^2, 0);  hq = |q'|^2
template <int KIND, int TB>
__device__ __forceinline__ kd4 kstar_value4(kd4 acc, double hq, const double *tab, const KstarDirect &dir,
                                            const double *s_q, int64_t row0, int col) {
  if (KIND == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += hq;
    return exp2_scaled4<TB>(acc, tab);
  }
  constexpr double C = (double)(1 << TB) / KSTAR_LN2;
  kd4 t, y, v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double r2 = fmax(fma(acc[r], -2.0, hq), 0.0);
    if (KIND == 1 && r2 < 1e-7 * (hq + 1.0)) r2 = kstar_direct_r2(dir, s_q, row0 + 4 * r, col);
    const double rr = sqrt(r2);
    t[r] = (KIND == 1) ? rr : rr * ((KIND == 2) ? 1.7320508075688772 : 2.23606797749979);
    y[r] = -C * t[r];
  }
  y = exp2_scaled4<TB>(y, tab);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (KIND == 1) v[r] = y[r];
    else if (KIND == 2) v[r] = (1.0 + t[r]) * y[r];
    else v[r] = (1.0 + t[r] + t[r] * t[r] / 3.0) * y[r];
  }
  return v;
}

// the augmented product of one 16 x 16 tile of (training row, query) pairs: KS chained MFMAs
template <int KS>
__device__ __forceinline__ kd4 kstar_tile_product(const double (&a)[KS], const double (&bq)[KS]) {
  kd4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bq[s], acc, 0, 0, 0);
  return acc;
}

// kernel values, stores and the mean's FMAs of one tile from its accumulator
template <int KIND, int TB, int ABL = 0>   // ABL: probe ablations (1: no stores, 2: no exponential, 3: nontemporal stores)
__device__ __forceinline__ void kstar_tile_finish(kd4 acc, double hq, kd4 al, double c, const double *s_tab, bool ragged,
                                                  int64_t row0, int64_t N, double *__restrict__ kout, int64_t Bcap, double &macc,
                                                  const KstarDirect &dir, const double *s_q, int col) {
  kd4 v = (ABL == 2) ? acc : kstar_value4<KIND, TB>(acc, hq, s_tab, dir, s_q, row0, col);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] += c;
  if (ragged) {                                           // wave-uniform: only the tiles that hold padded training rows
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (row0 + 4 * r >= N) v[r] = 0.0;                  // ... which contribute nothing
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (ABL == 3) __builtin_nontemporal_store(v[r], &kout[(int64_t)(4 * r) * Bcap]);
    else if (ABL != 1) kout[(int64_t)(4 * r) * Bcap] = v[r];
    macc = fma(al[r], v[r], macc);
  }
}

// The A fragments and alpha of a wave's j-tiles (loaded by the caller: where in its instruction stream is its choice).
template <int KS, int JTW>
struct KstarFrags {
  double a[JTW][KS];
  kd4 al[JTW];
};
template <int KS, int JTW, int NBW>
__device__ __forceinline__ void kstar_load_frags(KstarFrags<KS, JTW> &f, const double *__restrict__ Xa,
                                                 const double *__restrict__ alf, int64_t jt0, int lane, int wave) {
  constexpr int WC = 4 / NBW, WR = 4 / WC;
  const int wr = wave % WR, lq = lane >> 4;
#pragma unroll
  for (int jj = 0; jj < JTW; ++jj) {
    const int64_t jt = jt0 + wr * JTW + jj;
#pragma unroll
    for (int s = 0; s < KS; ++s) f.a[jj][s] = Xa[(jt * KS + s) * 64 + lane];
    f.al[jj] = *reinterpret_cast<const kd4 *>(alf + jt * 16 + lq * 4);
  }
}

// One workgroup (4 waves) of the cross-kernel: WR JTW j-tiles of 16 training rows x 64 query columns.
//   wave w: wave row wr = w % WR, wave column wc = w / WR (WC = 4 / NBW, WR = 4 / WC);
//           j-tiles jt0 + wr JTW .. + JTW - 1, b-tiles wc NBW .. + NBW - 1
//   NBW = 2, JTW = 2: 64 rows per workgroup (large batches);  NBW = 2, JTW = 1: 32 rows (small batches)
// s_q: LDS [64][8] raw (unscaled, zero-padded) query rows of the workgroup's columns; s_red: LDS [4][64].
// Returns, in wave 0, the workgroup's partial mean  sum_j alpha_j K[j][b]  of column b = lane (NaN if the query has one).
template <int KIND, int KS, int JTW, int NBW, int TB, int ABL = 0>
__device__ __forceinline__ double kstar_mfma_block(const double *s_q, const double *s_tab, double *s_red,
                                                   const KstarFrags<KS, JTW> &fr,
                                                   const double *__restrict__ qsc, const double *__restrict__ qof,
                                                   double c, int d, int64_t jt0, int64_t N, double *__restrict__ ks,
                                                   int64_t Bcap, const KstarDirect &dir, int lane, int wave) {
  constexpr int WC = 4 / NBW, WR = 4 / WC;
  const int wr = wave % WR, wc = wave / WR;
  const int ln = lane & 15, lq = lane >> 4;
  // B fragments: this lane's two (three) components of its NBW queries, scaled and centred; |q'|^2 per query
  double bq[NBW][KS], hq[NBW];
  double sc[KS], of[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) { sc[s] = qsc[4 * s + lq]; of[s] = qof[4 * s + lq]; }
#pragma unroll
  for (int bt = 0; bt < NBW; ++bt) {
    const int col = (wc * NBW + bt) * 16 + ln;
    double part = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int comp = 4 * s + lq;
      const double qv = (comp < 8) ? s_q[col * 8 + comp] : 0.0;
      const double v = fma(qv, sc[s], of[s]);
      bq[bt][s] = v;
      part = (comp < d) ? fma(v, v, part) : part;
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    hq[bt] = (KIND == 0) ? -0.5 * part : part;
  }
  double macc[NBW];
#pragma unroll
  for (int bt = 0; bt < NBW; ++bt) macc[bt] = 0.0;
  double *kcol = ks + (int64_t)lq * Bcap + wc * NBW * 16 + ln;
  // the wave's JTW x NBW tiles as one software-pipelined sequence: the MFMAs of tile i + 1 are issued before the
  // kernel values of tile i are formed, so their latency (and the wait states in front of the first read of an
  // accumulator) hides behind ~90 vector instructions instead of stalling the wave
  constexpr int NT = JTW * NBW;
  kd4 acc = kstar_tile_product<KS>(fr.a[0], bq[0]);
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int jj = i / NBW, bt = i % NBW;
    kd4 nxt = acc;
    if (i + 1 < NT) nxt = kstar_tile_product<KS>(fr.a[(i + 1) / NBW], bq[(i + 1) % NBW]);
    const int64_t jt = jt0 + wr * JTW + jj;
    kstar_tile_finish<KIND, TB, ABL>(acc, hq[bt], fr.al[jj], c, s_tab, (jt + 1) * 16 > N, jt * 16 + lq, N,
                                     kcol + jt * 16 * Bcap + bt * 16, Bcap, macc[bt], dir, s_q, wc * NBW * 16 + ln + bt * 16);
    acc = nxt;
  }
  // column sums: over the four lane groups, then over the wave rows in a fixed order
#pragma unroll
  for (int bt = 0; bt < NBW; ++bt) {
    double s = macc[bt];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (hq[bt] != hq[bt]) s = hq[bt];                // a NaN in the query: the column's mean says so
    if (lq == 0) s_red[wr * 64 + (wc * NBW + bt) * 16 + ln] = s;
  }
  __syncthreads();
  double sum = 0.0;
  if (wave == 0) {
    if (WR == 4) sum = (s_red[lane] + s_red[64 + lane]) + (s_red[128 + lane] + s_red[192 + lane]);
    else if (WR == 2) sum = s_red[lane] + s_red[64 + lane];
    else sum = s_red[lane];
  }
  return sum;
}

}  // namespace gpemu
