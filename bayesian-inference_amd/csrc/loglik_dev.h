// Per-walker device functions of the low-rank likelihood shared by loglik_lowrank_kernel (k_loglik.hip) and the
// fused sampler front kernel (k_front.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "internal.h"

namespace gpemu {

// ---- per-walker evaluation --------------------------------------------------------------------
// One wave per walker, lane = PC index p (k <= KMAX <= 64), 4 walkers per workgroup.  Row p of the
// k x k matrices lives in lane p's registers; columns are exchanged with wave shuffles, so the k x k
// Cholesky and the triangular solve are throughput- rather than LDS-latency-bound.  The kernel also
//   * sums the partial means / partial ||W k_*||^2 written by kstar_kernel / trmm_vsq_kernel,
//   * applies the strict box prior (ref: log_posterior.py:63-64),
//   * optionally finishes the stretch move for its walker (accept / reject, state update, chain
//     record: emcee moves/red_blue.py), so a half-step needs no further launch.
// sums the partials of walker b; returns (mu, sd) of PC `lane`
template <int KP>   // power of two >= k, <= 64
__device__ __forceinline__ void walker_mean_sd(const double *__restrict__ mean_part,
                                               const double *__restrict__ vsq_part,
                                               const double *__restrict__ kdiag, double *mean_out,
                                               double *var_out, int64_t b, int64_t Bcap, int k,
                                               int nchunk, int nrb, int lane, double &mu, double &sd) {
  // lane = sub * KP + pc: SUBS = 64 / KP lanes share one PC's partial sums (independent loads in flight
  // instead of one long dependent chain), combined by xor-shuffles; lanes < k end up with the totals
  constexpr int SUBS = 64 / KP;
  const int pc = lane & (KP - 1), sub = lane / KP;
  double mu_p = 0.0, vs_p = 0.0;
  if (pc < k) {
    // parts of one (walker, PC) are contiguous
    const double *mp = mean_part + (b * k + pc) * nchunk;
    const double *vp = vsq_part + (b * k + pc) * nrb;
#pragma unroll 8
    for (int c = sub; c < nchunk; c += SUBS) mu_p += mp[c];
#pragma unroll 8
    for (int r = sub; r < nrb; r += SUBS) vs_p += vp[r];
  }
#pragma unroll
  for (int off = KP; off < 64; off <<= 1) {
    mu_p += __shfl_xor(mu_p, off);
    vs_p += __shfl_xor(vs_p, off);
  }
  mu = 0.0;
  sd = 0.0;
  if (lane < k) {
    mu = mu_p;
    double v = kdiag[lane] - vs_p;
    if (v < 0.0) v = 0.0;     // skl _gpr.py:479-485
    sd = sqrt(v);
    if (mean_out) mean_out[b * k + lane] = mu;
    if (var_out) var_out[b * k + lane] = sd * sd;
  }
}

// writes the log-posterior of proposal b and, if enabled, finishes the stretch move for its walker
// operands of the accept step, loaded at kernel start so that their (dependent: index -> walker state)
// latency overlaps the partial-sum loads and the factorisation
struct AcceptOperands {
  int w = 0;
  double oldlp = 0.0, factor = 0.0, logu = 0.0, xold = 0.0, xnew = 0.0;
};
__device__ __forceinline__ AcceptOperands load_accept_operands(const double *__restrict__ Xq, int64_t b, int lane,
                                                               const AcceptArgs &aa) {
  AcceptOperands ao;
  if (!aa.enabled) return ao;
  ao.w = aa.idx_s[b];
  ao.oldlp = aa.logp[ao.w];
  ao.factor = aa.factors[b];
  ao.logu = aa.logu[b];
  if (lane < DPAD) {
    ao.xold = aa.X[(int64_t)ao.w * DPAD + lane];
    ao.xnew = Xq[b * DPAD + lane];
  }
  return ao;
}

__device__ __forceinline__ void finish_walker(double total, double *__restrict__ out, int64_t b, int d, int lane,
                                              int accumulate, const AcceptArgs &aa, const AcceptOperands &ao) {
  if (accumulate) total += out[b];
  if (!aa.enabled) {
    if (lane == 0) out[b] = total;
    return;
  }
  const int w = ao.w;
  const double oldlp = ao.oldlp;
  if (total != total && lane == 0) atomicAdd(aa.flags, 1);  // emcee raises on NaN
  const bool acc = (ao.factor + total - oldlp) > ao.logu;
  const double xold = ao.xold;
  if (lane < DPAD && acc) aa.X[(int64_t)w * DPAD + lane] = ao.xnew;
  if (lane == 0) {
    out[b] = total;
    if (acc) {
      aa.logp[w] = total;
      aa.naccept[w] += 1;
    }
  }
  if (aa.chain) {
    if (lane < d) aa.chain[(int64_t)w * d + lane] = acc ? ao.xnew : xold;
    if (lane == 0) aa.lpchain[w] = acc ? total : oldlp;
  }
}

// value of lane `l` (compile-time / wave-uniform index) as a scalar: v_readlane, no LDS round trip
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <int KMAX>
__device__ __forceinline__ double sum_first_lanes(double v) {
  double s = readlane_f64(v, 0);
#pragma unroll
  for (int q = 1; q < KMAX; ++q) s += readlane_f64(v, q);
  return s;
}

// Low-rank log-likelihood of one walker held by one wave (lane = PC index; see k_loglik.hip for the algebra):
// `mu`, `sd` are this lane's predictive mean and standard deviation; `gpre`, `gl_pre`, `sc0_pre`, `sc1_pre` the first
// observable block's constants, requested by the caller ahead of the GP partial sums (PRE; without it `gpre` is not
// read and the first block loads its row like the others: KMAX > 16, where the prefetched row would cost 2 KMAX
// registers more).  k <= KMAX <= 32; the k x k matrix M = I + D^1/2 G D^1/2 is padded to KMAX x KMAX with the identity
// so the factorisation is branch-free, row `lane` lives in registers, every cross-lane operand is a v_readlane of a
// compile-time lane.  The padding adds exact zeros and pivots of exactly 1: the value does not depend on KMAX.
// -inf outside the box.
// the term of observable block o (see walker_loglik_lowrank below, which adds the blocks' terms in turn)
template <int KMAX, bool PRE = true>
__device__ __forceinline__ double lowrank_block_term(int o, double mu, double sd, const double *gpre, double gl_pre, double sc0_pre,
                                                     double sc1_pre, const double *__restrict__ G, const double *__restrict__ g0,
                                                     const double *__restrict__ scal, int k, int lane) {
  const double *Go = G + (int64_t)o * k * k;
  // row `lane` of G_o (symmetric: read column-wise so that the wave's loads coalesce)
  double row[KMAX];
  double h = 0.0;
  // KMAX > 16: every lane loads from a clamped, valid position and the select follows -- scalar base per q plus ONE
  // vector offset, where the predicated form keeps an address per q in vector registers (2 KMAX of them)
  const double *Gl = Go + ((lane < k) ? lane : k - 1);
  // ... and the lane index is made opaque per block, or the KMAX unit-matrix terms (lane == q ? 1 : 0) are kept in
  // 2 KMAX registers across the loop over the blocks (KMAX = 32: 262 -> 200 -> 136 VGPRs)
  int lane_q = lane;
  if (!PRE) asm volatile("" : "+v"(lane_q));
#pragma unroll
  for (int q = 0; q < KMAX; ++q) {
    double gq;
    if (PRE) gq = (o == 0) ? gpre[q] : ((q < k && lane < k) ? Go[q * k + lane] : 0.0);
    else gq = (q < k && lane < k) ? Gl[((q < k) ? q : k - 1) * k] : 0.0;
    h = fma(gq, readlane_f64(mu, q), h);
    row[q] = ((lane_q == q) ? 1.0 : 0.0) + sd * gq * readlane_f64(sd, q);
  }
  const double gl = (o == 0) ? gl_pre : ((lane < k) ? g0[(int64_t)o * k + lane] : 0.0);
  const double sc0 = (o == 0) ? sc0_pre : scal[2 * o], sc1 = (o == 0) ? sc1_pre : scal[2 * o + 1];
  h += gl;
  const double quadA = sum_first_lanes<KMAX>((lane < k) ? mu * (h + gl) : 0.0) + sc0;
  // right-looking Cholesky; y = L_M^-1 (sd o h) by forward substitution alongside.  One reciprocal
  // square root per pivot on the critical path; the logarithms of the pivots are taken after the
  // loop, one per lane in parallel.  Entries above the diagonal (lane < column) are never read.
  double y = (lane < k) ? sd * h : 0.0;
  double mypiv2 = 1.0;
#pragma unroll
  for (int jx = 0; jx < KMAX; ++jx) {
    const double piv2 = readlane_f64(row[jx], jx);
    // 1/sqrt: hardware estimate + two Newton steps (a third of the IEEE sqrt-and-divide sequence, which
    // sits on the serial path of every pivot); relative error ~1e-16
    double rinv = __builtin_amdgcn_rsq(piv2);
    rinv = fma(0.5 * rinv, fma(-piv2 * rinv, rinv, 1.0), rinv);
    rinv = fma(0.5 * rinv, fma(-piv2 * rinv, rinv, 1.0), rinv);
    if (lane == jx) mypiv2 = piv2;
    const double lj = row[jx] * rinv;                             // column jx of L (lanes >= jx)
    const double zj = readlane_f64(y, jx) * rinv;
    y = (lane == jx) ? zj : ((lane > jx) ? fma(-lj, zj, y) : y);
#pragma unroll
    for (int c = jx + 1; c < KMAX; ++c) row[c] = fma(-lj, readlane_f64(lj, c), row[c]);
  }
  const double logdiag = 0.5 * log(mypiv2);                       // lanes >= k hold pivot 1
  const double ww = sum_first_lanes<KMAX>(y * y);
  const double ldsum = sum_first_lanes<KMAX>(logdiag);
  return -0.5 * (quadA - ww) - 0.5 * (sc1 + 2.0 * ldsum);
}

template <int KMAX, bool PRE = true>
__device__ __forceinline__ double walker_loglik_lowrank(bool inside, double mu, double sd, const double *gpre,
                                                        double gl_pre, double sc0_pre, double sc1_pre,
                                                        const double *__restrict__ G, const double *__restrict__ g0,
                                                        const double *__restrict__ scal, int k, int nblk, int lane) {
  double total = -INFINITY;
  if (inside) {
    total = 0.0;
    for (int o = 0; o < nblk; ++o)
      total += lowrank_block_term<KMAX, PRE>(o, mu, sd, gpre, gl_pre, sc0_pre, sc1_pre, G, g0, scal, k, lane);
  }
  return total;
}

// General k (33..64), the k x k matrix of the walker in LDS (`M`, leading dimension `ldm` >= k + 1, private to the
// wave): the arithmetic of loglik_lowrank_lds_kernel, shared with the fused front kernel so that a sharded run
// reproduces the single-GPU chain bit for bit.
__device__ __forceinline__ double walker_loglik_lowrank_lds(bool inside, double mu, double sd,
                                                            const double *__restrict__ G, const double *__restrict__ g0,
                                                            const double *__restrict__ scal, int k, int nblk, int lane,
                                                            double *M, int ldm) {
  double total = -INFINITY;
  if (inside) {
    total = 0.0;
    for (int o = 0; o < nblk; ++o) {
      const double *Go = G + (int64_t)o * k * k;
      double h = 0.0, gl = (lane < k) ? g0[(int64_t)o * k + lane] : 0.0;
      for (int q = 0; q < k; ++q) {
        double gq = (lane < k) ? Go[q * k + lane] : 0.0;
        h = fma(gq, __shfl(mu, q), h);
        double sq = __shfl(sd, q);
        if (lane < k) M[lane * ldm + q] = ((lane == q) ? 1.0 : 0.0) + sd * gq * sq;
      }
      h += gl;
      double t = (lane < k) ? mu * (h + gl) : 0.0;
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
      const double quadA = t + scal[2 * o];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double logdiag = 0.0;
      for (int j = 0; j < k; ++j) {
        double piv = sqrt(M[j * ldm + j]);
        __builtin_amdgcn_wave_barrier();
        if (lane == j) {
          M[j * ldm + j] = piv;
          logdiag = log(piv);
        }
        if (lane > j && lane < k) M[lane * ldm + j] = M[lane * ldm + j] / piv;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane > j && lane < k) {
          double lij = M[lane * ldm + j];
          for (int c = j + 1; c <= lane; ++c) M[lane * ldm + c] -= lij * M[c * ldm + j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      double y = (lane < k) ? sd * h : 0.0;
      for (int j = 0; j < k; ++j) {
        double zj = __shfl(y, j) / M[j * ldm + j];
        if (lane == j) y = zj;
        if (lane > j && lane < k) y = fma(-M[lane * ldm + j], zj, y);
      }
      double ww = (lane < k) ? y * y : 0.0;
      double ldsum = logdiag;
      for (int off = 32; off > 0; off >>= 1) {
        ww += __shfl_xor(ww, off);
        ldsum += __shfl_xor(ldsum, off);
      }
      total += -0.5 * (quadA - ww) - 0.5 * (scal[2 * o + 1] + 2.0 * ldsum);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  return total;
}

}  // namespace gpemu
