// Gaussian log-likelihood of the emulator prediction against the data, low-rank form.
//
// Replaces  ref: log_posterior.py:63-64, 87-99, 104-146  (box prior, dY, Sigma = cov + diag(y_err^2),
// dpotrf + dpotrs per walker) for one emulation group.  With  s = scaler.scale_,  S_k = components_[:k]:
//     Sigma(theta) = A + U diag(var(theta)) U^T,   A = (C_unexpl / n_div) o (s s^T) + diag(y_err^2),
//     U = diag(s) S_k^T (F x k),   r(theta) = U m(theta) + r0,   r0 = scaler.mean_ - y_exp
// (ref: emulation.py:508-539).  Woodbury + the matrix-determinant lemma give
//     log p = -1/2 [ m^T G m + 2 m^T g0 + q0 - || L_M^-1 D^1/2 (G m + g0) ||^2 ]
//             -1/2 [ log det A + 2 sum log diag L_M ],     M = I + D^1/2 G D^1/2 = L_M L_M^T,
// with G = U^T A^-1 U, g0 = U^T A^-1 r0, q0 = r0^T A^-1 r0 computed once per (model, n_div) on the
// device (setup kernel below).  The symmetric D^1/2 form stays finite when a variance is clipped to 0.
//
// Observable blocks: the reference's merge step keeps only the within-observable blocks of the
// covariance (ref: emulation.py:370-388, SortEmulationGroupObservables.convert + nd_block_diag), so
// Sigma is block diagonal over the observables of the group and the likelihood is a sum over
// blocks o, each of the form above with its own (G_o, g0_o, q0_o, logdet A_o) and the same m, var.
#include <algorithm>

#include "internal.h"
#include "linalg_dev.h"
#include "loglik_dev.h"

namespace gpemu {

// ---- setup ----------------------------------------------------------------------------------
__global__ void build_A_kernel(const double *__restrict__ cun, const double *__restrict__ s,
                               const double *__restrict__ yerr, double *__restrict__ A, int F,
                               double inv_ndiv) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)F * F) return;
  int f = (int)(idx / F), g = (int)(idx - (int64_t)f * F);
  double v = cun[idx] * inv_ndiv * (s[f] * s[g]);
  if (f == g) v += yerr[f] * yerr[f];
  A[idx] = v;
}

// one workgroup per observable block o (features f0..f1): chol(A_o); Z = C^-1 [U_o | r0_o];
// G_o = Zu^T Zu; g0_o = Zu^T zr; q0_o = zr^T zr; logdet A_o
__global__ __launch_bounds__(CHOL_THREADS) void lik_setup_kernel(
    double *A, double *PT, double *Z /*[F][k+1]*/, const double *__restrict__ comp,
    const double *__restrict__ s, const double *__restrict__ smean, const double *__restrict__ yexp,
    const int *__restrict__ blk_start, double *G, double *g0, double *scal, int F, int k, int *info, int max_nf,
    int nch) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int o = blockIdx.x, nblk = gridDim.x;
  const int f0 = blk_start[o], nf = blk_start[o + 1] - f0;
  if (nf > max_nf) return;       // large blocks go through the blocked path (launch_lik_setup)
  const int k1 = k + nch;
  double *Ao = A + (int64_t)f0 * F + f0;
  double *Zo = Z + (int64_t)f0 * k1;
  double *PTo = PT + (int64_t)o * chol_scratch_size(F);
  for (int idx = tid; idx < nf * k1; idx += nthr) {
    int f = f0 + idx / k1, p = idx % k1;
    Zo[idx] = (p < k) ? s[f] * comp[(int64_t)p * F + f] : (smean[f] - yexp[(int64_t)(p - k) * F + f]);
  }
  wg_cholesky_lower(Ao, nf, F, PTo, info + o);
  for (int c0 = 0; c0 < k1; c0 += nthr)        // one thread per right-hand side
    wg_forward_solve_multi(Ao, nf, F, Zo + c0, k1, (k1 - c0 < nthr) ? (k1 - c0) : nthr);
  for (int idx = tid; idx < k1 * k1; idx += nthr) {
    int p = idx / k1, q = idx - p * k1;
    const bool wanted = (p < k && q < k) || (p < k && q >= k) || (p == q);
    if (!wanted) continue;
    double acc = 0.0;
    for (int f = 0; f < nf; ++f) acc = fma(Zo[(int64_t)f * k1 + p], Zo[(int64_t)f * k1 + q], acc);
    if (p < k && q < k) G[((int64_t)o * k + p) * k + q] = acc;
    else if (p < k) g0[((int64_t)(q - k) * nblk + o) * k + p] = acc;
    else scal[((int64_t)(p - k) * nblk + o) * 2] = acc;
  }
  double ld = 0.0;
  for (int f = tid; f < nf; f += nthr) ld += log(Ao[(int64_t)f * F + f]);
  ld = wg_sum(ld);
  if (tid < nch) scal[((int64_t)tid * nblk + o) * 2 + 1] = 2.0 * ld;
}

// ---- large observable blocks: blocked MFMA Cholesky + triangular inverse instead of one workgroup ------------
constexpr int LIK_BLOCKED_MIN = 256;   // blocks up to this many features stay with the single-workgroup kernel

// Ab[Np][Np] = A_o (lower triangle) padded with the identity
__global__ void lik_pad_block_kernel(const double *__restrict__ A, int ld, int nf, double *__restrict__ Ab, int Np) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= Np) return;
  double v = 0.0;
  if (r < nf && c <= r) v = A[(int64_t)r * ld + c];
  else if (r >= nf && c == r) v = 1.0;
  Ab[(int64_t)r * Np + c] = v;
}

// Z[i][c] = sum_{j <= i} W[i][j] R[j][c],  R = [U_o | r0_o]  (nf x (k+1)); one wave per row i, lanes over j
__global__ __launch_bounds__(256) void lik_z_kernel(const double *__restrict__ W, int Np, int nf, int f0, int F, int k,
                                                    const double *__restrict__ comp, const double *__restrict__ s,
                                                    const double *__restrict__ smean, const double *__restrict__ yexp,
                                                    double *__restrict__ Z, int nch) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nf) return;
  const int k1 = k + nch;
  for (int c = 0; c < k1; ++c) {
    double acc = 0.0;
    for (int j = lane; j <= i; j += 64) {
      const int f = f0 + j;
      const double r = (c < k) ? s[f] * comp[(int64_t)c * F + f] : (smean[f] - yexp[(int64_t)(c - k) * F + f]);
      acc = fma(W[(int64_t)i * Np + j], r, acc);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) Z[(int64_t)(f0 + i) * k1 + c] = acc;
  }
}

// G_o = Zu^T Zu, g0_o = Zu^T zr, q0_o = zr^T zr, logdet A_o = 2 sum log diag C: one workgroup
__global__ __launch_bounds__(1024) void lik_gram_kernel(const double *__restrict__ Z, const double *__restrict__ Lb, int Np,
                                                        int nf, int f0, int k, int o, double *G, double *g0, double *scal,
                                                        int nch, int nblk) {
  const int tid = threadIdx.x, k1 = k + nch;
  const double *Zo = Z + (int64_t)f0 * k1;
  for (int idx = tid; idx < k1 * k1; idx += 1024) {
    const int p = idx / k1, q = idx - p * k1;
    const bool wanted = (p < k && q < k) || (p < k && q >= k) || (p == q);
    if (!wanted) continue;
    double acc = 0.0;
    for (int f = 0; f < nf; ++f) acc = fma(Zo[(int64_t)f * k1 + p], Zo[(int64_t)f * k1 + q], acc);
    if (p < k && q < k) G[((int64_t)o * k + p) * k + q] = acc;
    else if (p < k) g0[((int64_t)(q - k) * nblk + o) * k + p] = acc;
    else scal[((int64_t)(p - k) * nblk + o) * 2] = acc;
  }
  double ld = 0.0;
  for (int f = tid; f < nf; f += 1024) ld += log(Lb[(int64_t)f * Np + f]);
  ld = wg_sum(ld);
  if (tid < nch) scal[((int64_t)tid * nblk + o) * 2 + 1] = 2.0 * ld;
}

// hstart: the observable block boundaries (host copy).  Small blocks: one workgroup each (lik_setup_kernel, all of
// them in one launch, larger ones skipped); blocks of more than LIK_BLOCKED_MIN features: blocked Cholesky with MFMA
// trailing updates, W = C^-1 by the blocked triangular inverse, Z = W [U | r0].
int launch_lik_setup(gpemu_model *m, const std::vector<int> &hstart, double *dA, double *dPT, double *dZ, int *dinfo,
                     hipStream_t st) {
  const int F = (int)m->F, k = (int)m->k;
  int64_t n = (int64_t)F * F;
  hipLaunchKernelGGL(build_A_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->cunexpl,
                     m->sscale, m->yerr, dA, F, 1.0 / m->n_div);
  hipLaunchKernelGGL(lik_setup_kernel, dim3((unsigned)m->nblk), dim3(CHOL_THREADS), 0, st, dA, dPT, dZ,
                     m->comp, m->sscale, m->smean, m->yexp, m->blk_start, m->G, m->g0, m->scal, F,
                     k, dinfo, LIK_BLOCKED_MIN, m->lik_chains);
  GP_HIP(hipGetLastError());
  int64_t maxnp = 0;
  for (size_t o = 0; o + 1 < hstart.size(); ++o) {
    const int nf = hstart[o + 1] - hstart[o];
    if (nf > LIK_BLOCKED_MIN) maxnp = std::max<int64_t>(maxnp, round_up(nf, 64));
  }
  if (maxnp == 0) return GPEMU_OK;
  double *Ab = nullptr, *Dinv = nullptr, *W = nullptr, *T = nullptr;
  hipError_t e = hipMalloc((void **)&Ab, sizeof(double) * maxnp * maxnp);
  if (e == hipSuccess) e = hipMalloc((void **)&Dinv, sizeof(double) * maxnp * 64);
  if (e == hipSuccess) e = hipMalloc((void **)&W, sizeof(double) * maxnp * maxnp);
  if (e == hipSuccess) e = hipMalloc((void **)&T, sizeof(double) * maxnp * maxnp);
  int rc = GPEMU_OK;
  if (e != hipSuccess) { set_error("likelihood_setup: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  for (size_t o = 0; o + 1 < hstart.size() && rc == GPEMU_OK; ++o) {
    const int f0 = hstart[o], nf = hstart[o + 1] - f0;
    if (nf <= LIK_BLOCKED_MIN) continue;
    const int Np = (int)round_up(nf, 64);
    hipLaunchKernelGGL(lik_pad_block_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)Np), dim3(256), 0, st,
                       dA + (int64_t)f0 * F + f0, F, nf, Ab, Np);
    rc = device_cholesky_blocked(Ab, Np, Dinv, dinfo + o, st);
    if (rc == GPEMU_OK) rc = device_trtri_blocked(Ab, Np, Dinv, W, T, st);
    if (rc != GPEMU_OK) break;
    hipLaunchKernelGGL(lik_z_kernel, dim3((unsigned)((nf + 3) / 4)), dim3(256), 0, st, W, Np, nf, f0, F, k, m->comp,
                       m->sscale, m->smean, m->yexp, dZ, m->lik_chains);
    hipLaunchKernelGGL(lik_gram_kernel, dim3(1), dim3(1024), 0, st, dZ, Ab, Np, nf, f0, k, (int)o, m->G, m->g0, m->scal,
                       m->lik_chains, (int)m->nblk);
    if (hipGetLastError() != hipSuccess) { set_error("likelihood_setup: launch failed"); rc = GPEMU_ERR_HIP; }
  }
  if (rc == GPEMU_OK && hipStreamSynchronize(st) != hipSuccess) { set_error("likelihood_setup: sync failed"); rc = GPEMU_ERR_HIP; }
  (void)hipFree(Ab); (void)hipFree(Dinv); (void)hipFree(W); (void)hipFree(T);
  return rc;
}

// k <= KMAX <= 32.  The k x k matrix M = I + D^1/2 G D^1/2 is padded to KMAX x KMAX with the identity, so
// the factorisation below is branch-free; lane = row, the row lives in registers, and every cross-lane
// operand is a v_readlane of a compile-time lane.  (Round 4: KMAX 20 ... 32 -- the shipped analysis has a group of 25
// PCs, which took the LDS form below: 39.9 us per launch at 100 proposals against 7.7 us for 11 PCs here.)
template <int KMAX>
__global__ __launch_bounds__(256) void loglik_lowrank_kernel(
    const double *__restrict__ Xq, const double *__restrict__ lo, const double *__restrict__ hi,
    const double *__restrict__ mean_part, const double *__restrict__ vsq_part,
    const double *__restrict__ kdiag, const double *__restrict__ G, const double *__restrict__ g0,
    const double *__restrict__ scal, double *__restrict__ out, double *__restrict__ mean_out,
    double *__restrict__ var_out, int64_t B, int64_t Bcap, int d, int k, int nchunk, int nrb,
    int nblk, int accumulate, AcceptArgs aa) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= B) return;  // whole wave exits together; no workgroup barriers below
  if (aa.chain_per) {  // several chains stacked: this row's chain selects the data constants
    const int64_t ch = (aa.first + b) / aa.chain_per;
    g0 += ch * nblk * k;
    scal += ch * 2 * nblk;
  }

  // everything that does not depend on the GP stage is requested first: the accept operands (a dependent
  // index -> state chain) and the first observable block's constants
  const AcceptOperands ao = load_accept_operands(Xq, b, lane, aa);
  constexpr bool PRE = KMAX <= 16;
  double gpre[PRE ? KMAX : 1];
  if (PRE) {
#pragma unroll
    for (int q = 0; q < KMAX; ++q) gpre[q] = (q < k && lane < k) ? G[q * k + lane] : 0.0;
  }
  const double gl_pre = (lane < k) ? g0[lane] : 0.0;
  const double sc0_pre = scal[0], sc1_pre = scal[1];

  bool in = true;
  if (lane < d) in = (Xq[b * DPAD + lane] > lo[lane]) && (Xq[b * DPAD + lane] < hi[lane]);
  const bool inside = __all(in);

  double mu, sd;
  walker_mean_sd<(KMAX <= 16 ? 16 : 32)>(mean_part, vsq_part, kdiag, mean_out, var_out, b, Bcap, k, nchunk, nrb, lane, mu, sd);
  const double total = walker_loglik_lowrank<KMAX, PRE>(inside, mu, sd, gpre, gl_pre, sc0_pre, sc1_pre, G, g0, scal, k, nblk, lane);
  finish_walker(total, out, b, d, lane, accumulate, aa, ao);
}

// General k (33..64): the k x k matrix of each walker lives in LDS (one wave per walker, lane = row).
__global__ __launch_bounds__(256) void loglik_lowrank_lds_kernel(
    const double *__restrict__ Xq, const double *__restrict__ lo, const double *__restrict__ hi,
    const double *__restrict__ mean_part, const double *__restrict__ vsq_part,
    const double *__restrict__ kdiag, const double *__restrict__ G, const double *__restrict__ g0,
    const double *__restrict__ scal, double *__restrict__ out, double *__restrict__ mean_out,
    double *__restrict__ var_out, int64_t B, int64_t Bcap, int d, int k, int nchunk, int nrb,
    int nblk, int accumulate, AcceptArgs aa) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= B) return;
  if (aa.chain_per) {
    const int64_t ch = (aa.first + b) / aa.chain_per;
    g0 += ch * nblk * k;
    scal += ch * 2 * nblk;
  }
  double *M = smem + (size_t)wave * k * (k + 1);
  const int ldm = k + 1;
  bool in = true;
  if (lane < d) in = (Xq[b * DPAD + lane] > lo[lane]) && (Xq[b * DPAD + lane] < hi[lane]);
  const bool inside = __all(in);
  double mu, sd;
  if (k <= 32) walker_mean_sd<32>(mean_part, vsq_part, kdiag, mean_out, var_out, b, Bcap, k, nchunk, nrb, lane, mu, sd);
  else walker_mean_sd<64>(mean_part, vsq_part, kdiag, mean_out, var_out, b, Bcap, k, nchunk, nrb, lane, mu, sd);
  const double total = walker_loglik_lowrank_lds(inside, mu, sd, G, g0, scal, k, nblk, lane, M, ldm);
  finish_walker(total, out, b, d, lane, accumulate, aa, load_accept_operands(Xq, b, lane, aa));
}

// ---- several emulation groups in one launch ----------------------------------------------------------------------
// The log-posterior of a proposal is the sum of its groups' log-likelihoods (block-diagonal covariance per group,
// ref: emulation.py:346-406 -> log_posterior.py:87-101).  One launch per group adds its term to `out` in turn; here the
// groups of a walker are taken by different WAVES of one workgroup at the same time -- their k x k factorisations are
// the serial part of the half-step -- and summed in group order afterwards: the same additions, the same bits.
// 3 or 4 groups: one walker per workgroup; 2: two; more: the waves take several groups each.  k <= 32 in every group.
constexpr int LL_GROUPS_MAX = 8;
struct LoglikGroup {
  const double *lo, *hi, *mean_part, *vsq_part, *kdiag, *G, *g0, *scal;
  double *mean_out, *var_out;
  int64_t Bcap;
  int k, nchunk, nrb, nblk;
};
struct LoglikGroups {
  LoglikGroup g[LL_GROUPS_MAX];
  int ng;
};

template <int KMAX>
__device__ __forceinline__ double group_loglik(const LoglikGroup &gr, bool inside, int64_t b, int lane) {
  constexpr bool PRE = KMAX <= 16;
  const int k = gr.k;
  double gpre[PRE ? KMAX : 1];
  if (PRE) {
#pragma unroll
    for (int q = 0; q < KMAX; ++q) gpre[q] = (q < k && lane < k) ? gr.G[q * k + lane] : 0.0;
  }
  const double gl_pre = (lane < k) ? gr.g0[lane] : 0.0;
  const double sc0_pre = gr.scal[0], sc1_pre = gr.scal[1];
  double mu, sd;
  walker_mean_sd<(KMAX <= 16 ? 16 : 32)>(gr.mean_part, gr.vsq_part, gr.kdiag, gr.mean_out, gr.var_out, b, gr.Bcap, k,
                                         gr.nchunk, gr.nrb, lane, mu, sd);
  return walker_loglik_lowrank<KMAX, PRE>(inside, mu, sd, gpre, gl_pre, sc0_pre, sc1_pre, gr.G, gr.g0, gr.scal, k, gr.nblk, lane);
}

__global__ __launch_bounds__(256) void loglik_groups_kernel(const double *__restrict__ Xq, LoglikGroups lg,
                                                            double *__restrict__ out, int64_t B, int d, int accumulate,
                                                            AcceptArgs aa) {
  __shared__ double s_lp[2][LL_GROUPS_MAX];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpw = (lg.ng <= 2) ? 2 : 1, wv = 4 / wpw;        // walkers per workgroup, waves per walker
  const int slot = wave / wv, gw = wave % wv;
  const int64_t b = (int64_t)blockIdx.x * wpw + slot;
  const bool active = b < B;                                 // (wave-uniform; every wave reaches the barrier)
  AcceptOperands ao;
  if (active) {
    if (gw == 0) ao = load_accept_operands(Xq, b, lane, aa);
    for (int g = gw; g < lg.ng; g += wv) {
      const LoglikGroup &gr = lg.g[g];
      bool in = true;
      if (lane < d) in = (Xq[b * DPAD + lane] > gr.lo[lane]) && (Xq[b * DPAD + lane] < gr.hi[lane]);
      const bool inside = __all(in);
      double lp;
      // (the value does not depend on KMAX: loglik_dev.h)
      if (gr.k <= 4) lp = group_loglik<4>(gr, inside, b, lane);
      else if (gr.k <= 8) lp = group_loglik<8>(gr, inside, b, lane);
      else if (gr.k <= 12) lp = group_loglik<12>(gr, inside, b, lane);
      else if (gr.k <= 16) lp = group_loglik<16>(gr, inside, b, lane);
      else if (gr.k <= 20) lp = group_loglik<20>(gr, inside, b, lane);
      else if (gr.k <= 24) lp = group_loglik<24>(gr, inside, b, lane);
      else if (gr.k <= 28) lp = group_loglik<28>(gr, inside, b, lane);
      else lp = group_loglik<32>(gr, inside, b, lane);
      if (lane == 0) s_lp[slot][g] = lp;
    }
  }
  __syncthreads();
  if (active && gw == 0) {
    double total = s_lp[slot][0];
    for (int g = 1; g < lg.ng; ++g) total = s_lp[slot][g] + total;     // (a launch per group: total_g + out[b])
    finish_walker(total, out, b, d, lane, accumulate, aa, ao);
  }
}

// ---- the observable blocks of a proposal on different waves -----------------------------------------------------------
// The covariance is block diagonal over the OBSERVABLES of a group (ref: emulation.py:370-388), so a group's log-likelihood
// is a sum of per-observable terms, each with its own k x k factorisation -- the serial part of the half-step.  The
// reference's shipped configuration has 2 + 4 + 10 observables in its groups of 5 / 11 / 25 PCs: one wave per group
// factorises ten 25 x 25 matrices in turn, 78 us per launch (golden G7, 200 walkers: 199 us per step).  Here every
// (group, observable) pair of a proposal is a TASK on a wave of its own: four tasks per workgroup (one wave per SIMD: twelve
// on one CU were bound by that CU's vector ALUs, 24.8 us per launch), `nwg` workgroups per proposal, dealt longest first.
// A workgroup leaves its terms at the device's coherence point and takes a ticket; the LAST one to arrive for a proposal
// (nobody waits for anybody) reads them all, adds them in the order of the serial loop -- block after block, then group
// after group: the same bits -- and finishes the stretch move.
constexpr int LL_TASKS_MAX = 64, LL_TASK_WAVES = 4;
struct LoglikTasks {
  LoglikGroup g[LL_GROUPS_MAX];
  int ng, ntask, nwg;
  int first[LL_GROUPS_MAX + 1];              // slot of group g's block 0 in the list of terms
  // tasks in the order the waves take them: wave w of workgroup j has [wstart[4 j + w], wstart[4 j + w + 1])
  unsigned char tg[LL_TASKS_MAX], to[LL_TASKS_MAX];
  unsigned char wstart[LL_TASKS_MAX + 1];
  double *terms;                             // [B][ntask] the proposals' terms on their way to the last workgroup
  unsigned *tickets;                         // [B] workgroups of the proposal that have delivered (back to 0 by the last)
};

template <int KMAX>
__device__ __forceinline__ double task_term(const LoglikGroup &gr, int o, int64_t b, int lane) {
  constexpr bool PRE = KMAX <= 16;
  const int k = gr.k;
  // the block's own constants as "block 0" of lowrank_block_term: requested ahead of the partial sums where the registers
  // allow (PRE), the same values either way
  const double *G = gr.G + (int64_t)o * k * k, *g0 = gr.g0 + (int64_t)o * k, *scal = gr.scal + 2 * o;
  double gpre[PRE ? KMAX : 1];
  if (PRE) {
#pragma unroll
    for (int q = 0; q < KMAX; ++q) gpre[q] = (q < k && lane < k) ? G[q * k + lane] : 0.0;
  }
  const double gl_pre = (lane < k) ? g0[lane] : 0.0;
  const double sc0_pre = scal[0], sc1_pre = scal[1];
  double mu, sd;
  walker_mean_sd<(KMAX <= 16 ? 16 : 32)>(gr.mean_part, gr.vsq_part, gr.kdiag, o == 0 ? gr.mean_out : nullptr,
                                         o == 0 ? gr.var_out : nullptr, b, gr.Bcap, k, gr.nchunk, gr.nrb, lane, mu, sd);
  return lowrank_block_term<KMAX, PRE>(0, mu, sd, gpre, gl_pre, sc0_pre, sc1_pre, G, g0, scal, k, lane);
}

// grid: workgroup j of proposal b at index j B + b (the workgroups with the longest tasks first)
__global__ __launch_bounds__(64 * LL_TASK_WAVES) void loglik_tasks_kernel(const double *__restrict__ Xq, LoglikTasks lt,
                                                                          double *__restrict__ out, int64_t B, int d, int accumulate,
                                                                          AcceptArgs aa) {
  __shared__ double s_term[LL_TASKS_MAX];
  __shared__ unsigned s_ticket;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t b = (int64_t)blockIdx.x % B;
  const int j = (int)((int64_t)blockIdx.x / B);
  // (every workgroup asks for the accept step's operands at once -- a dependent index -> state chain: whichever turns out
  // to be the last has them by then)
  AcceptOperands ao;
  if (wave == 0) ao = load_accept_operands(Xq, b, lane, aa);
  const int slot = j * LL_TASK_WAVES + wave;
  for (int t = lt.wstart[slot]; t < lt.wstart[slot + 1]; ++t) {
    const int g = lt.tg[t], o = lt.to[t];
    const LoglikGroup &gr = lt.g[g];
    bool in = true;
    if (lane < d) in = (Xq[b * DPAD + lane] > gr.lo[lane]) && (Xq[b * DPAD + lane] < gr.hi[lane]);
    double term = 0.0;
    if (__all(in)) {                                         // (outside the box the group's term is -inf whatever the blocks say)
      if (gr.k <= 4) term = task_term<4>(gr, o, b, lane);
      else if (gr.k <= 8) term = task_term<8>(gr, o, b, lane);
      else if (gr.k <= 12) term = task_term<12>(gr, o, b, lane);
      else if (gr.k <= 16) term = task_term<16>(gr, o, b, lane);
      else if (gr.k <= 20) term = task_term<20>(gr, o, b, lane);
      else if (gr.k <= 24) term = task_term<24>(gr, o, b, lane);
      else if (gr.k <= 28) term = task_term<28>(gr, o, b, lane);
      else term = task_term<32>(gr, o, b, lane);
    }
    if (lane == 0) {
      s_term[lt.first[g] + o] = term;
      // straight to the device's coherence point: the workgroup that reads it may run on another XCD, whose L2 may still
      // hold the half-step before's value of this address
      if (lt.nwg > 1) __hip_atomic_store(lt.terms + b * lt.ntask + lt.first[g] + o, term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (lt.nwg > 1) {
    // the stores are acknowledged (s_waitcnt), the waves have met: one relaxed increment delivers them.  (No release /
    // acquire pair at device scope: it writes back / invalidates an XCD's whole L2, profiles/r05_halfstep_small.txt.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(lt.tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (unsigned)(lt.nwg - 1)) return;          // not the last: done (whole workgroup)
    if (wave == 0) {
      if (lane == 0) __hip_atomic_store(lt.tickets + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // for the next launch
      if (lane < lt.ntask) s_term[lane] = __hip_atomic_load(lt.terms + b * lt.ntask + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (wave == 0) {
    double total = 0.0;
    for (int g = 0; g < lt.ng; ++g) {
      const LoglikGroup &gr = lt.g[g];
      bool in = true;
      if (lane < d) in = (Xq[b * DPAD + lane] > gr.lo[lane]) && (Xq[b * DPAD + lane] < gr.hi[lane]);
      double lp = -INFINITY;
      if (__all(in)) {
        lp = 0.0;
        for (int o = lt.first[g]; o < lt.first[g + 1]; ++o) lp += s_term[o];      // walker_loglik_lowrank's loop
      }
      total = (g == 0) ? lp : lp + total;                      // loglik_groups_kernel's sum (a launch per group: total_g + out[b])
    }
    finish_walker(total, out, b, d, lane, accumulate, aa, ao);
  }
}

// GPEMU_ERR_UNSUPPORTED (nothing launched, no error set) where it does not apply: no group with more than one observable
// block, more than 64 blocks in all, more than 32 PCs in a group, stacked chains; GPEMU_NO_LOGLIK_TASKS (tests; read per call)
int launch_loglik_tasks(gpemu_model *const *ms, int ng, int64_t B, const double *dXq, double *dout, int accumulate,
                        hipStream_t st, const AcceptArgs *aa) {
  if (getenv("GPEMU_NO_LOGLIK_TASKS") != nullptr || ng < 1 || ng > LL_GROUPS_MAX || B < 1) return GPEMU_ERR_UNSUPPORTED;
  if (aa && aa->chain_per != 0) return GPEMU_ERR_UNSUPPORTED;
  int ntask = 0;
  for (int g = 0; g < ng; ++g) {
    if (ms[g]->k > 32 || ms[g]->nblk < 1) return GPEMU_ERR_UNSUPPORTED;
    ntask += (int)ms[g]->nblk;
  }
  if (ntask <= ng || ntask > LL_TASKS_MAX) return GPEMU_ERR_UNSUPPORTED;
  gpemu_model *m0 = ms[0];
  LoglikTasks lt;
  memset(&lt, 0, sizeof(lt));
  lt.ng = ng;
  lt.ntask = ntask;
  lt.nwg = (ntask + LL_TASK_WAVES - 1) / LL_TASK_WAVES;
  // A wave per task pays where the proposals alone leave most of the chip idle (the shipped ensembles: 50 - 100 per half-step:
  // 78 -> 19 us per launch) or a proposal has many blocks; measured at C3 size, 512 proposals per half-step
  // (tools/time_c3_blocks.py): 10 blocks 0.2714 -> 0.2634 ms per step, 5 blocks 0.2476 -> 0.2499, 2 blocks 0.2344 -> 0.2364
  {
    const char *e = getenv("GPEMU_LOGLIK_TASKS_MAX_ROWS");
    if (B > (e ? atoll(e) : 256) && ntask < 8) return GPEMU_ERR_UNSUPPORTED;
  }
  if (lt.nwg > 1) {
    // the terms' way to the last workgroup: per model (the first group's), grown with the batch
    if (m0->lik_terms_cap < B) {
      GP_HIP(hipStreamSynchronize(st));
      (void)hipFree(m0->lik_terms);
      (void)hipFree(m0->lik_tickets);
      m0->lik_terms = nullptr; m0->lik_tickets = nullptr; m0->lik_terms_cap = 0;
      const int64_t cap = round_up(B, 128);
      GP_HIP(hipMalloc((void **)&m0->lik_terms, sizeof(double) * (size_t)cap * LL_TASKS_MAX));
      GP_HIP(hipMalloc((void **)&m0->lik_tickets, sizeof(unsigned) * (size_t)cap));
      GP_HIP(hipMemsetAsync(m0->lik_tickets, 0, sizeof(unsigned) * (size_t)cap, st));
      m0->lik_terms_cap = cap;
    }
    lt.terms = m0->lik_terms;
    lt.tickets = m0->lik_tickets;
  }
  for (int g = 0; g < ng; ++g) {
    const gpemu_model *m = ms[g];
    const Workspace &w = m->ws;
    lt.g[g] = LoglikGroup{m->lo, m->hi, w.mean_part, w.vsq_part, m->kdiag, m->G, m->g0, m->scal, w.mean, w.var, w.Bcap,
                          (int)m->k, w.cur_nchunk, w.cur_nrb, (int)m->nblk};
    lt.first[g + 1] = lt.first[g] + (int)m->nblk;
  }
  // who takes which task: longest first, each to the wave with the least so far (waves of the first workgroups first: they
  // are dispatched first).  Depends on the groups' (PCs, blocks) only: kept per host thread for the next launch.
  struct Deal { int ng, k[LL_GROUPS_MAX], nblk[LL_GROUPS_MAX]; unsigned char tg[LL_TASKS_MAX], to[LL_TASKS_MAX], wstart[LL_TASKS_MAX + 1]; };
  static thread_local std::vector<Deal> deals;
  const Deal *deal = nullptr;
  for (const Deal &dl : deals) {
    bool same = dl.ng == ng;
    for (int g = 0; same && g < ng; ++g) same = dl.k[g] == (int)ms[g]->k && dl.nblk[g] == (int)ms[g]->nblk;
    if (same) { deal = &dl; break; }
  }
  if (!deal) {
    struct T { double cost; int g, o; };
    std::vector<T> tasks;
    for (int g = 0; g < ng; ++g) {
      const double kk = (double)((ms[g]->k + 3) / 4 * 4);
      for (int o = 0; o < (int)ms[g]->nblk; ++o) tasks.push_back(T{kk * kk + 8.0 * kk + 40.0, g, o});   // (factorisation + row loads + fixed part)
    }
    std::stable_sort(tasks.begin(), tasks.end(), [](const T &a, const T &b) { return a.cost > b.cost; });
    const int nslots = lt.nwg * LL_TASK_WAVES;
    std::vector<std::vector<T>> per(nslots);
    std::vector<double> load(nslots, 0.0);
    for (const T &t : tasks) {
      int best = 0;
      for (int w2 = 1; w2 < nslots; ++w2)
        if (load[w2] < load[best]) best = w2;
      per[best].push_back(t);
      load[best] += t.cost;
    }
    Deal dl;
    memset(&dl, 0, sizeof(dl));
    dl.ng = ng;
    for (int g = 0; g < ng; ++g) { dl.k[g] = (int)ms[g]->k; dl.nblk[g] = (int)ms[g]->nblk; }
    int n = 0;
    for (int w2 = 0; w2 < nslots; ++w2) {
      dl.wstart[w2] = (unsigned char)n;
      for (const T &t : per[w2]) { dl.tg[n] = (unsigned char)t.g; dl.to[n] = (unsigned char)t.o; ++n; }
    }
    for (int w2 = nslots; w2 <= LL_TASKS_MAX; ++w2) dl.wstart[w2] = (unsigned char)n;
    if (deals.size() >= 16) deals.erase(deals.begin());
    deals.push_back(dl);
    deal = &deals.back();
  }
  memcpy(lt.tg, deal->tg, sizeof(lt.tg));
  memcpy(lt.to, deal->to, sizeof(lt.to));
  memcpy(lt.wstart, deal->wstart, sizeof(lt.wstart));
  const AcceptArgs a = aa ? *aa : AcceptArgs();
  hipLaunchKernelGGL(loglik_tasks_kernel, dim3((unsigned)(B * lt.nwg)), dim3(64 * LL_TASK_WAVES), 0, st, dXq, lt, dout, B,
                     (int)ms[0]->d, accumulate, a);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

// the likelihoods of ng groups (k <= 32 each, one chain) for the B proposals whose partial sums their cross-kernel and
// triangular GEMM launches have just written; `aa` finishes the stretch move
int launch_loglik_groups(gpemu_model *const *ms, int ng, int64_t B, const double *dXq, double *dout, int accumulate,
                         hipStream_t st, const AcceptArgs *aa) {
  {
    const int rc = launch_loglik_tasks(ms, ng, B, dXq, dout, accumulate, st, aa);      // observable blocks on different waves
    if (rc != GPEMU_ERR_UNSUPPORTED) return rc;
  }
  LoglikGroups lg;
  lg.ng = ng;
  for (int g = 0; g < ng; ++g) {
    const gpemu_model *m = ms[g];
    const Workspace &w = m->ws;
    lg.g[g] = LoglikGroup{m->lo, m->hi, w.mean_part, w.vsq_part, m->kdiag, m->G, m->g0, m->scal, w.mean, w.var, w.Bcap,
                          (int)m->k, w.cur_nchunk, w.cur_nrb, (int)m->nblk};
  }
  const AcceptArgs a = aa ? *aa : AcceptArgs();
  const int wpw = ng <= 2 ? 2 : 1;
  hipLaunchKernelGGL(loglik_groups_kernel, dim3((unsigned)((B + wpw - 1) / wpw)), dim3(256), 0, st, dXq, lg, dout, B,
                     (int)ms[0]->d, accumulate, a);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

int launch_loglik_lowrank(gpemu_model *m, int64_t B, const double *dXq, double *dout, int accumulate,
                          hipStream_t st, const AcceptArgs *aa) {
  if (m->nblk > 1) {
    gpemu_model *one[1] = {m};
    const int rc = launch_loglik_tasks(one, 1, B, dXq, dout, accumulate, st, aa);      // observable blocks on different waves
    if (rc != GPEMU_ERR_UNSUPPORTED) return rc;
  }
  const Workspace &w = m->ws;
  const int k = (int)m->k;
  AcceptArgs a = aa ? *aa : AcceptArgs();
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
#define GP_LAUNCH_LL(KM)                                                                            \
  hipLaunchKernelGGL(loglik_lowrank_kernel<KM>, grid, block, 0, st, dXq, m->lo, m->hi, w.mean_part,  \
                     w.vsq_part, m->kdiag, m->G, m->g0, m->scal, dout, w.mean, w.var, B, w.Bcap,     \
                     (int)m->d, k, w.cur_nchunk, w.cur_nrb, (int)m->nblk, accumulate, a)
  if (k <= 4) {
    GP_LAUNCH_LL(4);
  } else if (k <= 8) {
    GP_LAUNCH_LL(8);
  } else if (k <= 12) {
    GP_LAUNCH_LL(12);
  } else if (k <= 16) {
    GP_LAUNCH_LL(16);
  } else if (k <= 20) {
    GP_LAUNCH_LL(20);
  } else if (k <= 24) {
    GP_LAUNCH_LL(24);
  } else if (k <= 28) {
    GP_LAUNCH_LL(28);
  } else if (k <= 32) {
    GP_LAUNCH_LL(32);
  } else {
    size_t shm = sizeof(double) * 4 * (size_t)k * (k + 1);
    if (shm > 64 * 1024)
      GP_HIP(hipFuncSetAttribute((const void *)loglik_lowrank_lds_kernel,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(loglik_lowrank_lds_kernel, grid, block, shm, st, dXq, m->lo, m->hi, w.mean_part,
                       w.vsq_part, m->kdiag, m->G, m->g0, m->scal, dout, w.mean, w.var, B, w.Bcap,
                       (int)m->d, k, w.cur_nchunk, w.cur_nrb, (int)m->nblk, accumulate, a);
  }
#undef GP_LAUNCH_LL
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
