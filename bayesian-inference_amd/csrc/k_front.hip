// Fused stretch-move run: TWO launches per half-step instead of four plus a collective.
//
// Replaces, for up to 8 emulation groups of up to 64 PCs each (the shipped analysis has three, with 5 / 11 / 25 PCs:
// ref: config/jet_substructure.yaml:243-278), the per-half-step chain
//     kstar (+proposal) -> triangular GEMM -> likelihood -> [all-gather] -> accept
// of k_sampler.hip (ref: mcmc.py:77-107 -> emcee RedBlueMove / StretchMove, the pool.map over walkers) by
//     front kernel  ->  triangular GEMM (one per group)
// where ONE front launch does, for the half-step BEFORE the one it opens:
//     (A) the low-rank log-likelihood of this rank's share of that half's proposals (a few workgroups, first);
//     (B) the exchange: each new log-probability is stored -- one 8-byte system-scope store per value and rank --
//         straight into every rank's gather buffer (peer memory over xGMI; the own buffer on one GPU).  A slot holds a
//         NaN with a reserved payload until its value arrives, so the value is its own flag: no fence, no counter;
//     (C) the accept / reject, recomputed by whoever needs a walker's position (the workgroups of the next half's
//         cross-kernel evaluation, for the two walkers of each of their proposals) from the previous half's draws;
//         a few extra workgroups write the accepted ensemble, the counters and the chain row into the OTHER half of
//         the double-buffered state;
// and then, for the half-step it opens, the stretch proposal and the cross-kernel K_*^T with its partial means
// (exactly kstar_kernel's arithmetic).  Every rank draws the same randomness (Philox, k_sampler.hip) and applies the
// same decisions, so the chain equals the single-GPU chain bit for bit.
#include <algorithm>

#include "internal.h"
#include "kstar_host.h"
#include "loglik_dev.h"
#include "predict_dev.h"
#include "sampler_internal.h"

namespace gpemu {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int FRONT_MAX_GROUPS = 8;

struct FrontGroup {
  // cross-kernel of the half being opened (matrix-core operands, kstar_host.h; Xs / inv_ls: Matern-0.5 only)
  const double *Xa, *alf, *qsc, *qof, *etab, *constv, *Xs, *inv_ls;
  double *KS, *mean_part_next;
  // likelihood of the half before
  const double *mean_part_prev, *vsq_part, *kdiag, *G, *g0, *scal;
  int64_t N, Npad, Bcap;
  int has_const, kind, k;
  int nchunk, wg0;                     // this group's cross-kernel workgroups: [wg0, wg0 + ncolblk * nchunk * k)
  const int *perm;                     // [nchunk * k][ncolblk] workgroup -> (chunk + nchunk * PC) of its column block, or null
  int nchunk_prev, nrb_prev, nblk;
};

struct FrontArgs {
  FrontGroup grp[FRONT_MAX_GROUPS];
  int ngroups;
  int lds_k;                           // largest k > 32 among the groups (0: none): that likelihood's matrix is in LDS
  int d, W;
  int ncolblk;                         // cross-kernel workgroups per group: (column block, row chunk, PC)
  int nkstar, nks;                     // all groups' / nks = max(nkstar, likelihood workgroups): the state workgroups follow
  // the half being opened: this rank's share of its proposals (pointers already offset to the share)
  int have_next, next_cnt;
  const int *idx_next, *partner_next;
  const double *zz_next;
  double *Xq_next;
  // the half before: likelihood of this rank's share, exchange, accept
  int have_prev, hp, prev_n, prev_lo, prev_cnt, n_llwg;
  const int *inds_prev, *pos_prev, *partner_prev;   // [W], [W], [prev_n]
  const double *zz_prev, *fac_prev, *logu_prev;     // [prev_n]
  const double *Xq_prev, *lo, *hi;
  const double *gath;                  // this rank's gather slot of that half: [prev_n]
  double *const *peers;                // every rank's gather buffer
  int world;
  int64_t slot_off;                    // offset of that half's slot inside a gather buffer
  // ensemble state (double buffered) and bookkeeping
  const double *Xcur, *lpcur;
  double *Xnext, *lpnext;
  long long *naccept;
  int *flags;                          // [0] NaN log-probabilities, [1] exchange time-outs
  double *chain_row, *lp_row;          // row of the step the accept completes, or null
  double *reset;                       // gather entries handed back to "not arrived"
  int reset_lo, reset_cnt;
  int slow_polls;                      // long naps before an exchange is declared lost
};

// Polls before an exchange is declared lost: GATHER_FAST_POLLS short naps (the normal case: the value is at most a
// half-step away, ~0.5 ms covers it), then long naps of ~3.5 us each, `slow_polls` of them (host: peer_timeout_polls(),
// 5 s unless GPEMU_PEER_TIMEOUT_MS says otherwise).  The ranks enter every run together (the host side puts a barrier
// in front of gpemu_sampler_run_peer), so what the wait has to absorb is the jitter of the ranks' launch streams, not a
// host-side hold-up between runs.  A rank that is really gone ends the wait, so the grid always drains; the value
// returned then is GATHER_LOST -- a NaN payload of its own, so that a lost exchange is not mistaken for a NaN
// log-probability of the model.
constexpr int GATHER_FAST_POLLS = 1 << 12;
constexpr unsigned long long GATHER_LOST = 0x7FF8DEADBEEF0002ull;

__device__ __forceinline__ unsigned long long gather_wait_bits(const double *entry, int *flags, int slow_polls) {
  const unsigned long long *p = reinterpret_cast<const unsigned long long *>(entry);
  unsigned long long bits;
  int spins = 0;
  for (;;) {
    bits = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (bits != GATHER_EMPTY) break;
    if (++spins > GATHER_FAST_POLLS + slow_polls) {          // every wave reaches this exit
      atomicAdd(flags + 1, 1);
      bits = GATHER_LOST;
      break;
    }
    if (spins <= GATHER_FAST_POLLS) {
      __builtin_amdgcn_s_sleep(4);
    } else {
      // once an exchange has been declared lost (by any wave, in this or an earlier launch of the run) nobody waits long
      if ((spins & 255) == 0 && __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        spins = GATHER_FAST_POLLS + slow_polls;
      __builtin_amdgcn_s_sleep(127);
    }
  }
  return bits;
}

// One-off check of the exchange path before a chain depends on it: every rank stores a token (1000 + its rank) into
// entry `rank` of EVERY rank's buffer through the mapped pointers, then waits (bounded) until the tokens of all ranks
// have arrived in its own buffer, and restores the entries.  *result = number of ranks whose token did not arrive.
__global__ void peer_selftest_kernel(double *const *peers, double *mine, int world, int rank, int *result) {
  const int t = threadIdx.x;
  int missing = 0;
  for (int q = t; q < world; q += blockDim.x) {
    const unsigned long long token = (unsigned long long)__double_as_longlong(1000.0 + rank);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(peers[q] + rank), token, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  for (int q = t; q < world; q += blockDim.x) {
    const unsigned long long want = (unsigned long long)__double_as_longlong(1000.0 + q);
    unsigned long long *p = reinterpret_cast<unsigned long long *>(mine + q);
    bool ok = false;
    for (int spins = 0; spins < (1 << 21); ++spins) {        // ~3 s at most: the grid always drains
      if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == want) { ok = true; break; }
      __builtin_amdgcn_s_sleep(32);
    }
    if (!ok) ++missing;
    __hip_atomic_store(p, GATHER_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (missing) atomicAdd(result, missing);
}

// Position (and log-probability) of walker x AFTER the previous half's accept / reject (emcee moves/red_blue.py):
// unchanged unless x proposed in that half and its proposal was accepted.
__device__ __forceinline__ void state_after_prev(const FrontArgs &fa, int x, double (&px)[DPAD], double &lp, bool &acc,
                                                 double &nlp, bool &in_prev, bool *lost = nullptr) {
  in_prev = fa.have_prev && fa.inds_prev[x] == fa.hp;
  double z = 1.0;
  int pj = x;
  acc = false;
  nlp = 0.0;
  const double oldlp = fa.lpcur[x];
  if (in_prev) {
    const int i = fa.pos_prev[x];
    pj = fa.partner_prev[i];
    z = fa.zz_prev[i];
    const double fc = fa.fac_prev[i], lu = fa.logu_prev[i];
    const unsigned long long bits = gather_wait_bits(fa.gath + i, fa.flags, fa.slow_polls);
    if (lost) *lost = bits == GATHER_LOST;
    nlp = __longlong_as_double((long long)bits);
    acc = (fc + nlp - oldlp) > lu;
  }
#pragma unroll
  for (int dd = 0; dd < DPAD; ++dd) {
    const double sw = fa.Xcur[(int64_t)x * DPAD + dd];
    double v = sw;
    if (acc && dd < fa.d) {
      const double cj = fa.Xcur[(int64_t)pj * DPAD + dd];
      v = cj - (cj - sw) * z;                 // emcee moves/stretch.py get_proposal
    }
    px[dd] = v;
  }
  lp = acc ? nlp : oldlp;
}

// one group's low-rank log-likelihood of proposal b (a wave per proposal), k <= 32: the arithmetic of
// loglik_lowrank_kernel<KMAX> (whose value does not depend on KMAX: one instantiation serves 17 <= k <= 32 here)
template <int KMAX>
__device__ __forceinline__ double front_loglik(const FrontGroup &gr, bool inside, int64_t b, int lane) {
  constexpr bool PRE = KMAX <= 16;
  double gpre[PRE ? KMAX : 1];
  if (PRE) {
#pragma unroll
    for (int q = 0; q < KMAX; ++q) gpre[q] = (q < gr.k && lane < gr.k) ? gr.G[q * gr.k + lane] : 0.0;
  }
  const double gl_pre = (lane < gr.k) ? gr.g0[lane] : 0.0;
  const double sc0_pre = gr.scal[0], sc1_pre = gr.scal[1];
  double mu, sd;
  walker_mean_sd<(KMAX <= 16 ? 16 : 32)>(gr.mean_part_prev, gr.vsq_part, gr.kdiag, nullptr, nullptr, b, gr.Bcap, gr.k,
                                         gr.nchunk_prev, gr.nrb_prev, lane, mu, sd);
  return walker_loglik_lowrank<KMAX, PRE>(inside, mu, sd, gpre, gl_pre, sc0_pre, sc1_pre, gr.G, gr.g0, gr.scal, gr.k,
                                          gr.nblk, lane);
}

// 32 < k <= 64: the arithmetic of loglik_lowrank_lds_kernel, the wave's k x (k + 1) matrix in `M`
__device__ __forceinline__ double front_loglik_lds(const FrontGroup &gr, bool inside, int64_t b, int lane, double *M) {
  double mu, sd;
  if (gr.k <= 32)
    walker_mean_sd<32>(gr.mean_part_prev, gr.vsq_part, gr.kdiag, nullptr, nullptr, b, gr.Bcap, gr.k, gr.nchunk_prev,
                       gr.nrb_prev, lane, mu, sd);
  else
    walker_mean_sd<64>(gr.mean_part_prev, gr.vsq_part, gr.kdiag, nullptr, nullptr, b, gr.Bcap, gr.k, gr.nchunk_prev,
                       gr.nrb_prev, lane, mu, sd);
  return walker_loglik_lowrank_lds(inside, mu, sd, gr.G, gr.g0, gr.scal, gr.k, gr.nblk, lane, M, gr.k + 1);
}

// the cross-kernel rows of one workgroup (kstar_kernel's arithmetic: predict_dev.h), base kernel chosen at run time
template <int JTW>
__device__ __forceinline__ double front_kstar_block(const FrontGroup &gk, const KstarFrags<2, JTW> &fr, const double *s_q,
                                                    const double *s_tab, double *s_red, int p, int chunk, int64_t b0, int d,
                                                    int lane, int wave) {
  constexpr int JT = 2 * JTW;                         // NBW = 2: two wave rows x JTW j-tiles
  const double c = gk.has_const ? gk.constv[p] : 0.0;
  const double *qsc = gk.qsc + p * 8, *qof = gk.qof + p * 8;
  double *ks = gk.KS + (int64_t)p * gk.Npad * gk.Bcap + b0;
  const KstarDirect none{nullptr, nullptr};
  switch (gk.kind) {
    case 0: return kstar_mfma_block<0, 2, JTW, 2, KSTAR_TB>(s_q, s_tab, s_red, fr, qsc, qof, c, d, (int64_t)chunk * JT, gk.N, ks, gk.Bcap, none, lane, wave);
    case 1: return kstar_mfma_block<1, 2, JTW, 2, KSTAR_TB>(s_q, s_tab, s_red, fr, qsc, qof, c, d, (int64_t)chunk * JT, gk.N, ks, gk.Bcap,
                                                           KstarDirect{gk.Xs + (int64_t)p * gk.Npad * DPAD, gk.inv_ls + p * DPAD}, lane, wave);
    case 2: return kstar_mfma_block<2, 2, JTW, 2, KSTAR_TB>(s_q, s_tab, s_red, fr, qsc, qof, c, d, (int64_t)chunk * JT, gk.N, ks, gk.Bcap, none, lane, wave);
    default: return kstar_mfma_block<3, 2, JTW, 2, KSTAR_TB>(s_q, s_tab, s_red, fr, qsc, qof, c, d, (int64_t)chunk * JT, gk.N, ks, gk.Bcap, none, lane, wave);
  }
}

// JTW: j-tiles per wave of the cross-kernel workgroups: 1 = 32 training rows per workgroup, 2 = 64.  KBIG: a group has
// 17 ... 32 PCs -- its likelihood in registers like the smaller ones (walker_loglik_lowrank<32>: 256 VGPRs; the
// instantiation without it keeps the 168 of the C3-sized run)
// The KBIG instantiation spills 9 VGPRs (40 B of scratch per lane) at the three-workgroups-per-CU budget of 168 registers; at two
// per CU (-DGPEMU_FRONT_KBIG_WPE=2) it takes 190 and spills nothing.  Three stays: measured on the shipped three-group
// shape (tools/ab_front_spill.sh, profiles/r05_front_spill.txt) the spill costs nothing that shows, and two per CU is a
// third less room where several ranks share a device (the two-rank rehearsal of the shipped shape no longer fits: 660
// workgroups against 512).
#ifndef GPEMU_FRONT_KBIG_WPE
#define GPEMU_FRONT_KBIG_WPE 3
#endif
template <int JTW, bool KBIG>
__global__ __launch_bounds__(256, KBIG ? GPEMU_FRONT_KBIG_WPE : 3) void front_kernel(FrontArgs fa) {
  __shared__ double s_tab[1 << KSTAR_TB];
  __shared__ __attribute__((aligned(16))) double s_q[64 * DPAD];
  __shared__ double s_eff[2][64][DPAD];
  __shared__ double red[4 * 64];
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];   // [4 waves][lds_k (lds_k + 1)]: likelihood, k > 32
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.x;

  if (g >= fa.nks) {
    // ---- state workgroups: the ensemble after the previous half, counters, chain row, gather hand-back ----
    const int x = (g - fa.nks) * 256 + threadIdx.x;
    if (x < fa.W) {
      double px[DPAD], lp, nlp;
      bool acc, in_prev, lost = false;
      state_after_prev(fa, x, px, lp, acc, nlp, in_prev, &lost);
#pragma unroll
      for (int dd = 0; dd < DPAD; ++dd) fa.Xnext[(int64_t)x * DPAD + dd] = px[dd];
      fa.lpnext[x] = lp;
      if (in_prev && nlp != nlp && !lost) atomicAdd(fa.flags, 1);   // emcee raises on a NaN log-probability
      if (acc) fa.naccept[x] += 1;
      if (fa.chain_row) {
        for (int dd = 0; dd < fa.d; ++dd) fa.chain_row[(int64_t)x * fa.d + dd] = px[dd];
        fa.lp_row[x] = lp;
      }
    }
    if (x < fa.reset_cnt)
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(fa.reset + fa.reset_lo + x), GATHER_EMPTY,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }

  // ---- cross-kernel workgroups: (group, PC, row chunk, column block) ----
  int gi = 0;
#pragma unroll
  for (int t = 1; t < FRONT_MAX_GROUPS; ++t)
    if (t < fa.ngroups && g >= fa.grp[t].wg0) gi = t;
  const FrontGroup &gk = fa.grp[gi];
  const int gl = g - gk.wg0;
  const int cb = gl % fa.ncolblk;
  // which (row chunk, PC) of column block cb: in index order, or -- XCD-aware -- the one whose K_*^T rows the triangular
  // GEMM will read from THIS workgroup's XCD (workgroup i runs on XCD i % 8; front_perm_for).  Needed only after the
  // waits below, which hide the load -- except with the 32-PC likelihood (KBIG), which has no register to carry it
  // across (the instantiation spilled 9 VGPRs at the three-workgroups-per-CU budget): there it is asked for afterwards.
  int rest = 0;
  if (!KBIG) rest = (gk.perm && g < fa.nkstar) ? gk.perm[gl] : gl / fa.ncolblk;
  const bool does_kstar = fa.have_next && g < fa.nkstar;
  const bool does_ll = fa.have_prev && g < fa.n_llwg;
  if (does_kstar && threadIdx.x < (1 << KSTAR_TB)) s_tab[threadIdx.x] = gk.etab[threadIdx.x];

  // (A) + (B): likelihood of the previous half's proposals of this rank (first workgroups), stored to every rank
  if (does_ll) {
    const int i = g * 4 + wave;
    if (i < fa.prev_cnt) {
      bool in = true;
      if (lane < fa.d) in = (fa.Xq_prev[(int64_t)i * DPAD + lane] > fa.lo[lane]) && (fa.Xq_prev[(int64_t)i * DPAD + lane] < fa.hi[lane]);
      const bool inside = __all(in);
      // the groups' log-likelihoods, added in the order the single-GPU run accumulates them (lp_g + sum so far)
      double total = 0.0;
      for (int t = 0; t < fa.ngroups; ++t) {
        const FrontGroup &gr = fa.grp[t];
        double lp;
        if (gr.k <= 4) lp = front_loglik<4>(gr, inside, i, lane);
        else if (gr.k <= 8) lp = front_loglik<8>(gr, inside, i, lane);
        else if (gr.k <= 12) lp = front_loglik<12>(gr, inside, i, lane);
        else if (gr.k <= 16) lp = front_loglik<16>(gr, inside, i, lane);
        else if (KBIG && gr.k <= 32) lp = front_loglik<32>(gr, inside, i, lane);
        else lp = front_loglik_lds(gr, inside, i, lane, dyn_lds + (size_t)wave * fa.lds_k * (fa.lds_k + 1));
        total = (t == 0) ? lp : lp + total;
      }
      if (lane < fa.world)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(fa.peers[lane] + fa.slot_off + fa.prev_lo + i),
                           (unsigned long long)__double_as_longlong(total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (!does_kstar) return;
  if (KBIG) rest = gk.perm ? gk.perm[gl] : gl / fa.ncolblk;
  const int chunk = rest % gk.nchunk;
  const int p = rest / gk.nchunk;
  const int64_t b = (int64_t)cb * 64 + lane;
  // the workgroup's training fragments.  Requested here, not at the top: held across the waits below they cost 12-24
  // VGPRs (174 / 186 instead of 162) and with them the third resident workgroup per CU -- measured slower at 2 and 4
  // ranks (0.175 / 0.115 vs 0.170 / 0.109 ms per step), equal at 8
  KstarFrags<2, JTW> fr;
  const int64_t njt = gk.Npad / 16;
  kstar_load_frags<2, JTW, 2>(fr, gk.Xa + (int64_t)p * njt * 2 * 64, gk.alf + (int64_t)p * njt * 16, (int64_t)chunk * (2 * JTW),
                              lane, wave);

  // (C) the two walkers of this column's proposal, as they stand after the previous half; wave 0: the proposing
  // walker, wave 1: its partner of the complementary set
  const bool live = b < fa.next_cnt;
  if (wave < 2) {
    double px[DPAD];
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) px[dd] = 0.0;
    if (live) {
      const int x = (wave == 0) ? fa.idx_next[b] : fa.partner_next[b];
      double lp, nlp;
      bool acc, in_prev;
      state_after_prev(fa, x, px, lp, acc, nlp, in_prev);
    }
#pragma unroll
    for (int dd = 0; dd < DPAD; ++dd) s_eff[wave][lane][dd] = px[dd];
  }
  __syncthreads();

  // stretch proposal q = c - (c - s) z  (emcee moves/stretch.py), two components per thread, stored once per column
  {
    const double z = live ? fa.zz_next[b] : 1.0;
    const int c0 = wave, c1 = wave + 4;
    double q0 = 0.0, q1 = 0.0;
    if (live && c0 < fa.d) {
      const double cj = s_eff[1][lane][c0], sw = s_eff[0][lane][c0];
      q0 = cj - (cj - sw) * z;
    }
    if (live && c1 < fa.d) {
      const double cj = s_eff[1][lane][c1], sw = s_eff[0][lane][c1];
      q1 = cj - (cj - sw) * z;
    }
    if (gi == 0 && chunk == 0 && p == 0) {
      fa.Xq_next[b * DPAD + c0] = q0;
      fa.Xq_next[b * DPAD + c1] = q1;
    }
    s_q[lane * DPAD + c0] = q0;
    s_q[lane * DPAD + c1] = q1;
  }
  __syncthreads();
  const double sum = front_kstar_block<JTW>(gk, fr, s_q, s_tab, red, p, chunk, (int64_t)cb * 64, fa.d, lane, wave);
  if (wave == 0) gk.mean_part_next[(b * gk.k + p) * gk.nchunk + chunk] = sum;
}

__global__ void gather_fill_kernel(double *p, int64_t n, unsigned long long bits) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) reinterpret_cast<unsigned long long *>(p)[i] = bits;
}

// ---- host side --------------------------------------------------------------------------------------------
static int ensure_gather(gpemu_sampler *s) {
  if (s->gather) return GPEMU_OK;
  const size_t bytes = sizeof(double) * GATHER_SLOTS * (size_t)s->ns[0];
  // uncached device memory: the values are written by other workgroups / other GPUs and polled here
  hipError_t e = hipExtMallocWithFlags((void **)&s->gather, bytes, hipDeviceMallocUncached);
  s->gather_uncached = (e == hipSuccess);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    GP_HIP(hipMalloc((void **)&s->gather, bytes));
  }
  const int64_t n = GATHER_SLOTS * s->ns[0];
  hipLaunchKernelGGL(gather_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->gather, n,
                     GATHER_EMPTY);
  GP_HIP(hipStreamSynchronize(s->stream));
  return GPEMU_OK;
}

static int set_local_peer(gpemu_sampler *s) {
  if (s->peers && s->peer_world == 1) return GPEMU_OK;
  GP_HIP(hipStreamSynchronize(s->stream));
  for (void *q : s->peer_opened) (void)hipIpcCloseMemHandle(q);
  s->peer_opened.clear();
  (void)hipFree(s->peers);
  s->peers = nullptr;
  GP_HIP(hipMalloc((void **)&s->peers, sizeof(double *)));
  GP_HIP(hipMemcpy(s->peers, &s->gather, sizeof(double *), hipMemcpyHostToDevice));
  s->peer_world = 1;
  s->peer_rank = 0;
  return GPEMU_OK;
}

void front_release(gpemu_sampler *s) {
  for (const gpemu_sampler::FrontPerm &e : s->front_perms) (void)hipFree(e.dperm);
  s->front_perms.clear();
  for (void *q : s->peer_opened) (void)hipIpcCloseMemHandle(q);
  s->peer_opened.clear();
  (void)hipFree(s->peers);
  (void)hipFree(s->gather);
  s->peers = nullptr;
  s->gather = nullptr;
  s->peer_world = 0;
}

// Long naps (~3.5 us each: s_sleep 127 = 8128 cycles) a waiter spends before it declares an exchange lost.
static int peer_timeout_polls() {        // read per launch: a getenv, nothing next to a kernel launch
  double ms = 5000.0;
  if (const char *e = getenv("GPEMU_PEER_TIMEOUT_MS")) ms = atof(e);
  ms = std::min(std::max(ms, 1.0), 60000.0);
  return (int)(ms * 1000.0 / 3.5);
}

// Who waits for whom in a front launch: the state and the cross-kernel workgroups poll gather entries, which the
// LIKELIHOOD workgroups of every rank store -- the first n_llwg workgroups of their launch, which wait for nothing of
// that launch (their inputs are the previous launches' partial sums, ordered by the stream).  A 1-D grid is dispatched in
// index order (round-robin over the XCDs, each XCD taking its share in order), so on every XCD the likelihood
// workgroups are placed before any poller of the same launch can hold a slot: the pollers never keep them out, and a
// launch may be (and at C3 with 2 ranks is: 1 285 workgroups against 768 resident ones) larger than what is resident
// at once.  One GPU per rank therefore needs NO residency rule.  Only when several ranks SHARE a device (the one-GPU
// rehearsals of the tests and of bench.py) can one process's resident pollers keep another process's likelihood
// workgroups off the chip; then every rank's launch must fit beside the others': grid x ranks on the device <= resident
// workgroups, taken from the runtime's occupancy figure for the kernel actually launched (3 per CU at 168 VGPRs, not the 8
// an LDS-only count gives).  The ranks learn the share from gpemu_sampler_peer_share; a launch that does not fit makes
// the import fail with GPEMU_ERR_UNSUPPORTED and the ranks fall back to the collective transports together.
static int front_lds_k(const gpemu_sampler *s) {
  int lds_k = 0;
  for (const gpemu_model *m : s->groups)
    if (m->k > 32) lds_k = std::max(lds_k, (int)m->k);
  return lds_k;
}
static size_t front_dyn_lds(const gpemu_sampler *s) {
  const int lds_k = front_lds_k(s);
  return sizeof(double) * 4 * (size_t)lds_k * (lds_k + 1);
}
static bool front_kbig(const gpemu_sampler *s) {
  for (const gpemu_model *m : s->groups)
    if (m->k > 16 && m->k <= 32) return true;
  return false;
}
static const void *front_kernel_ptr(bool small, bool kbig) {
  if (small) return kbig ? (const void *)front_kernel<1, true> : (const void *)front_kernel<1, false>;
  return kbig ? (const void *)front_kernel<2, true> : (const void *)front_kernel<2, false>;
}
static int front_set_lds_limit() {
  static bool attr_set = false;
  if (!attr_set) {
    for (int v = 0; v < 4; ++v)
      GP_HIP(hipFuncSetAttribute(front_kernel_ptr(v & 1, v & 2), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
    attr_set = true;
  }
  return GPEMU_OK;
}
// workgroups of front_kernel resident at once on the device (runtime occupancy x CUs); 0 if the runtime cannot say
static int64_t front_capacity(const gpemu_sampler *s, bool small) {
  const size_t dyn = front_dyn_lds(s);
  if (dyn > 40 * 1024 && front_set_lds_limit() != GPEMU_OK) return 0;
  int per_cu = 0;
  const bool kbig = front_kbig(s);
  hipError_t e;
  if (small) e = kbig ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, front_kernel<1, true>, 256, dyn)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, front_kernel<1, false>, 256, dyn);
  else e = kbig ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, front_kernel<2, true>, 256, dyn)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, front_kernel<2, false>, 256, dyn);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return (int64_t)per_cu * s->groups[0]->num_cu;
}
// XCD-aware order of one group's cross-kernel workgroups for a share of `cnt` proposals (cached per sampler).  The L2s of
// the 8 XCDs are kept coherent by hardware, so a K_*^T line written on one XCD and read on another travels through the
// fabric and the store has to invalidate the reader's stale copy first (measured on the single-GPU step: 19.6 -> 13.6 us
// for the cross-kernel when its rows are written where they are read).  Workgroup wg0 + gl runs on XCD (wg0 + gl) % 8 and
// handles column block gl % ncolblk; its (row chunk, PC) is dealt so that the PC's rows are written on the XCD whose
// workers read them (trmm_xcd_of); tasks without a preference fill the remaining slots.  Any bijection is correct.
static const int *front_perm_for(gpemu_sampler *s, int g, int64_t cnt, int wg0, int ncolblk, int nchunk) {
  for (const gpemu_sampler::FrontPerm &e : s->front_perms)
    if (e.cnt == cnt && e.group == g && e.wg0 == wg0) return e.dperm;
  const gpemu_model *m = s->groups[g];
  const int k = (int)m->k, nrest = nchunk * k;
  std::vector<int> perm((size_t)nrest * ncolblk, -1);
  bool any_pref = false;
  for (int cb = 0; cb < ncolblk; ++cb) {
    std::vector<std::vector<int>> want(9);                 // tasks (chunk + nchunk * p) by preferred XCD; [8]: none
    for (int p = 0; p < k; ++p) {
      const int x = trmm_xcd_of(m, cnt, p, (int64_t)cb * 64);
      any_pref |= x >= 0;
      for (int c = nchunk - 1; c >= 0; --c) want[x >= 0 ? x : 8].push_back(c + nchunk * p);   // popped from the back: ascending
    }
    std::vector<int> open_slots;
    for (int r = 0; r < nrest; ++r) {
      const int x = (wg0 + r * ncolblk + cb) % 8;
      if (!want[x].empty()) { perm[(size_t)r * ncolblk + cb] = want[x].back(); want[x].pop_back(); }
      else open_slots.push_back(r);
    }
    size_t o = 0;
    for (int x = 8; x >= 0; --x)                            // the rest: tasks without a preference first
      while (!want[x].empty()) { perm[(size_t)open_slots[o++] * ncolblk + cb] = want[x].back(); want[x].pop_back(); }
  }
  int *d = nullptr;
  if (any_pref) {
    if (hipMalloc((void **)&d, sizeof(int) * perm.size()) != hipSuccess ||
        hipMemcpy(d, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(d);
      d = nullptr;                                          // no table: index order (correct, only slower)
    }
  }
  s->front_perms.push_back({cnt, g, wg0, d});
  return d;
}

// workgroups a front launch needs for `cnt` proposals of this rank
static int64_t front_grid(const gpemu_sampler *s, int64_t cnt) {
  const int rows_per_wg = cnt <= KSTAR_SMALL_MAX ? KSTAR_ROWS_SMALL : KSTAR_ROWS_BIG;
  const int64_t ncolblk = ((cnt <= 64) ? 64 : round_up(cnt, TILE)) / 64;
  int64_t nkstar = 0;
  for (const gpemu_model *m : s->groups) nkstar += ncolblk * (m->Npad / rows_per_wg) * m->k;
  const int64_t n_llwg = (s->ns[0] + 3) / 4;                  // at most: the whole half on one rank
  return std::max(nkstar, std::min<int64_t>(n_llwg, (cnt + 3) / 4 + 1)) + (s->W + 255) / 256 + 1;
}

bool front_eligible(const gpemu_sampler *s) {
  static const bool off = getenv("GPEMU_NO_FUSED") != nullptr;
  if (off || s->groups.empty() || (int)s->groups.size() > FRONT_MAX_GROUPS || s->nchains != 1) return false;
  for (const gpemu_model *m : s->groups)
    if (m->k > 64 || m->ksteps != 2 || m->device != s->groups[0]->device) return false;   // ksteps 3: d = 8 parameters
  return true;
}

// can `cnt` proposals per rank and half run fused?  Always with one rank per device; with `device_share` ranks on one
// device only if all their launches can be resident together (see above)
static bool front_fits(const gpemu_sampler *s, int64_t cnt) {
  if (s->device_share <= 1) return true;
  cnt = std::max<int64_t>(cnt, 1);
  return front_grid(s, cnt) * s->device_share <= front_capacity(s, cnt <= KSTAR_SMALL_MAX);
}

bool front_eligible_for(const gpemu_sampler *s, int world) {
  return front_eligible(s) && front_fits(s, (s->ns[0] + world - 1) / std::max(world, 1));
}

struct Pending {                    // the half whose likelihood / accept is still to be done
  bool have = false;
  int h = 0;
  size_t slot = 0;                  // randomness ring slot of its step
  int64_t lo = 0, cnt = 0;
  int nchunk[FRONT_MAX_GROUPS] = {0}, nrb[FRONT_MAX_GROUPS] = {0};   // per group: chunks / row blocks of its partial sums
  int parity = 0;                   // which q2 / mean_part buffer its proposals and partial means are in
};

static void share_of(const gpemu_sampler *s, int h, int world, int rank, int64_t &lo, int64_t &cnt) {
  const int64_t share = (s->ns[h] + world - 1) / world;        // == sampler.shard_bounds()
  lo = std::min<int64_t>((int64_t)rank * share, s->ns[h]);
  cnt = std::min<int64_t>(share, s->ns[h] - lo);
}

static int launch_front(gpemu_sampler *s, const Pending &pv, bool have_next, int h, int64_t lo, int64_t cnt,
                        int world, bool emulate, int store_row, Pending &out) {
  gpemu_model *m0 = s->groups[0];
  hipStream_t st = s->stream;
  const int64_t W = s->W;
  const int ng = (int)s->groups.size();
  FrontArgs fa;
  memset(&fa, 0, sizeof(fa));
  const int parity = (int)(s->front_count & 1);
  fa.ngroups = ng;
  fa.lds_k = front_lds_k(s);
  fa.d = (int)s->d; fa.W = (int)W;
  const bool small = cnt <= KSTAR_SMALL_MAX;
  const int rows_per_wg = small ? KSTAR_ROWS_SMALL : KSTAR_ROWS_BIG;
  const int64_t ncols = (cnt <= 64) ? 64 : round_up(cnt, TILE);
  fa.ncolblk = (int)(ncols / 64);
  int wg = 0;
  for (int g = 0; g < ng; ++g) {
    gpemu_model *m = s->groups[g];
    Workspace &w = m->ws;
    FrontGroup &fg = fa.grp[g];
    fg.Xa = m->Xa; fg.alf = m->alf; fg.qsc = m->qsc; fg.qof = m->qof; fg.etab = m->etab; fg.constv = m->constv;
    fg.Xs = m->Xs; fg.inv_ls = m->inv_ls;
    fg.KS = w.KS;
    fg.mean_part_next = parity ? w.mean_part2 : w.mean_part;
    fg.N = m->N; fg.Npad = m->Npad; fg.Bcap = w.Bcap;
    fg.has_const = m->has_const; fg.k = (int)m->k;
    fg.kind = kstar_kind(m);
    fg.nchunk = (int)(m->Npad / rows_per_wg);
    fg.wg0 = wg;
    fg.perm = have_next ? front_perm_for(s, g, cnt, wg, fa.ncolblk, fg.nchunk) : nullptr;
    if (have_next) wg += fa.ncolblk * fg.nchunk * fg.k;
    if (pv.have) {
      fg.mean_part_prev = pv.parity ? w.mean_part2 : w.mean_part;
      fg.vsq_part = w.vsq_part; fg.kdiag = m->kdiag; fg.G = m->G; fg.g0 = m->g0; fg.scal = m->scal;
      fg.nchunk_prev = pv.nchunk[g]; fg.nrb_prev = pv.nrb[g]; fg.nblk = (int)m->nblk;
    }
  }
  fa.nkstar = wg;
  fa.have_next = have_next ? 1 : 0;
  if (have_next) {
    const size_t o2 = s->step_counter % RNG_RING * 2 * W;
    fa.next_cnt = (int)cnt;
    fa.idx_next = s->idx + o2 + h * W + lo;
    fa.partner_next = s->rint + o2 + h * W + lo;
    fa.zz_next = s->zz + o2 + h * W + lo;
    fa.Xq_next = s->q2 + (size_t)parity * s->qcap * DPAD;
  }
  fa.have_prev = pv.have ? 1 : 0;
  if (pv.have) {
    const size_t o2 = pv.slot * 2 * W;
    fa.hp = pv.h;
    fa.prev_n = (int)s->ns[pv.h];
    fa.prev_lo = (int)pv.lo; fa.prev_cnt = (int)pv.cnt;
    fa.n_llwg = (int)((pv.cnt + 3) / 4);
    fa.inds_prev = s->inds + pv.slot * W;
    fa.pos_prev = s->pos + pv.slot * W;
    fa.partner_prev = s->rint + o2 + pv.h * W;
    fa.zz_prev = s->zz + o2 + pv.h * W;
    fa.fac_prev = s->fac + o2 + pv.h * W;
    fa.logu_prev = s->logu + o2 + pv.h * W;
    fa.Xq_prev = s->q2 + (size_t)pv.parity * s->qcap * DPAD;
    fa.lo = m0->lo; fa.hi = m0->hi;
    // the previous half's values live in slot (front_count - 1) % GATHER_SLOTS
    const int slot = (int)((s->front_count + GATHER_SLOTS - 1) % GATHER_SLOTS);
    fa.slot_off = (int64_t)slot * s->ns[0];
    fa.gath = s->gather + fa.slot_off;
    fa.peers = s->peers;
    fa.world = emulate ? 1 : world;
  }
  fa.nks = std::max(fa.nkstar, fa.n_llwg);
  fa.Xcur = s->Xbuf + (size_t)s->cur * W * DPAD;
  fa.lpcur = s->lpbuf + (size_t)s->cur * W;
  fa.Xnext = s->Xbuf + (size_t)(s->cur ^ 1) * W * DPAD;
  fa.lpnext = s->lpbuf + (size_t)(s->cur ^ 1) * W;
  fa.naccept = s->naccept; fa.flags = s->flags;
  fa.slow_polls = peer_timeout_polls();
  if (store_row >= 0) {
    fa.chain_row = s->chain + (size_t)store_row * W * s->d;
    fa.lp_row = s->lpchain + (size_t)store_row * W;
  }
  // hand back the slot the launch after next will receive into: the whole slot, or -- when one GPU stands in for
  // rank 0 of a larger job -- only the share somebody writes (the rest stays -inf: rejected)
  {
    const int rslot = (int)((s->front_count + 1) % GATHER_SLOTS);
    fa.reset = s->gather + (int64_t)rslot * s->ns[0];
    fa.reset_lo = 0;
    fa.reset_cnt = (int)s->ns[0];
    if (emulate) {
      int64_t l0, c0, l1, c1;
      share_of(s, 0, world, 0, l0, c0);
      share_of(s, 1, world, 0, l1, c1);
      fa.reset_cnt = (int)std::max(c0, c1);
    }
  }
  const int nstate = (int)((std::max<int64_t>(W, fa.reset_cnt) + 255) / 256);
  const dim3 grid((unsigned)(fa.nks + nstate)), block(256);
  const size_t dyn = front_dyn_lds(s);
  if (dyn > 40 * 1024 && front_set_lds_limit() != GPEMU_OK) return GPEMU_ERR_HIP;
  const int pe0 = prof_mark(m0, st);
  const bool kbig = front_kbig(s);
  if (small && kbig) hipLaunchKernelGGL((front_kernel<1, true>), grid, block, dyn, st, fa);
  else if (small) hipLaunchKernelGGL((front_kernel<1, false>), grid, block, dyn, st, fa);
  else if (kbig) hipLaunchKernelGGL((front_kernel<2, true>), grid, block, dyn, st, fa);
  else hipLaunchKernelGGL((front_kernel<2, false>), grid, block, dyn, st, fa);
  GP_HIP(hipGetLastError());
  prof_pair(m0, 1, pe0, prof_mark(m0, st));
  // bookkeeping
  s->cur ^= 1;
  s->X = s->Xbuf + (size_t)s->cur * W * DPAD;
  s->logp = s->lpbuf + (size_t)s->cur * W;
  out = Pending();
  if (have_next) {
    out.have = true; out.h = h; out.slot = s->step_counter % RNG_RING;
    out.lo = lo; out.cnt = cnt; out.parity = parity;
    for (int g = 0; g < ng; ++g) {
      out.nchunk[g] = fa.grp[g].nchunk;
      s->groups[g]->ws.cur_nchunk = fa.grp[g].nchunk;
    }
  }
  s->front_count += 1;
  return GPEMU_OK;
}

int front_run(gpemu_sampler *s, int64_t steps, int store_chain, int world, int rank, bool emulate) {
  hipStream_t st = s->stream;
  for (gpemu_model *m : s->groups)
    if (!m->lik_ready) { set_error("gpemu_likelihood_setup has not been called"); return GPEMU_ERR_STATE; }
  int rc = ensure_gather(s);
  if (rc != GPEMU_OK) return rc;
  if (world == 1 || emulate) {
    rc = set_local_peer(s);
    if (rc != GPEMU_OK) return rc;
  } else if (!s->peers || s->peer_world != world || s->peer_rank != rank) {
    set_error("fused sharded run: the ranks' gather buffers have not been exchanged (gpemu_sampler_peer_import)");
    return GPEMU_ERR_STATE;
  }
  int64_t lo[2], cnt[2];
  for (int h = 0; h < 2; ++h) share_of(s, h, world, emulate ? 0 : rank, lo[h], cnt[h]);
  if (!front_fits(s, std::max(cnt[0], cnt[1]))) {
    set_error("the fused run's launches for %lld proposals per rank over %d group(s) do not fit beside those of the %d ranks "
              "sharing this device", (long long)std::max(cnt[0], cnt[1]), (int)s->groups.size(), s->device_share);
    return GPEMU_ERR_UNSUPPORTED;
  }
  for (gpemu_model *m : s->groups) {
    rc = ensure_workspace(m, std::max<int64_t>(std::max(cnt[0], cnt[1]), 1));
    if (rc != GPEMU_OK) return rc;
  }
  if (emulate) {
    // proposals nobody evaluates carry -inf (rejected); the share that IS evaluated starts as "not arrived"
    const int64_t n = GATHER_SLOTS * s->ns[0];
    hipLaunchKernelGGL(gather_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->gather, n,
                       0xFFF0000000000000ull);
    for (int sl = 0; sl < GATHER_SLOTS; ++sl)
      hipLaunchKernelGGL(gather_fill_kernel, dim3((unsigned)((std::max(cnt[0], cnt[1]) + 255) / 256)), dim3(256), 0, st,
                         s->gather + (int64_t)sl * s->ns[0], std::max(cnt[0], cnt[1]), GATHER_EMPTY);
  }
  if (store_chain) {
    rc = sampler_ensure_chain(s, s->chain_len + steps);
    if (rc != GPEMU_OK) return rc;
  }
  Pending pend;
  int64_t rec = s->chain_len;       // next chain row to be written
  for (int64_t it = 0; it < steps; ++it) {
    rc = sampler_launch_rng(s, st, steps - it);
    if (rc != GPEMU_OK) return rc;
    for (int h = 0; h < 2; ++h) {
      // the accept of half 1 of the previous step completes that step: its chain row is written now
      const int row = (pend.have && pend.h == 1 && store_chain) ? (int)(rec++) : -1;
      Pending next;
      if (cnt[h] > 0) {
        rc = launch_front(s, pend, true, h, lo[h], cnt[h], world, emulate, row, next);
        if (rc != GPEMU_OK) return rc;
        for (size_t g = 0; g < s->groups.size(); ++g) {
          rc = launch_trmm_vsq(s->groups[g], cnt[h], st);
          if (rc != GPEMU_OK) return rc;
          next.nrb[g] = s->groups[g]->ws.cur_nrb;
        }
      } else {
        // this rank has no proposal in this half (more ranks than proposals): likelihood / accept only
        rc = launch_front(s, pend, false, h, 0, 0, world, emulate, row, next);
        if (rc != GPEMU_OK) return rc;
        next.have = true; next.h = h; next.slot = s->step_counter % RNG_RING; next.lo = lo[h]; next.cnt = 0;
      }
      pend = next;
    }
    s->step_counter += 1;
  }
  if (pend.have) {                 // the last half's likelihood, exchange and accept
    const int row = (pend.h == 1 && store_chain) ? (int)(rec++) : -1;
    Pending none;
    rc = launch_front(s, pend, false, 0, 0, 0, world, emulate, row, none);
    if (rc != GPEMU_OK) return rc;
  }
  if (store_chain) s->chain_len += steps;
  s->iterations += steps;
  if (emulate) {                   // leave the buffer as a real run expects it
    const int64_t n = GATHER_SLOTS * s->ns[0];
    hipLaunchKernelGGL(gather_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->gather, n,
                       GATHER_EMPTY);
  }
  // a lost exchange first: its walkers carry the GATHER_LOST payload, which must not be reported as a NaN of the model,
  // and its flag must not outlive this run (every later wait would give up after the fast polls)
  int fl[2] = {0, 0};
  GP_HIP(hipMemcpyAsync(fl, s->flags, sizeof(fl), hipMemcpyDeviceToHost, st));
  GP_HIP(hipStreamSynchronize(st));
  if (fl[1]) {
    (void)hipMemsetAsync(s->flags, 0, sizeof(fl), st);
    (void)hipStreamSynchronize(st);
    set_error("fused run: %d log-probability exchanges timed out (a rank did not deliver its share)", fl[1]);
    return GPEMU_ERR_STATE;
  }
  return sampler_check_nan(s);
}

}  // namespace gpemu

using namespace gpemu;

extern "C" {

int gpemu_sampler_peer_export(gpemu_sampler *s, char *handle_out64) {
  GP_ARG(s && handle_out64, "null pointer");
  GP_HIP(hipSetDevice(s->device));
  if (!front_eligible(s)) {     // the caller (every rank alike) then stays on the collective transports
    set_error("the fused run needs at most %d emulation groups of at most 64 PCs and 7 parameters, one chain", FRONT_MAX_GROUPS);
    return GPEMU_ERR_UNSUPPORTED;
  }
  int rc = ensure_gather(s);
  if (rc != GPEMU_OK) return rc;
  hipIpcMemHandle_t h;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  GP_HIP(hipIpcGetMemHandle(&h, s->gather));
  memcpy(handle_out64, &h, 64);
  return GPEMU_OK;
}

int gpemu_sampler_peer_import(gpemu_sampler *s, int world, int rank, const char *handles) {
  GP_ARG(s && handles && world >= 1 && world <= 64 && rank >= 0 && rank < world, "world / rank / handles");
  GP_HIP(hipSetDevice(s->device));
  if (!front_eligible_for(s, world)) {
    set_error("the fused run cannot take this sampler at %d rank(s): at most %d groups of at most 64 PCs and 7 parameters, "
              "one chain, and -- with %d ranks on this device -- all their launches resident together", world,
              FRONT_MAX_GROUPS, s->device_share);
    return GPEMU_ERR_UNSUPPORTED;
  }
  int rc = ensure_gather(s);
  if (rc != GPEMU_OK) return rc;
  GP_HIP(hipStreamSynchronize(s->stream));
  for (void *q : s->peer_opened) (void)hipIpcCloseMemHandle(q);
  s->peer_opened.clear();
  (void)hipFree(s->peers);
  s->peers = nullptr;
  s->peer_world = 0;
  std::vector<double *> ptrs((size_t)world, nullptr);
  for (int r = 0; r < world; ++r) {
    if (r == rank) { ptrs[r] = s->gather; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * 64, 64);
    void *q = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      set_error("hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
      for (void *o : s->peer_opened) (void)hipIpcCloseMemHandle(o);
      s->peer_opened.clear();
      return GPEMU_ERR_HIP;
    }
    s->peer_opened.push_back(q);
    ptrs[r] = (double *)q;
  }
  GP_HIP(hipMalloc((void **)&s->peers, sizeof(double *) * world));
  GP_HIP(hipMemcpy(s->peers, ptrs.data(), sizeof(double *) * world, hipMemcpyHostToDevice));
  s->peer_world = world;
  s->peer_rank = rank;
  return GPEMU_OK;
}

int gpemu_sampler_peer_share(gpemu_sampler *s, int ranks_on_device) {
  GP_ARG(s && ranks_on_device >= 1 && ranks_on_device <= 64, "sampler / ranks_on_device");
  s->device_share = ranks_on_device;
  return GPEMU_OK;
}

int gpemu_sampler_peer_selftest(gpemu_sampler *s) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  if (s->peer_world < 1 || !s->peers) { set_error("gpemu_sampler_peer_import has not been called"); return GPEMU_ERR_STATE; }
  if ((int64_t)s->peer_world > GATHER_SLOTS * s->ns[0]) { set_error("peer self-test: more ranks than buffer entries"); return GPEMU_ERR_ARG; }
  int *dres = nullptr;
  GP_HIP(hipMalloc((void **)&dres, sizeof(int)));
  GP_HIP(hipMemsetAsync(dres, 0, sizeof(int), s->stream));
  hipLaunchKernelGGL(peer_selftest_kernel, dim3(1), dim3(64), 0, s->stream, s->peers, s->gather, s->peer_world, s->peer_rank, dres);
  int res = -1;
  hipError_t e = hipMemcpyAsync(&res, dres, sizeof(int), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  (void)hipFree(dres);
  if (e != hipSuccess) { set_error("peer self-test: %s", hipGetErrorString(e)); return GPEMU_ERR_HIP; }
  if (res != 0) {
    set_error("peer self-test: the tokens of %d rank(s) did not arrive through the mapped buffers", res);
    return GPEMU_ERR_STATE;
  }
  return GPEMU_OK;
}

int gpemu_sampler_run_peer(gpemu_sampler *s, int64_t steps, int store_chain) {
  GP_ARG(s && steps >= 0, "sampler / steps");
  GP_HIP(hipSetDevice(s->device));
  if (!front_eligible(s)) {
    set_error("the fused run needs at most %d emulation groups of at most 64 PCs, one chain", FRONT_MAX_GROUPS);
    return GPEMU_ERR_UNSUPPORTED;
  }
  if (s->peer_world < 1) { set_error("gpemu_sampler_peer_import has not been called"); return GPEMU_ERR_STATE; }
  return front_run(s, steps, store_chain, s->peer_world, s->peer_rank, false);
}

}  // extern "C"
