// Affine-invariant stretch-move ensemble sampler, resident on the device.
//
// Replaces  ref: mcmc.py:83-107, 187-204  (emcee 3.1.4 EnsembleSampler + StretchMove(a=2) inside
// RedBlueMove(nsplits=2, randomize_split=True), driven through a multiprocessing pool).  emcee is
// a third-party dependency that is not vendored in the reference (ref: pdm.lock:504-505); the move
// is restated from its published algorithm (Goodman & Weare 2010; emcee moves/red_blue.py,
// moves/stretch.py), see oracle/sampler_oracle.py.  Per step:
//     inds = shuffle(arange(W) % 2)
//     for split in (0, 1):  s = walkers with inds == split,  c = the others (CURRENT positions)
//         zz = ((a-1) u + 1)^2 / a ; rint = randint(Nc) ; q = c[rint] - (c[rint] - s) zz
//         accept iff (d-1) log zz + logp(q) - logp(s) > log(u')
// Randomness is either drawn on the device (Philox4x32-10, counter = (step, stream, index, draw),
// identical on every rank so no communication is needed for it) or supplied by the host in emcee's
// draw order (gpemu_sampler_step_host_rng) to replay a numpy RandomState stream.
#include "internal.h"
#include "sampler_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace gpemu {

// ---- Philox4x32-10 (Salmon et al., SC'11) ----------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };
__host__ __device__ static inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
    u32x4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}
// LDS bitonic sort of the split keys (rng_step_kernel): power-of-two size for W walkers, 0 if it would not fit
constexpr int RNG_SORT_MAX = 2048;
__host__ __device__ static inline int rng_sort_size(int W) {
  if (W > RNG_SORT_MAX) return 0;
  int P = 2;
  while (P < W) P <<= 1;
  return P;
}
__host__ __device__ static inline double u01_from(uint32_t hi, uint32_t lo) {
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)(v >> 11) * (1.0 / 9007199254740992.0);  // [0,1), 53 bits
}

}  // namespace gpemu

#define GP_TRY0(expr)           \
  do {                          \
    int rc0__ = (expr);         \
    if (rc0__ != GPEMU_OK) return rc0__; \
  } while (0)

namespace gpemu {

// ---- kernels ------------------------------------------------------------------------------------
// One workgroup per (step, chain) (grid = steps generated ahead x chains): random balanced split (rank of W random
// keys by counting), set member lists (ballot prefix sums), and the step's zz / rint / log u draws
// for both halves.  Step s writes slot s % RNG_RING of the ring buffers.
// Several independent chains (closure tests, ref: steer_analysis.py:168-183) share one sampler: chain c owns walkers
// c W .. c W + W - 1 of the Wt = C W stacked ones and proposals c n_h .. of each half's C n_h; its draws depend only
// on its own seed and on LOCAL indices, so it is the chain a one-chain sampler with that seed produces.  Stored
// walker / proposal indices are global (offset by the chain), which is all the evaluation kernels see.

__global__ __launch_bounds__(1024) void rng_step_kernel(int *inds_r, int *idx_r, double *zz_r,
                                                        double *logu_r, int *rint_r, double *fac_r, int *pos_r,
                                                        int W, int n0, int n1, int d, double a,
                                                        const unsigned long long *__restrict__ seeds, int Wt,
                                                        unsigned long long step0) {
  extern __shared__ unsigned long long keys[];  // [W]
  __shared__ int wcnt[2][16];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  const unsigned long long step = step0 + blockIdx.x;
  const uint32_t step_lo = (uint32_t)step, step_hi = (uint32_t)(step >> 32);
  const int slot = (int)(step % RNG_RING);
  const int ch = blockIdx.y;
  const uint32_t k0 = (uint32_t)seeds[ch], k1 = (uint32_t)(seeds[ch] >> 32);
  const int woff = ch * W;                         // first walker of this chain
  const int poff[2] = {ch * n0, ch * n1};          // first proposal of this chain in each half
  int *inds = inds_r + (size_t)slot * Wt + woff;
  int *idx = idx_r + (size_t)slot * 2 * Wt;        // [2][Wt], half h at h * Wt
  double *zz = zz_r + (size_t)slot * 2 * Wt;
  double *logu = logu_r + (size_t)slot * 2 * Wt;
  int *rint = rint_r + (size_t)slot * 2 * Wt;
  double *fac = fac_r + (size_t)slot * 2 * Wt;
  int *pos = pos_r + (size_t)slot * Wt + woff;

  // keys[w] = (32 random bits << 32) | w: unique, so the rank of every walker's key is defined.  Up to RNG_SORT_MAX
  // walkers the ranks come from a bitonic sort of the keys in LDS (P = next power of two, padded with all-ones
  // sentinels: 55 compare-exchange passes at P = 1024) instead of W comparisons per thread (one workgroup per step did
  // 1024^2 comparisons on a single CU: 41 us per batch of 16 steps with the rest of the chip idle); the sorted position of
  // key (r, w) is w's rank, written back into keys[w]'s slot of a second array.
  const int P = rng_sort_size(W);                   // 0: too many walkers for the LDS sort
  unsigned long long *ranks = keys + (P ? P : W);   // [W] (only with the sort)
  for (int w = tid; w < (P ? P : W); w += nthr) {
    if (w < W) {
      u32x4 r = philox4x32_10(u32x4{(uint32_t)w, 0u, step_lo, step_hi}, k0, k1);
      keys[w] = ((unsigned long long)r.x << 32) | (unsigned)w;
    } else {
      keys[w] = ~0ull;
    }
  }
  __syncthreads();
  if (P) {
    for (int kk = 2; kk <= P; kk <<= 1)
      for (int j = kk >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < P / 2; i += nthr) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
          const unsigned long long a = keys[lo], b = keys[hi];
          const bool up = (lo & kk) == 0;
          if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
        }
        __syncthreads();
      }
    for (int r = tid; r < W; r += nthr) ranks[(unsigned)keys[r]] = (unsigned long long)r;   // low word = walker
    __syncthreads();
  }
  // split[w] = rank(w) & 1  == (arange(W) % 2) after a uniform shuffle; lists in ascending w
  int base0 = 0, base1 = 0;
  for (int c0 = 0; c0 < W; c0 += nthr) {
    const int w = c0 + tid;
    int sp = -1;
    if (w < W) {
      int rank = 0;
      if (P) {
        rank = (int)ranks[w];
      } else {
        const unsigned long long kw = keys[w];
#pragma unroll 8
        for (int jx = 0; jx < W; ++jx) rank += (keys[jx] < kw) ? 1 : 0;
      }
      sp = rank & 1;
      inds[w] = sp;
    }
    const unsigned long long b0 = __ballot(sp == 0), b1 = __ballot(sp == 1);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) { wcnt[0][wave] = __popcll(b0); wcnt[1][wave] = __popcll(b1); }
    __syncthreads();
    int off0 = base0, off1 = base1, tot0 = 0, tot1 = 0;
    for (int v = 0; v < nwave; ++v) {
      if (v < wave) { off0 += wcnt[0][v]; off1 += wcnt[1][v]; }
      tot0 += wcnt[0][v]; tot1 += wcnt[1][v];
    }
    if (sp == 0) { const int i = poff[0] + off0 + __popcll(b0 & lt); idx[i] = woff + w; pos[w] = i; }
    if (sp == 1) { const int i = poff[1] + off1 + __popcll(b1 & lt); idx[Wt + i] = woff + w; pos[w] = i; }
    base0 += tot0; base1 += tot1;
    __syncthreads();
  }
  for (int h = 0; h < 2; ++h) {
    const int ns = h == 0 ? n0 : n1, nc = W - ns;
    for (int i = tid; i < ns; i += nthr) {
      u32x4 r = philox4x32_10(u32x4{(uint32_t)i, (uint32_t)(1 + h), step_lo, step_hi}, k0, k1);
      u32x4 r2 = philox4x32_10(u32x4{(uint32_t)i, (uint32_t)(3 + h), step_lo, step_hi}, k0, k1);
      double u = u01_from(r.x, r.y);
      double t = (a - 1.0) * u + 1.0;
      const double z = t * t / a;
      const size_t o = (size_t)h * Wt + poff[h] + i;
      zz[o] = z;
      fac[o] = (d - 1.0) * log(z);             // emcee moves/stretch.py: factors = (ndim - 1) * log(zz)
      // partner = member `randint(nc)` of the complementary set, stored as a walker index so that the
      // consumers need one dependent load less (the set lists of this step are complete: barrier above)
      const int rpos = (int)(((unsigned long long)r.z * (unsigned long long)nc) >> 32);
      rint[o] = idx[(size_t)(1 - h) * Wt + poff[1 - h] + rpos];
      logu[o] = log(u01_from(r2.x, r2.y));
    }
  }
}

// set lists from host-supplied inds (emcee order)
__global__ void build_sets_kernel(const int *inds, int *idx, int W) {
  if (threadIdx.x < 2 && blockIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < W; ++w)
      if (inds[w] == (int)threadIdx.x) idx[threadIdx.x * W + c++] = w;
  }
}

// accept / reject of one half with log-probabilities gathered from all ranks.  The proposal is
// recomputed from the ensemble (the complementary set is unchanged during this half), so no rank
// needs the other ranks' proposal rows; optionally records the chain row of its walkers.
__global__ void accept_kernel(double *__restrict__ X, double *__restrict__ logp,
                              const int *__restrict__ idx_s,
                              const double *__restrict__ zz, const int *__restrict__ partner,
                              const double *__restrict__ newlp, const double *__restrict__ logu,
                              long long *__restrict__ naccept, int *__restrict__ flags, int ns, int d,
                              double *__restrict__ chain, double *__restrict__ lpchain) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const int w = idx_s[i], j = partner[i];
  const double z = zz[i];
  const double nlp = newlp[i];
  if (nlp != nlp) atomicAdd(flags, 1);  // emcee raises on NaN log-probability
  const double oldlp = logp[w];
  const double lnpdiff = (d - 1.0) * log(z) + nlp - oldlp;
  const bool acc = lnpdiff > logu[i];
#pragma unroll
  for (int dd = 0; dd < DPAD; ++dd) {
    const double sw = X[w * DPAD + dd];
    double v = sw;
    if (acc && dd < d) {
      const double cj = X[j * DPAD + dd];
      v = cj - (cj - sw) * z;              // emcee moves/stretch.py get_proposal
      X[w * DPAD + dd] = v;
    }
    if (chain && dd < d) chain[(int64_t)w * d + dd] = v;
  }
  if (acc) {
    logp[w] = nlp;
    naccept[w] += 1;
  }
  if (lpchain) lpchain[w] = acc ? nlp : oldlp;
}

__global__ void pad_rows_kernel(const double *__restrict__ src, double *__restrict__ dst, int n, int d) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * DPAD) return;
  int r = idx / DPAD, dd = idx % DPAD;
  dst[idx] = dd < d ? src[r * d + dd] : 0.0;
}

__global__ void unpad_rows_kernel(const double *__restrict__ src, double *__restrict__ dst, int n, int d) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * d) return;
  dst[idx] = src[(idx / d) * DPAD + idx % d];
}

// ---- host helpers ---------------------------------------------------------------------------------
// log-posterior of B padded query rows summed over the groups; `pa` lets the first group's kernel
// build the rows from the ensemble (fused proposal), `aa` lets the last group's kernel finish the move
static int eval_logpost(gpemu_sampler *s, double *dq, int64_t B, double *dout, hipStream_t st,
                        const AcceptArgs *aa = nullptr, const ProposeArgs *pa = nullptr) {
  const size_t ng = s->groups.size();
  {                  // small emulators: cross-kernel and GEMM of all groups in one launch (k_halfstep.hip)
    const int rc = logpost_small(s->groups.data(), (int)ng, B, dq, dout, st, aa, pa);
    if (rc != GPEMU_ERR_UNSUPPORTED) return rc;
  }
  if (ng >= 2) {     // one launch per stage for all groups where that applies (gpemu_api.hip: logpost_groups)
    const int rc = logpost_groups(s->groups.data(), (int)ng, B, dq, dout, st, aa, pa);
    if (rc != GPEMU_ERR_UNSUPPORTED) return rc;
  }
  AcceptArgs chain_only;                 // groups before the last: no accept, but the rows' chains (data constants)
  if (aa) { chain_only.chain_per = aa->chain_per; chain_only.first = aa->first; }
  for (size_t g = 0; g < ng; ++g) {
    int rc = logpost_padded(s->groups[g], B, dq, dout, g > 0 ? 1 : 0, st, g + 1 == ng ? aa : (aa ? &chain_only : nullptr),
                            g == 0 ? pa : nullptr);
    if (rc != GPEMU_OK) return rc;
  }
  return GPEMU_OK;
}

static int ensure_chain(gpemu_sampler *s, int64_t need) {
  if (need <= s->chain_cap) return GPEMU_OK;
  int64_t cap = s->chain_cap ? s->chain_cap : 256;
  while (cap < need) cap *= 2;
  double *nc = nullptr, *nl = nullptr;
  GP_HIP(hipMalloc((void **)&nc, sizeof(double) * (size_t)(cap * s->W * s->d)));
  GP_HIP(hipMalloc((void **)&nl, sizeof(double) * (size_t)(cap * s->W)));
  if (s->chain_len > 0) {
    GP_HIP(hipMemcpyAsync(nc, s->chain, sizeof(double) * (size_t)(s->chain_len * s->W * s->d),
                          hipMemcpyDeviceToDevice, s->stream));
    GP_HIP(hipMemcpyAsync(nl, s->lpchain, sizeof(double) * (size_t)(s->chain_len * s->W),
                          hipMemcpyDeviceToDevice, s->stream));
  }
  GP_HIP(hipStreamSynchronize(s->stream));
  (void)hipFree(s->chain);
  (void)hipFree(s->lpchain);
  s->chain = nc; s->lpchain = nl; s->chain_cap = cap;
  return GPEMU_OK;
}

static int launch_rng_batch(gpemu_sampler *s, hipStream_t st, uint64_t first, int64_t n) {
  const int C = s->nchains, Wc = (int)(s->W / C);
  const int P = rng_sort_size(Wc);
  size_t shm = sizeof(unsigned long long) * (P ? (size_t)P + Wc : (size_t)Wc);   // keys (+ ranks with the LDS sort)
  hipLaunchKernelGGL(rng_step_kernel, dim3((unsigned)n, (unsigned)C), dim3(1024), shm, st, s->inds, s->idx, s->zz,
                     s->logu, s->rint, s->fac, s->pos, Wc, (int)(s->ns[0] / C), (int)(s->ns[1] / C), (int)s->d, s->a,
                     s->seeds, (int)s->W, (unsigned long long)first);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

// Make sure the randomness of step s->step_counter is in the ring.  A batch is at most RNG_BATCH steps and never
// crosses a multiple of RNG_BATCH, so it fills one half of the ring and leaves the other half (the previous steps'
// draws, still read by the fused run) untouched.  Generated in the chain's own stream: producing the next batch on
// a side stream while the current one is consumed was tried and LOST (C3: 0.262 -> 0.265 ms per step on one GPU,
// 0.187 -> 0.214 at two emulated ranks) -- the generator's workgroups land on CUs that each hold one persistent
// triangular-GEMM worker and turn those workers into stragglers.
static int launch_rng(gpemu_sampler *s, hipStream_t st, int64_t ahead = 1) {
  const uint64_t step = s->step_counter;
  if (step < s->rng_ready_until) return GPEMU_OK;
  const int64_t to_edge = RNG_BATCH - (int64_t)(step % RNG_BATCH);
  const int64_t n = ahead < 1 ? 1 : (ahead > to_edge ? to_edge : ahead);
  GP_TRY0(launch_rng_batch(s, st, step, n));
  s->rng_ready_until = step + (uint64_t)n;
  return GPEMU_OK;
}

int sampler_launch_rng(gpemu_sampler *s, hipStream_t st, int64_t ahead) { return launch_rng(s, st, ahead); }
int sampler_ensure_chain(gpemu_sampler *s, int64_t need) { return ensure_chain(s, need); }

static inline size_t rslot(const gpemu_sampler *s) { return (size_t)(s->step_counter % RNG_RING); }

static int launch_accept(gpemu_sampler *s, int h, const double *dnewlp, int store_chain, hipStream_t st) {
  const int ns = (int)s->ns[h];
  const size_t o2 = rslot(s) * 2 * s->W;
  double *chain = nullptr, *lpchain = nullptr;
  if (store_chain) {
    int rc = ensure_chain(s, s->chain_len + 1);
    if (rc != GPEMU_OK) return rc;
    chain = s->chain + s->chain_len * s->W * s->d;
    lpchain = s->lpchain + s->chain_len * s->W;
  }
  hipLaunchKernelGGL(accept_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, s->X, s->logp,
                     s->idx + o2 + h * s->W, s->zz + o2 + h * s->W,
                     s->rint + o2 + h * s->W, dnewlp, s->logu + o2 + h * s->W, s->naccept, s->flags, ns,
                     (int)s->d, chain, lpchain);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

// ProposeArgs of half h for proposals [lo, lo + n)
static ProposeArgs propose_args(gpemu_sampler *s, int h, int64_t lo, int64_t n) {
  const size_t o2 = rslot(s) * 2 * s->W;
  ProposeArgs pa;
  pa.enabled = 1;
  pa.X = s->X;
  pa.idx_s = s->idx + o2 + h * s->W + lo;
  pa.zz = s->zz + o2 + h * s->W + lo;
  pa.partner = s->rint + o2 + h * s->W + lo;
  pa.factors = s->factors + lo;
  pa.n = (int)n;
  pa.d = (int)s->d;
  return pa;
}

// propose + log-posterior + (fused) accept / record of one half on this device.  Large halves (several chains
// stacked, or more than 4096 walkers) go through in chunks of proposals: a proposal only reads walkers of the
// complementary set, which no chunk of this half modifies.  The kernel variants (row-chunk sizes of the partial sums)
// are chosen by the PER-CHAIN half size, so that a chain stacked with others is evaluated exactly as it would be alone.
static int half_step_fused(gpemu_sampler *s, int h, int store_chain, hipStream_t st) {
  const size_t o2 = rslot(s) * 2 * s->W;
  const int64_t n = s->ns[h], per_chain = n / s->nchains;
  const int64_t chunk_max = (per_chain <= 128) ? 1024 : 2048;
  for (int64_t lo = 0; lo < n; lo += chunk_max) {
    const int64_t nb = std::min<int64_t>(chunk_max, n - lo);
    const ProposeArgs pa = propose_args(s, h, lo, nb);
    AcceptArgs aa;
    aa.enabled = 1;
    aa.X = s->X; aa.logp = s->logp;
    aa.idx_s = s->idx + o2 + h * s->W + lo;
    aa.factors = s->factors + lo;
    aa.logu = s->logu + o2 + h * s->W + lo;
    aa.naccept = s->naccept; aa.flags = s->flags;
    aa.chain_per = s->nchains > 1 ? (int)per_chain : 0;
    aa.first = lo;
    if (store_chain) {
      aa.chain = s->chain + s->chain_len * s->W * s->d;
      aa.lpchain = s->lpchain + s->chain_len * s->W;
    }
    for (gpemu_model *m : s->groups) m->variant_B = per_chain;
    const int rc = eval_logpost(s, s->q, nb, s->newlp + lo, st, &aa, &pa);
    for (gpemu_model *m : s->groups) m->variant_B = 0;
    if (rc != GPEMU_OK) return rc;
  }
  return GPEMU_OK;
}

// bookkeeping after both halves (the chain row was written by the fused / accept kernels)
static int end_step(gpemu_sampler *s, int store_chain, hipStream_t st, bool recorded = true) {
  (void)st; (void)recorded;
  if (store_chain) s->chain_len += 1;
  s->iterations += 1;
  s->step_counter += 1;
  return GPEMU_OK;
}

}  // namespace gpemu

using namespace gpemu;
#define GP_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != GPEMU_OK) return rc__; \
  } while (0)

extern "C" {

int gpemu_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                     uint32_t *out4) {
  GP_ARG(out4, "out4");
  u32x4 r = philox4x32_10(u32x4{c0, c1, c2, c3}, k0, k1);
  out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
  return GPEMU_OK;
}

int gpemu_sampler_create(gpemu_sampler **out, gpemu_model *const *groups, int n_groups, int64_t W,
                         double a, uint64_t seed) {
  return gpemu_sampler_create_chains(out, groups, n_groups, W, a, &seed, 1);
}

int gpemu_sampler_create_chains(gpemu_sampler **out, gpemu_model *const *groups, int n_groups, int64_t Wc,
                                double a, const uint64_t *seeds, int n_chains) {
  GP_ARG(out && groups && n_groups > 0 && seeds, "groups / seeds");
  *out = nullptr;
  GP_ARG(a > 1.0, "stretch scale a must be > 1");
  GP_ARG(n_chains >= 1 && n_chains <= 4096, "n_chains must be in [1, 4096]");
  for (int g = 0; g < n_groups; ++g) {
    GP_ARG(groups[g], "null group");
    GP_ARG(groups[g]->d == groups[0]->d && groups[g]->device == groups[0]->device,
           "groups must share the parameter dimension and the device");
    if (!groups[g]->lik_ready) { set_error("gpemu_likelihood_setup must be called on every group first"); return GPEMU_ERR_STATE; }
    if (groups[g]->lik_chains != 1 && groups[g]->lik_chains != n_chains) {
      set_error("group %d carries data for %d chains, the sampler has %d", g, groups[g]->lik_chains, n_chains);
      return GPEMU_ERR_STATE;
    }
  }
  GP_ARG(Wc >= 2 && Wc <= 8192, "n_walkers (per chain) must be in [2, 8192]");
  const int64_t W = Wc * n_chains;          // all walkers, chain after chain
  const uint64_t seed = seeds[0];
  const int64_t d = groups[0]->d;
  GP_HIP(hipSetDevice(groups[0]->device));
  gpemu_sampler *s = new gpemu_sampler();
  s->device = groups[0]->device;
  s->groups.assign(groups, groups + n_groups);
  s->W = W; s->d = d; s->a = a; s->seed = seed;
  s->nchains = n_chains;
  s->ns[0] = (Wc + 1) / 2 * n_chains; s->ns[1] = Wc / 2 * n_chains;     // proposals of a half, chain after chain
  s->qcap = round_up(s->ns[0], TILE) + TILE;
  s->stream = groups[0]->stream;
  hipError_t e = hipSuccess;
  auto A = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 8); };
  A((void **)&s->Xbuf, sizeof(double) * 2 * W * DPAD);
  A((void **)&s->lpbuf, sizeof(double) * 2 * W);
  A((void **)&s->inds, sizeof(int) * W * RNG_RING);
  A((void **)&s->idx, sizeof(int) * 2 * W * RNG_RING);
  A((void **)&s->zz, sizeof(double) * 2 * W * RNG_RING);
  A((void **)&s->logu, sizeof(double) * 2 * W * RNG_RING);
  A((void **)&s->rint, sizeof(int) * 2 * W * RNG_RING);
  A((void **)&s->fac, sizeof(double) * 2 * W * RNG_RING);
  A((void **)&s->pos, sizeof(int) * W * RNG_RING);
  A((void **)&s->q2, sizeof(double) * 2 * s->qcap * DPAD);
  A((void **)&s->q, sizeof(double) * s->qcap * DPAD);
  A((void **)&s->factors, sizeof(double) * W);
  A((void **)&s->newlp, sizeof(double) * s->qcap);
  A((void **)&s->naccept, sizeof(long long) * W);
  A((void **)&s->flags, sizeof(int) * 2);
  A((void **)&s->seeds, sizeof(unsigned long long) * n_chains);
  if (e == hipSuccess) e = hipMemcpy(s->seeds, seeds, sizeof(unsigned long long) * n_chains, hipMemcpyHostToDevice);
  if (e == hipSuccess) { s->X = s->Xbuf; s->logp = s->lpbuf; s->cur = 0; }

  if (e == hipSuccess) e = hipMemsetAsync(s->q, 0, sizeof(double) * s->qcap * DPAD, s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(s->q2, 0, sizeof(double) * 2 * s->qcap * DPAD, s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(s->lpbuf, 0, sizeof(double) * 2 * W, s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(s->naccept, 0, sizeof(long long) * W, s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(s->flags, 0, sizeof(int) * 2, s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(s->Xbuf, 0, sizeof(double) * 2 * W * DPAD, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess) {
    set_error("sampler_create: %s", hipGetErrorString(e));
    gpemu_sampler_destroy(s);
    return GPEMU_ERR_HIP;
  }
  *out = s;
  return GPEMU_OK;
}

int gpemu_sampler_destroy(gpemu_sampler *s) {
  if (!s) return GPEMU_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  front_release(s);
  (void)hipFree(s->Xbuf); (void)hipFree(s->lpbuf); (void)hipFree(s->inds); (void)hipFree(s->idx);
  (void)hipFree(s->fac); (void)hipFree(s->pos); (void)hipFree(s->q2); (void)hipFree(s->seeds);
  (void)hipFree(s->zz); (void)hipFree(s->logu); (void)hipFree(s->rint); (void)hipFree(s->q);
  (void)hipFree(s->factors); (void)hipFree(s->newlp); (void)hipFree(s->naccept); (void)hipFree(s->flags);
  (void)hipFree(s->chain); (void)hipFree(s->lpchain);
  (void)hipFree(s->snapX); (void)hipFree(s->snaplp); (void)hipFree(s->snapacc);
  (void)hipFree(s->acf_part); (void)hipFree(s->acf_acf); (void)hipFree(s->acf_mean); (void)hipFree(s->acf_acf0);
  for (int h = 0; h < 2; ++h) { (void)hipFree(s->gmine[h]); (void)hipFree(s->gfull[h]); }
  delete s;
  return GPEMU_OK;
}

int gpemu_sampler_set_state(gpemu_sampler *s, const double *X0, const double *logp0) {
  GP_ARG(s && X0, "null pointer");
  GP_HIP(hipSetDevice(s->device));
  hipStream_t st = s->stream;
  const int64_t W = s->W, d = s->d;
  double *tmp = nullptr;
  GP_HIP(hipMalloc((void **)&tmp, sizeof(double) * W * d));
  hipError_t e = hipMemcpyAsync(tmp, X0, sizeof(double) * W * d, hipMemcpyHostToDevice, st);
  int rc = GPEMU_OK;
  if (e != hipSuccess) { set_error("set_state: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  if (rc == GPEMU_OK) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((W * DPAD + 255) / 256)), dim3(256), 0, st, tmp,
                       s->X, (int)W, (int)d);
    if (logp0) {
      e = hipMemcpyAsync(s->logp, logp0, sizeof(double) * W, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) { set_error("set_state: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
    } else {
      // evaluate all walkers; X has exactly W rows, so go through a padded scratch copy in chunks
      // chunks never straddle a chain: every chunk is a whole number of chains or a piece of one
      const int64_t Wc = W / s->nchains;
      const int64_t cap = std::min<int64_t>(s->ns[0], 1024);
      const int64_t step = (Wc <= cap) ? (cap / Wc) * Wc : cap;
      for (int64_t off = 0; off < W && rc == GPEMU_OK;) {
        int64_t nb = std::min<int64_t>(step, W - off);
        if (Wc > cap) nb = std::min<int64_t>(nb, Wc - off % Wc);
        e = hipMemcpyAsync(s->q, s->X + off * DPAD, sizeof(double) * nb * DPAD, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { set_error("set_state: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; break; }
        AcceptArgs ca;                       // not an accept: only tells the likelihood which chain a row belongs to
        ca.chain_per = s->nchains > 1 ? (int)Wc : 0;
        ca.first = off;
        for (gpemu_model *m : s->groups) m->variant_B = (Wc + 1) / 2;
        rc = eval_logpost(s, s->q, nb, s->newlp, st, &ca);
        for (gpemu_model *m : s->groups) m->variant_B = 0;
        if (rc == GPEMU_OK) {
          e = hipMemcpyAsync(s->logp + off, s->newlp, sizeof(double) * nb, hipMemcpyDeviceToDevice, st);
          if (e != hipSuccess) { set_error("set_state: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
        }
        off += nb;
      }
    }
  }
  e = hipStreamSynchronize(st);
  if (rc == GPEMU_OK && e != hipSuccess) { set_error("set_state: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  (void)hipFree(tmp);
  return rc;
}

int gpemu_sampler_get_state(gpemu_sampler *s, double *X, double *logp) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  hipStream_t st = s->stream;
  const int64_t W = s->W, d = s->d;
  if (X) {
    double *tmp = nullptr;
    GP_HIP(hipMalloc((void **)&tmp, sizeof(double) * W * d));
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)((W * d + 255) / 256)), dim3(256), 0, st, s->X,
                       tmp, (int)W, (int)d);
    hipError_t e = hipMemcpyAsync(X, tmp, sizeof(double) * W * d, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e != hipSuccess) { set_error("get_state: %s", hipGetErrorString(e)); return GPEMU_ERR_HIP; }
  }
  if (logp) {
    GP_HIP(hipMemcpyAsync(logp, s->logp, sizeof(double) * W, hipMemcpyDeviceToHost, st));
    GP_HIP(hipStreamSynchronize(st));
  }
  return GPEMU_OK;
}

int gpemu_sampler_reset(gpemu_sampler *s) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  GP_HIP(hipMemsetAsync(s->naccept, 0, sizeof(long long) * s->W, s->stream));
  GP_HIP(hipStreamSynchronize(s->stream));
  s->chain_len = 0;
  s->iterations = 0;
  return GPEMU_OK;
}

// The chain state as it stands (ensemble, log-probabilities, acceptance counters, step / chain counters), kept on the
// device; gpemu_sampler_restore puts it back.  The random stream is counter based (Philox: counter = step), so a block
// of steps rerun after a restore draws what the failed attempt drew: the chain is that of an unbroken run.
int gpemu_sampler_snapshot(gpemu_sampler *s) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  const int64_t W = s->W;
  if (!s->snapX) {
    GP_HIP(hipMalloc((void **)&s->snapX, sizeof(double) * W * DPAD));
    GP_HIP(hipMalloc((void **)&s->snaplp, sizeof(double) * W));
    GP_HIP(hipMalloc((void **)&s->snapacc, sizeof(long long) * W));
  }
  GP_HIP(hipMemcpyAsync(s->snapX, s->X, sizeof(double) * W * DPAD, hipMemcpyDeviceToDevice, s->stream));
  GP_HIP(hipMemcpyAsync(s->snaplp, s->logp, sizeof(double) * W, hipMemcpyDeviceToDevice, s->stream));
  GP_HIP(hipMemcpyAsync(s->snapacc, s->naccept, sizeof(long long) * W, hipMemcpyDeviceToDevice, s->stream));
  s->snap_step_counter = s->step_counter;
  s->snap_iterations = s->iterations;
  s->snap_chain_len = s->chain_len;
  s->snap_valid = true;
  return GPEMU_OK;
}

int gpemu_sampler_restore(gpemu_sampler *s) {
  GP_ARG(s, "sampler");
  if (!s->snap_valid) { set_error("gpemu_sampler_restore without a snapshot"); return GPEMU_ERR_STATE; }
  GP_HIP(hipSetDevice(s->device));
  const int64_t W = s->W;
  GP_HIP(hipStreamSynchronize(s->stream));
  GP_HIP(hipMemcpyAsync(s->X, s->snapX, sizeof(double) * W * DPAD, hipMemcpyDeviceToDevice, s->stream));
  GP_HIP(hipMemcpyAsync(s->logp, s->snaplp, sizeof(double) * W, hipMemcpyDeviceToDevice, s->stream));
  GP_HIP(hipMemcpyAsync(s->naccept, s->snapacc, sizeof(long long) * W, hipMemcpyDeviceToDevice, s->stream));
  GP_HIP(hipMemsetAsync(s->flags, 0, sizeof(int) * 2, s->stream));
  GP_HIP(hipStreamSynchronize(s->stream));
  s->step_counter = s->snap_step_counter;
  s->rng_ready_until = 0;              // the ring may hold later steps' draws in these slots: generate again
  s->iterations = s->snap_iterations;
  s->chain_len = s->snap_chain_len;    // rows written by the failed attempt are overwritten
  return GPEMU_OK;
}

static int check_nan(gpemu_sampler *s) {
  int flag = 0;
  GP_HIP(hipMemcpyAsync(&flag, s->flags, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  GP_HIP(hipStreamSynchronize(s->stream));
  if (flag) {
    (void)hipMemsetAsync(s->flags, 0, sizeof(int), s->stream);
    set_error("log-probability returned NaN for %d proposals (emcee raises ValueError here)", flag);
    return 1;
  }
  return GPEMU_OK;
}

int gpemu_sampler_run(gpemu_sampler *s, int64_t steps, int store_chain) {
  GP_ARG(s && steps >= 0, "sampler / steps");
  GP_HIP(hipSetDevice(s->device));
  hipStream_t st = s->stream;
  // On one GPU the three-launch half-step below is the faster form (0.2275 vs 0.2516 ms per step at C3,
  // tools/time_fused_single.py: the fused front kernel runs its likelihood and cross-kernel phases back to back); the two-launch form at one rank is
  // reachable through gpemu_sampler_run_peer with a one-rank import (test_fused_run_world1_equals_three_launch_run).
  if (store_chain) GP_TRY(ensure_chain(s, s->chain_len + steps));
  for (int64_t it = 0; it < steps; ++it) {
    GP_TRY(launch_rng(s, st, steps - it));
    for (int h = 0; h < 2; ++h) GP_TRY(half_step_fused(s, h, store_chain, st));
    GP_TRY(end_step(s, store_chain, st, true));
  }
  return check_nan(s);
}

int gpemu_sampler_step_host_rng(gpemu_sampler *s, const int32_t *inds, const double *zz,
                                const int64_t *rint, const double *logu, int store_chain) {
  GP_ARG(s && inds && zz && rint && logu, "null pointer");
  GP_HIP(hipSetDevice(s->device));
  hipStream_t st = s->stream;
  const int64_t W = s->W;
  int64_t n0 = 0;
  for (int64_t w = 0; w < W; ++w) {
    GP_ARG(inds[w] == 0 || inds[w] == 1, "inds must be 0/1");
    n0 += (inds[w] == 0);
  }
  GP_ARG(n0 == s->ns[0], "inds must hold ceil(W/2) zeros");
  std::vector<int> hr(2 * W, 0);
  std::vector<double> hz(2 * W, 1.0), hu(2 * W, 0.0);
  std::vector<int> members[2];                         // set lists in ascending walker order (as build_sets_kernel)
  for (int64_t w = 0; w < W; ++w) members[inds[w]].push_back((int)w);
  for (int h = 0; h < 2; ++h) {
    const int64_t off = h == 0 ? 0 : s->ns[0], nc = W - s->ns[h];
    for (int64_t i = 0; i < s->ns[h]; ++i) {
      GP_ARG(rint[off + i] >= 0 && rint[off + i] < nc, "rint out of range");
      hr[h * W + i] = members[1 - h][(size_t)rint[off + i]];   // partner as a walker index
      hz[h * W + i] = zz[off + i];
      hu[h * W + i] = logu[off + i];
    }
  }
  const size_t sl = rslot(s);
  s->rng_ready_until = 0;  // host-supplied draws replace whatever the device generated ahead
  GP_HIP(hipMemcpyAsync(s->inds + sl * W, inds, sizeof(int) * W, hipMemcpyHostToDevice, st));
  GP_HIP(hipMemcpyAsync(s->rint + sl * 2 * W, hr.data(), sizeof(int) * 2 * W, hipMemcpyHostToDevice, st));
  GP_HIP(hipMemcpyAsync(s->zz + sl * 2 * W, hz.data(), sizeof(double) * 2 * W, hipMemcpyHostToDevice, st));
  GP_HIP(hipMemcpyAsync(s->logu + sl * 2 * W, hu.data(), sizeof(double) * 2 * W, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(build_sets_kernel, dim3(1), dim3(64), 0, st, s->inds + sl * W, s->idx + sl * 2 * W, (int)W);
  GP_TRY(ensure_chain(s, s->chain_len + 1));
  for (int h = 0; h < 2; ++h) GP_TRY(half_step_fused(s, h, store_chain, st));
  GP_TRY(end_step(s, store_chain, st, true));
  return check_nan(s);  // also synchronises, so the host staging vectors may go out of scope
}

int gpemu_sampler_get_chain(gpemu_sampler *s, int64_t first, int64_t n, double *chain_out,
                            double *logp_out) {
  GP_ARG(s && first >= 0 && n >= 0 && first + n <= s->chain_len, "chain range");
  GP_HIP(hipSetDevice(s->device));
  if (n == 0) return GPEMU_OK;
  if (chain_out)
    GP_HIP(hipMemcpyAsync(chain_out, s->chain + first * s->W * s->d, sizeof(double) * n * s->W * s->d,
                          hipMemcpyDeviceToHost, s->stream));
  if (logp_out)
    GP_HIP(hipMemcpyAsync(logp_out, s->lpchain + first * s->W, sizeof(double) * n * s->W,
                          hipMemcpyDeviceToHost, s->stream));
  GP_HIP(hipStreamSynchronize(s->stream));
  return GPEMU_OK;
}

int gpemu_sampler_get_counts(gpemu_sampler *s, int64_t *naccepted, int64_t *iterations,
                             int64_t *chain_len) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  if (naccepted) {
    GP_HIP(hipMemcpyAsync(naccepted, s->naccept, sizeof(long long) * s->W, hipMemcpyDeviceToHost, s->stream));
    GP_HIP(hipStreamSynchronize(s->stream));
  }
  if (iterations) *iterations = s->iterations;
  if (chain_len) *chain_len = s->chain_len;
  return GPEMU_OK;
}

// ---- phases for the multi-GPU driver (walkers sharded over ranks; RCCL all-gather in between) ----
int gpemu_sampler_reserve_chain(gpemu_sampler *s, int64_t additional_steps) {
  GP_ARG(s && additional_steps >= 0, "sampler / steps");
  GP_HIP(hipSetDevice(s->device));
  return ensure_chain(s, s->chain_len + additional_steps);
}

int gpemu_sampler_begin_step(gpemu_sampler *s) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  return launch_rng(s, s->stream, RNG_BATCH);
}

int gpemu_sampler_half_propose_eval(gpemu_sampler *s, int half, int64_t lo, int64_t hi,
                                    double *dnewlp_slice) {
  GP_ARG(s && (half == 0 || half == 1), "half");
  GP_ARG(lo >= 0 && lo <= hi && hi <= s->ns[half] && dnewlp_slice, "slice");
  GP_HIP(hipSetDevice(s->device));
  if (hi > lo) {
    // the cross-kernel kernel builds this rank's proposal rows itself; the likelihood kernel writes the
    // log-probabilities straight into the caller's all-gather buffer
    const ProposeArgs pa = propose_args(s, half, lo, hi - lo);
    GP_TRY(eval_logpost(s, s->q, hi - lo, dnewlp_slice, s->stream, nullptr, &pa));
  }
  return GPEMU_OK;
}

int gpemu_sampler_half_accept(gpemu_sampler *s, int half, const double *dnewlp_all, int store_chain) {
  GP_ARG(s && (half == 0 || half == 1) && dnewlp_all, "half / newlp");
  GP_HIP(hipSetDevice(s->device));
  return launch_accept(s, half, dnewlp_all, store_chain, s->stream);
}

int gpemu_sampler_end_step(gpemu_sampler *s, int store_chain) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  return end_step(s, store_chain, s->stream, true);   // the accept kernels recorded the chain row
}

int gpemu_sampler_check(gpemu_sampler *s) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  return check_nan(s);
}

int gpemu_sampler_set_stream(gpemu_sampler *s, void *stream) {
  GP_ARG(s, "sampler");
  GP_HIP(hipSetDevice(s->device));
  GP_HIP(hipStreamSynchronize(s->stream));
  s->stream = stream ? (hipStream_t)stream : s->groups[0]->stream;
  return GPEMU_OK;
}

// ---- RCCL communicator + the whole sharded run in one call -----------------------------------------
// librccl is bound at run time (dlopen) so that the process keeps ONE copy of it: the host side passes the
// path of the copy torch.distributed already loaded, or NULL for the system "librccl.so".
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // optional
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;   // optional
};
static RcclApi g_rccl;

static int load_rccl(const char *path) {
  if (g_rccl.lib) return GPEMU_OK;
  const char *name = (path && path[0]) ? path : "librccl.so";
  void *lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
  if (!lib) { set_error("cannot load RCCL (%s): %s", name, dlerror()); return GPEMU_ERR_STATE; }
  RcclApi api;
  api.lib = lib;
  *(void **)&api.GetUniqueId = dlsym(lib, "ncclGetUniqueId");
  *(void **)&api.CommInitRank = dlsym(lib, "ncclCommInitRank");
  *(void **)&api.CommDestroy = dlsym(lib, "ncclCommDestroy");
  *(void **)&api.AllGather = dlsym(lib, "ncclAllGather");
  *(void **)&api.GetErrorString = dlsym(lib, "ncclGetErrorString");
  *(void **)&api.CommCount = dlsym(lib, "ncclCommCount");
  *(void **)&api.CommUserRank = dlsym(lib, "ncclCommUserRank");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
    set_error("%s does not export the RCCL entry points", name);
    dlclose(lib);
    return GPEMU_ERR_STATE;
  }
  g_rccl = api;
  return GPEMU_OK;
}

#define GP_RCCL(call)                                                                      \
  do {                                                                                     \
    ncclResult_t r_ = (call);                                                              \
    if (r_ != ncclSuccess) {                                                               \
      set_error("%s: %s", #call, g_rccl.GetErrorString(r_));                               \
      return GPEMU_ERR_HIP;                                                                \
    }                                                                                      \
  } while (0)

struct gpemu_comm {
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
};

int gpemu_comm_unique_id(const char *librccl_path, char *id_out128) {
  GP_ARG(id_out128, "id_out");
  GP_TRY(load_rccl(librccl_path));
  ncclUniqueId id;
  GP_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return GPEMU_OK;
}

int gpemu_comm_create(gpemu_comm **out, int device, int rank, int world, const char *id128,
                      const char *librccl_path) {
  GP_ARG(out && id128, "null pointer");
  GP_ARG(world >= 1 && rank >= 0 && rank < world, "rank / world");
  GP_TRY(load_rccl(librccl_path));
  GP_HIP(hipSetDevice(device));
  gpemu_comm *c = new gpemu_comm();
  c->device = device; c->rank = rank; c->world = world;
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank: %s", g_rccl.GetErrorString(r));
    delete c;
    return GPEMU_ERR_HIP;
  }
  *out = c;
  return GPEMU_OK;
}

int gpemu_comm_destroy(gpemu_comm *c) {
  if (!c) return GPEMU_OK;
  (void)hipSetDevice(c->device);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return GPEMU_OK;
}

int gpemu_comm_dims(const gpemu_comm *c, int *rank, int *world) {
  GP_ARG(c, "comm");
  // what the communicator itself reports (ncclCommCount / ncclCommUserRank), not what it was asked to be
  int r = c->rank, w = c->world;
  if (c->comm && g_rccl.CommCount) GP_RCCL(g_rccl.CommCount(c->comm, &w));
  if (c->comm && g_rccl.CommUserRank) GP_RCCL(g_rccl.CommUserRank(c->comm, &r));
  if (rank) *rank = r;
  if (world) *world = w;
  return GPEMU_OK;
}

int gpemu_comm_all_gather(gpemu_comm *c, const double *dsend, double *drecv, int64_t count, void *stream) {
  GP_ARG(c && dsend && drecv && count > 0, "all_gather arguments");
  GP_HIP(hipSetDevice(c->device));
  GP_RCCL(g_rccl.AllGather(dsend, drecv, (size_t)count, ncclDouble, c->comm, (hipStream_t)stream));
  return GPEMU_OK;
}

__global__ void fill_kernel(double *p, int64_t n, double v) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// Walkers sharded over the communicator's ranks: every rank draws the same randomness, evaluates its block
// of each half's proposals, all-gathers the log-probabilities (8 bytes per proposal) and applies the same
// accept decisions to its replica of the ensemble.  emulate_world > 0 (measurement aid, communicator of one
// rank): evaluate only the share rank 0 of an `emulate_world`-rank job would; everything else is rejected.
int gpemu_sampler_run_sharded(gpemu_sampler *s, gpemu_comm *c, int64_t steps, int store_chain,
                              int emulate_world) {
  GP_ARG(s && steps >= 0 && emulate_world >= 0, "sampler / steps / emulate_world");
  GP_HIP(hipSetDevice(s->device));
  // the timing aid needs no communicator where the fused two-launch half-step applies (bench.py's scaling_model leg)
  if (emulate_world > 0 && front_eligible_for(s, emulate_world)) return front_run(s, steps, store_chain, emulate_world, 0, true);
  if (emulate_world > 0 && !c) {
    set_error("emulate_world without a communicator: this sampler is outside the fused run's limits");
    return GPEMU_ERR_UNSUPPORTED;
  }
  GP_ARG(c, "comm");
  GP_ARG(c->device == s->device, "communicator and sampler are bound to different devices");
  GP_ARG(emulate_world == 0 || c->world == 1, "emulate_world needs a one-rank communicator");
  hipStream_t st = s->stream;
  const int world = c->world;
  const int split = emulate_world > 0 ? emulate_world : world;
  int64_t lo[2], hi[2];
  for (int h = 0; h < 2; ++h) {
    const int64_t share = (s->ns[h] + split - 1) / split;       // == sampler.shard_bounds()
    const int64_t per = emulate_world > 0 ? s->ns[h] : share;
    const int r = emulate_world > 0 ? 0 : c->rank;
    lo[h] = std::min<int64_t>((int64_t)r * share, s->ns[h]);
    hi[h] = std::min<int64_t>(lo[h] + share, s->ns[h]);
    if (s->gworld != world || s->gper[h] != per) {
      (void)hipFree(s->gmine[h]); (void)hipFree(s->gfull[h]);
      s->gmine[h] = s->gfull[h] = nullptr;
      GP_HIP(hipMalloc((void **)&s->gmine[h], sizeof(double) * per));
      GP_HIP(hipMalloc((void **)&s->gfull[h], sizeof(double) * per * world));
      s->gper[h] = per;
    }
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, s->gmine[h], per,
                       emulate_world > 0 ? -INFINITY : 0.0);
  }
  s->gworld = world;
  // A rank that hits an error of its own keeps taking part in the remaining all-gathers (its peers are inside them and
  // would otherwise wait forever), skips its own launches and reports the first error at the end.
  int err = GPEMU_OK;
  auto keep = [&](int rc) { if (rc != GPEMU_OK && err == GPEMU_OK) err = rc; return err == GPEMU_OK; };
  if (store_chain) keep(ensure_chain(s, s->chain_len + steps));
  for (int64_t it = 0; it < steps; ++it) {
    if (err == GPEMU_OK) keep(launch_rng(s, st, steps - it));
    for (int h = 0; h < 2; ++h) {
      if (err == GPEMU_OK && hi[h] > lo[h]) {
        const ProposeArgs pa = propose_args(s, h, lo[h], hi[h] - lo[h]);
        keep(eval_logpost(s, s->q, hi[h] - lo[h], s->gmine[h], st, nullptr, &pa));
      }
      const ncclResult_t r = g_rccl.AllGather(s->gmine[h], s->gfull[h], (size_t)s->gper[h], ncclDouble, c->comm, st);
      if (r != ncclSuccess) {                   // the communicator itself failed: nothing more can be exchanged
        set_error("ncclAllGather: %s", g_rccl.GetErrorString(r));
        return err != GPEMU_OK ? err : GPEMU_ERR_HIP;
      }
      if (err == GPEMU_OK) keep(launch_accept(s, h, s->gfull[h], store_chain, st));
    }
    if (err == GPEMU_OK) keep(end_step(s, store_chain, st, true));
  }
  if (err != GPEMU_OK) { (void)hipStreamSynchronize(st); return err; }
  return check_nan(s);
}

}  // extern "C"

namespace gpemu {
int sampler_check_nan(gpemu_sampler *s) { return check_nan(s); }
}  // namespace gpemu
