// Internal declarations shared by the translation units of libgpemu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gpemu.h"

namespace gpemu {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define GP_HIP(call)                                                                        \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) {                                                                \
      gpemu::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,    \
                       __LINE__);                                                           \
      return GPEMU_ERR_HIP;                                                                 \
    }                                                                                       \
  } while (0)
#define GP_ARG(cond, msg)                \
  do {                                   \
    if (!(cond)) {                       \
      gpemu::set_error("bad argument: %s", msg); \
      return GPEMU_ERR_ARG;              \
    }                                    \
  } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

constexpr int DPAD = 8;        // parameter dimensions padded to 8 (reference uses d = 6 or 7)
constexpr int TILE = 128;      // row / column tile of the triangular GEMM
constexpr int KSTAR_ROWS_BIG = 64;    // training rows per workgroup of the cross-kernel: batches of more than 256 columns
constexpr int KSTAR_ROWS_SMALL = 32;  // ... and of at most KSTAR_SMALL_MAX
constexpr int KSTAR_SMALL_MAX = 128;  // (256 until round 4: at 129 .. 256 columns the 32-row form is 1 280 front workgroups against 768 resident)

// ---- device model -----------------------------------------------------------------------------
struct Workspace {
  int64_t Bcap = 0;            // padded batch capacity (multiple of TILE)
  double *Xq = nullptr;        // [Bcap][DPAD]   query points (padded with 0)
  double *KS = nullptr;        // [k][Npad][Bcap] cross-kernel K_*^T per PC
  double *mean_part = nullptr; // [Bcap][k][nchunk] partial K_* . alpha per 64- (or 32-) row chunk
  double *mean_part2 = nullptr; // second copy: the fused sampler run reads one half-step's while the next is written
  double *vsq_part = nullptr;  // [Bcap][k][nrb]    partial ||W k_*||^2 per 64- (or 32-) row block
  double *mean = nullptr;      // [Bcap][k]
  double *var = nullptr;       // [Bcap][k]
  double *logp = nullptr;      // [Bcap]
  int cur_nrb = 0;             // row blocks of vsq_part written by the last triangular GEMM launch
  int cur_nchunk = 0;          // row chunks of mean_part written by the last kstar launch
};

}  // namespace gpemu

struct gpemu_model {
  int device = 0;
  int64_t N = 0, d = 0, F = 0, k = 0;
  int64_t Npad = 0;            // N rounded up to TILE
  int num_cu = 256;
  // LPT schedule of the persistent triangular GEMM for the current number of column tiles
  void *sched_items = nullptr; // TrmmItem[sched_workers][sched_max_items]
  int *sched_cnt = nullptr;    // [sched_workers]
  int sched_ncb = -1, sched_max_items = 0, sched_workers = 0;
  int sched_cap = 0;           // worker cap the current schedule was built for
  struct SchedEntry { int ncb, cap; void *items; int *cnt; int max_items, workers; };
  std::vector<SchedEntry> sched_cache;   // every schedule built so far (sched_items / sched_cnt point into one of them)
  int64_t vsq_nrb = 0;         // row blocks of partial ||W k_*||^2 the triangular GEMM writes
  // schedule of the small-batch triangular GEMM (k_trmm_small.hip) for the current number of 32-column blocks
  void *sm_items = nullptr;
  int *sm_cnt = nullptr;
  int sm_ncb = -1, sm_max_items = 0, sm_workers = 0, sm_cap = 0;
  // every small-batch schedule built so far, keyed by (ncb, cap): the stand-alone launch (cap = num_cu) and the launch
  // shared with other groups (cap = -worker cap) alternate in a drop-in run (log_posterior(X) of a few rows between
  // sampler blocks), and a launch in flight keeps reading the one it was given -- so none is freed before the model is
  // (a few KB each; bounded: the oldest goes, after a stream sync, beyond 16)
  std::vector<SchedEntry> sm_cache;
  // likelihood with the observable blocks on different workgroups (k_loglik.hip: loglik_tasks_kernel): the terms' way
  // to the workgroup that adds them, and the tickets that tell which one that is
  double *lik_terms = nullptr;      // [lik_terms_cap][64]
  unsigned *lik_tickets = nullptr;  // [lik_terms_cap], zero between launches
  int64_t lik_terms_cap = 0;
  int kernel_kind = 0;
  double nu = 0;
  int has_const = 0, has_noise = 0;
  hipStream_t stream = nullptr;

  // per-PC GP state on the device
  // cross-kernel operands for the matrix cores (kstar_host.h; predict_dev.h: kstar_mfma_block)
  int ksteps = 2;              // MFMA k-steps of the augmented product: 4 ksteps >= d + 1
  double *Xa = nullptr;        // [k][Npad/16][ksteps][64]  augmented, centred training rows in A-fragment order
  double *alf = nullptr;       // [k][Npad/16][16]          alpha in accumulator-row order
  double *qsc = nullptr;       // [k][4 ksteps]             query side: q' = q qsc + qof
  double *qof = nullptr;
  double *etab = nullptr;      // [2^KSTAR_TB]              2^(j / 2^KSTAR_TB)
  // Matern-0.5 only (the direct distance of near-coincident pairs): row-major scaled rows, else null
  double *Xs = nullptr;        // [k][Npad][DPAD]  X_train / ls_p  (padded rows/dims = 0)
  double *inv_ls = nullptr;    // [k][DPAD]        1 / ls (the query side multiplies; the training side X / ls is exact)
  double *ls = nullptr;        // [k][DPAD]        length scales (padded dims = 1)
  double *constv = nullptr;    // [k]
  double *kdiag = nullptr;     // [k]  kernel_.diag = 1 (+const) (+noise)
  double *alpha = nullptr;     // [k][Npad] (padded = 0)
  double *Wt = nullptr;        // [k][Npad][Npad]  Wt[p][j][i] = (L_p^-1)[i][j]  (upper triangular)

  // PCA / scaler
  double *comp = nullptr;      // [k][F]
  double *smean = nullptr;     // [F]
  double *sscale = nullptr;    // [F]
  double *cunexpl = nullptr;   // [F][F] (zeros if not given)

  // likelihood state (gpemu_likelihood_setup)
  bool lik_ready = false;
  double n_div = 1.0;
  int lik_chains = 1;          // data vectors the likelihood was set up for (one per chain of a multi-chain sampler)
  int64_t variant_B = 0;       // if > 0: pick the kernel variants as for a batch of this size (sampler with several chains)
  // the constants (G, g0, scal) depend on the data AND on n_div (the reference divides the truncation covariance by
  // the number of in-bounds rows of each call): one entry per n_div seen with the current data, so that a batch
  // size that comes back costs nothing.  G / g0 / scal below point into the current entry.
  struct LikEntry { double n_div; double *G, *g0, *scal; };
  std::vector<LikEntry> lik_cache;
  std::vector<double> lik_host;        // y_exp | y_err | lo | hi | block starts the cache belongs to
  double *yexp = nullptr, *yerr = nullptr, *lo = nullptr, *hi = nullptr;  // [F],[F],[DPAD],[DPAD]
  int64_t nblk = 1;            // observable blocks of the (block-diagonal) covariance
  int *blk_start = nullptr;    // [nblk+1] first feature of each block
  int *blk_of = nullptr;       // [F]      block index of each feature
  double *G = nullptr;         // [nblk][k][k]   U_o^T A_o^-1 U_o
  double *g0 = nullptr;        // [nblk][k]      U_o^T A_o^-1 r0_o
  double *scal = nullptr;      // [nblk][2]      q0_o, logdet A_o

  // exact-form (validation) scratch: per-workgroup Sigma, panel and residual
  double *exact_scratch = nullptr;
  int64_t exact_scratch_size = 0;

  // optional per-kernel timing (gpemu_model_profile): HIP event pairs around the two hot kernels
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;         // reusable events
  std::vector<std::pair<int, int>> ev_trmm, ev_kstar;  // indices into ev_pool (start, stop)
  size_t ev_next = 0;
  double prof_ms[2] = {0.0, 0.0};          // accumulated: [0] trmm_vsq, [1] kstar
  int64_t prof_n[2] = {0, 0};

  gpemu::Workspace ws;
};

namespace gpemu {
// optional fused stretch-move finish (accept / reject + chain record) for the walker of each proposal
struct AcceptArgs {
  int enabled = 0;
  double *X = nullptr;            // [W][DPAD] ensemble positions (updated in place)
  double *logp = nullptr;         // [W]
  const int *idx_s = nullptr;     // [ns] walker of proposal i
  const double *factors = nullptr;  // [ns] (d-1) log zz
  const double *logu = nullptr;   // [ns] log of the accept uniform
  long long *naccept = nullptr;   // [W]
  int *flags = nullptr;           // [1] NaN counter
  double *chain = nullptr;        // [W][d] row of this step, or null
  double *lpchain = nullptr;      // [W]
  // several chains stacked in one batch: row b of the launch is row first + b of the stacked list, which holds
  // chain_per rows per chain; 0 = one chain.  Selects the chain's data constants (g0, q0) in the likelihood.
  int chain_per = 0;
  int64_t first = 0;
};

// optional fused stretch-move proposal: kstar_kernel builds its query rows from the ensemble
// (q_i = c[rint_i] - (c[rint_i] - s_i) zz_i, emcee moves/stretch.py) instead of reading them
struct ProposeArgs {
  int enabled = 0;
  const double *X = nullptr;      // [W][DPAD]
  const int *idx_s = nullptr;     // [n] walker of proposal i (already offset to the evaluated slice)
  const double *zz = nullptr;     // [n]
  const int *partner = nullptr;   // [n] walker index of the complementary-set member drawn for proposal i
  double *factors = nullptr;      // [n] out: (d - 1) log zz
  int n = 0, d = 0;
  // enabled == 0 and raw != nullptr: the queries come as caller rows [n][d] (gpemu_gp_predict / predict_full); the
  // kernel pads them to [..][DPAD] on the fly and stores the padded rows once (what pad_queries_kernel used to do)
  const double *raw = nullptr;
};

int ensure_workspace(gpemu_model *m, int64_t B);
// base kernel of the cross-kernel templates: 0 RBF, 1 / 2 / 3 Matern 0.5 / 1.5 / 2.5
static inline int kstar_kind(const gpemu_model *m) {
  if (m->kernel_kind != GPEMU_KERNEL_MATERN) return 0;
  return (m->nu == 0.5) ? 1 : (m->nu == 1.5 ? 2 : 3);
}

// kernels (launchers; all asynchronous on `st`)
// dXq_padded is read, or -- with pa->enabled -- written (rows [0, round_up(B, 128))) by the kernel
int launch_kstar(gpemu_model *m, int64_t B, double *dXq_padded, hipStream_t st, const ProposeArgs *pa = nullptr);
int launch_trmm_vsq(gpemu_model *m, int64_t B, hipStream_t st);
int small_trmm_xcd_of(const gpemu_model *m, int64_t B, int p, int64_t col);
int trmm_xcd_of(const gpemu_model *m, int64_t B, int p, int64_t col);   // XCD that reads K_*^T rows (p, col) in launch_trmm_vsq(m, B); -1: any
int launch_trmm_vsq_small(gpemu_model *m, int64_t B, hipStream_t st);   // B <= 128; GPEMU_ERR_UNSUPPORTED if the shape does not fit
int launch_reduce_mean_var(gpemu_model *m, int64_t B, double *dmean, double *dvar, hipStream_t st);
int launch_loglik_lowrank(gpemu_model *m, int64_t B, const double *dXq_padded, double *dout,
                          int accumulate, hipStream_t st, const AcceptArgs *aa = nullptr);
int logpost_padded(gpemu_model *m, int64_t B, double *dXq, double *dout, int accumulate,
                   hipStream_t st, const AcceptArgs *aa = nullptr, const ProposeArgs *pa = nullptr);
// several emulation groups, one launch per stage instead of one per group and stage (k_predict.hip, k_trmm_small.hip,
// k_loglik.hip): the same arithmetic as the per-group launches.  logpost_groups returns GPEMU_ERR_UNSUPPORTED -- nothing
// launched, no error set -- where the per-group launches must be used.
int launch_kstar_groups(gpemu_model *const *ms, int ng, int64_t B, double *dXq_padded, hipStream_t st, const ProposeArgs *pa);
int prepare_trmm_vsq_small_groups(gpemu_model *const *ms, int ng, int64_t B, hipStream_t st);   // the schedules only
int launch_trmm_vsq_small_groups(gpemu_model *const *ms, int ng, int64_t B, hipStream_t st);
int launch_loglik_groups(gpemu_model *const *ms, int ng, int64_t B, const double *dXq_padded, double *dout, int accumulate,
                         hipStream_t st, const AcceptArgs *aa);
int logpost_groups(gpemu_model *const *ms, int ng, int64_t B, double *dXq, double *dout, hipStream_t st,
                   const AcceptArgs *aa, const ProposeArgs *pa);
// small emulators (N <= 256 design points, k_halfstep.hip): cross-kernel + triangular GEMM of all groups in one launch,
// then the likelihood launch; the bits of the general path.  GPEMU_ERR_UNSUPPORTED (nothing launched) where it does not apply
int launch_halfstep_small(gpemu_model *const *ms, int ng, int64_t B, double *dXq_padded, hipStream_t st, const ProposeArgs *pa);
int logpost_small(gpemu_model *const *ms, int ng, int64_t B, double *dXq, double *dout, hipStream_t st,
                  const AcceptArgs *aa, const ProposeArgs *pa);
// fit-side building blocks (k_fit.hip): in-place blocked Cholesky of an Np x Np matrix (Np multiple of 64) with the
// inverted diagonal blocks in Dinv [Np/64][64][64], and W = L^-1 from it (T: Np x Np scratch)
// Optional look-ahead of the blocked Cholesky: a second (lower-priority) stream that applies a panel's update to the
// columns beyond the next panel while the next panel is factored on the caller's stream (k_fit.hip).
struct CholOverlap {
  hipStream_t side = nullptr;
  hipEvent_t panel_done = nullptr, rest_done = nullptr;
  int *flags = nullptr;      // device, 20 ints per problem: enables the one-launch-per-panel kernel (k_fit.hip)
};
int device_cholesky_blocked(double *A, int64_t Np, double *Dinv, int *dinfo, hipStream_t st, int nb = 1,
                            const CholOverlap *ov = nullptr);
int device_trtri_blocked(const double *L, int64_t Np, const double *Dinv, double *W, double *T, hipStream_t st, int nb = 1,
                         bool zero_upper = true);
int device_invert_factor_to_Wt(const double *dL, int64_t N, double *Wt, int64_t Npad, double *A, double *Dinv,
                               double *W, double *T, hipStream_t st);
// profiling helpers: record an event on `st` and return its pool index (-1 when profiling is off)
int prof_mark(gpemu_model *m, hipStream_t st);
void prof_pair(gpemu_model *m, int which, int e0, int e1);
}  // namespace gpemu
