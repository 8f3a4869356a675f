/*
 * gpemu.h -- C ABI of libgpemu.so: the MI355X (gfx950) implementation of the GP-emulator +
 * MCMC log-posterior hot path of jdmulligan/bayesian-inference.
 *
 * The reference has no FFI of its own (it is pure Python); each entry point below replaces the
 * arithmetic behind one reference call site, cited as  ref: <file>:<lines>  relative to
 * /root/reference/src/bayesian_inference/, and  skl: <file>:<lines>  for the scikit-learn 1.7.2
 * code the reference delegates to.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  All real arrays are row-major float64, indices int64.
 *   - Buffers are caller-allocated.  Functions named *_dev take DEVICE pointers and a hipStream_t
 *     (passed as void*; NULL = the handle's own stream) and do not synchronise; the others take
 *     HOST pointers, copy, run and synchronise before returning.
 *   - Every function returns int: 0 = ok, <0 = argument / runtime error, >0 = numerical failure
 *     (e.g. index+1 of a non-positive Cholesky pivot).  gpemu_last_error() gives the thread-local
 *     message of the last failure.
 *   - Opaque handles own all device memory they allocate and are bound to one HIP device.  A handle
 *     is not thread-safe (one host thread per handle); calls release no Python state, so ctypes may
 *     drop the GIL around them.
 *   - There is NO CPU implementation behind these symbols: without a HIP device every compute
 *     entry point fails with GPEMU_ERR_NO_DEVICE.
 */
#ifndef GPEMU_H
#define GPEMU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPEMU_OK 0
#define GPEMU_ERR_ARG (-1)
#define GPEMU_ERR_HIP (-2)
#define GPEMU_ERR_NO_DEVICE (-3)
#define GPEMU_ERR_STATE (-4)
#define GPEMU_ERR_UNSUPPORTED (-5)

/* kernel_kind (ref: emulation.py:133-149) */
#define GPEMU_KERNEL_RBF 0    /* skl gaussian_process/kernels.py:1553-1582 */
#define GPEMU_KERNEL_MATERN 1 /* skl gaussian_process/kernels.py:1708-1781, nu in {0.5,1.5,2.5} */

/* gpemu_logpost mode */
#define GPEMU_LOGPOST_LOWRANK 0 /* k x k Woodbury form, the throughput path          */
#define GPEMU_LOGPOST_EXACT 1   /* materialise Sigma (F x F) + batched Cholesky, reference form  */

typedef struct gpemu_model gpemu_model;     /* one emulation group on one device */
typedef struct gpemu_sampler gpemu_sampler; /* stretch-move ensemble over >= 1 groups */
typedef struct gpemu_fit gpemu_fit;         /* GP fit workspace for one design matrix */

/* ---- library / device ------------------------------------------------------------------- */
const char *gpemu_version(void);
const char *gpemu_last_error(void);
int gpemu_device_count(void); /* number of HIP devices, 0 if none (never an error) */
int gpemu_device_name(int device, char *buf, int64_t buflen);
/* PCI bus id of the device ("0000:c1:00.0"): with the host name it tells whether two ranks of a job share a GPU
 * (gpemu_sampler_peer_share) */
int gpemu_device_bus_id(int device, char *buf, int64_t buflen);
/* free / total device memory in bytes (hipMemGetInfo): the fit sizes its batches within what is free */
int gpemu_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes);

/* ---- model: one emulation group --------------------------------------------------------- */
/* Replaces the per-worker state of ref: log_posterior.py:26-38 (initialize_pool_variables) and
 * the sklearn objects in the results dict of ref: emulation.py:181-192.
 *   X_train[N*d]     design (GaussianProcessRegressor.X_train_)
 *   ls[k*d]          per-PC ARD length scales of kernel_ (not logs)
 *   constv[k]        ConstantKernel value per PC   (read iff has_const)
 *   noise[k]         WhiteKernel noise level per PC (read iff has_noise)
 *   alpha[k*N]       GaussianProcessRegressor.alpha_
 *   L[k*N*N]         GaussianProcessRegressor.L_ (lower; the strict upper triangle is ignored)
 *   components[k*F]  pca.components_[:k]
 *   scaler_mean[F], scaler_scale[F]   StandardScaler
 *   cov_unexplained[F*F]  ref: emulation.py:246-249, or NULL (then 0)
 * The factor is inverted on the device once (W = L^-1) so that prediction is a triangular GEMM.
 */
int gpemu_model_create(gpemu_model **out, int device, int64_t N, int64_t d, int64_t F, int64_t k,
                       int kernel_kind, double nu, int has_const, int has_noise,
                       const double *X_train, const double *ls, const double *constv,
                       const double *noise, const double *alpha, const double *L,
                       const double *components, const double *scaler_mean,
                       const double *scaler_scale, const double *cov_unexplained);
int gpemu_model_destroy(gpemu_model *m);
int gpemu_model_dims(const gpemu_model *m, int64_t *N, int64_t *d, int64_t *F, int64_t *k);
int gpemu_model_device(const gpemu_model *m);
int gpemu_model_sync(gpemu_model *m); /* wait for the handle's stream */
/* Measurement aid for bench.py: when enabled, every launch of the two hot kernels is bracketed by
 * HIP events on the stream it is launched on.  read -> ms_total[2], launches[2]:
 * [0] = trmm_vsq_kernel (triangular GEMM, MFMA f64), [1] = kstar_kernel (cross-kernel build). */
int gpemu_model_profile(gpemu_model *m, int enable);
int gpemu_model_profile_read(gpemu_model *m, double *ms_total, int64_t *launches);

/* ref: emulation.py:494-499 -> skl _gpr.py:441-494 (predict(X, return_std=True), std squared):
 * X[B*d] -> mean_out[B*k], var_out[B*k]  (negative variances clipped to 0, skl _gpr.py:479-485) */
int gpemu_gp_predict(gpemu_model *m, int64_t B, const double *X, double *mean_out, double *var_out);
int gpemu_gp_predict_dev(gpemu_model *m, int64_t B, const double *dX, double *dmean, double *dvar,
                         void *stream);

/* ref: emulation.py:466-548 (predict_emulation_group): central_value[B*F], cov[B*F*F];
 * n_div = the reference's n_samples divisor of the truncation covariance (emulation.py:531-532). */
int gpemu_predict_full(gpemu_model *m, int64_t B, const double *X, double n_div, double *cv_out,
                       double *cov_out);
int gpemu_predict_full_dev(gpemu_model *m, int64_t B, const double *dX, double n_div, double *dcv,
                           double *dcov, void *stream);

/* ref: log_posterior.py:63-64, 73-74, 92-94: box prior + experimental data for this group's
 * features (already gathered into the group's column order).  n_div as above (1 for MCMC).
 * block_start[n_blocks+1]: first feature of every observable of the group (ascending, 0 .. F).
 * The reference's merge keeps only within-observable covariance blocks (ref: emulation.py:370-388),
 * so the likelihood factorises over these blocks.  n_blocks <= 0 or NULL = one block (whole group). */
int gpemu_likelihood_setup(gpemu_model *m, const double *y_exp, const double *y_err,
                           const double *lo, const double *hi, double n_div, int64_t n_blocks,
                           const int64_t *block_start);

/* The same for SEVERAL data vectors against the same uncertainties: the closure tests of ref: steer_analysis.py:168-183
 * condition one chain each on the pseudo-data of one validation point (ref: data_IO.py:362-372).  y_exp[n_chains*F];
 * chain c of a sampler created with gpemu_sampler_create_chains then uses vector c.  Calls outside a multi-chain
 * sampler (gpemu_logpost) use vector 0. */
int gpemu_likelihood_setup_chains(gpemu_model *m, int n_chains, const double *y_exp, const double *y_err,
                                  const double *lo, const double *hi, double n_div, int64_t n_blocks,
                                  const int64_t *block_start);

/* ref: log_posterior.py:42-101 + 104-146: X[B*d] -> out[B]; rows outside the open box -> -inf.
 * A non-positive-definite covariance yields NaN (the reference does not detect it either,
 * log_posterior.py:125-135). */
int gpemu_logpost(gpemu_model *m, int64_t B, const double *X, double *out, int mode);
int gpemu_logpost_dev(gpemu_model *m, int64_t B, const double *dX, double *dout, int mode,
                      void *stream);

/* ---- GP fit: kernel matrix, Cholesky, log-marginal likelihood + gradient -------------------------
 * Replaces the arithmetic inside ref: emulation.py:169-172 (GaussianProcessRegressor(...).fit):
 * skl _gpr.py:537-652 log_marginal_likelihood(theta, eval_gradient=True) and :346-364 (final L_,
 * alpha_).  theta = log([l_1..l_d, (constant_value), (noise_level)]) in sklearn's order
 * (skl kernels.py:733-760, 861-866); jitter = GaussianProcessRegressor(alpha=...).  The optimiser
 * (L-BFGS-B + restarts, skl _gpr.py:299-337) stays on the host and calls gpemu_fit_lml.
 * A non-positive-definite kernel matrix returns > 0 (index + 1 of the failing pivot); sklearn
 * raises LinAlgError there (skl _gpr.py:350-358).
 */
int gpemu_fit_create(gpemu_fit **out, int device, int64_t N, int64_t d, const double *X,
                     int kernel_kind, double nu, int has_const, int has_noise, double jitter);
int gpemu_fit_destroy(gpemu_fit *f);
/* lml and, if grad != NULL, d lml / d theta [n_theta] for target y[N] */
int gpemu_fit_lml(gpemu_fit *f, const double *y, const double *theta, int64_t n_theta, double *lml,
                  double *grad);
/* The same for n_problems (target, theta) pairs AT ONCE: ys[n_problems*N], thetas[n_problems*n_theta] ->
 * lml[n_problems], grad[n_problems*n_theta] (or NULL), info[n_problems] (0, or the failing pivot of a kernel matrix
 * that is not positive definite: that problem's lml / grad are meaningless, the others are valid).  Every launch of the
 * evaluation chain serves the whole batch: the k GPs x (1 + n_restarts) independent maximisations of
 * ref: emulation.py:169-172 (n_restarts: 50 in config/jet_substructure.yaml:80) advance in lock step. */
int gpemu_fit_lml_batch(gpemu_fit *f, int64_t n_problems, const double *ys, const double *thetas, int64_t n_theta,
                        double *lml, double *grad, int32_t *info);
/* L_out[N*N] (lower, zeros above), alpha_out[N], lml at theta; any output may be NULL */
int gpemu_fit_factor(gpemu_fit *f, const double *y, const double *theta, int64_t n_theta,
                     double *L_out, double *alpha_out, double *lml);
/* K_out[N*N] = kernel_(X) (+ jitter on the diagonal): skl kernels.py:1553-1582, 1708-1781 */
int gpemu_kernel_matrix(int device, int64_t N, int64_t d, const double *X, const double *theta,
                        int64_t n_theta, int kernel_kind, double nu, int has_const, int has_noise,
                        double jitter, double *K_out);
/* in-place lower Cholesky of a symmetric positive definite N x N matrix (scipy.linalg.cholesky
 * (lower=True), skl _gpr.py:349): blocked, MFMA f64 SYRK trailing updates */
int gpemu_cholesky(int device, int64_t N, double *A_inout);

/* ---- StandardScaler + PCA ---------------------------------------------------------------------
 * Replaces ref: emulation.py:109-118: scaler.fit_transform(Y) followed by
 * PCA(n_components, svd_solver='full').fit_transform (skl preprocessing/_data.py:1015-1051,
 * decomposition/_pca.py:544-702, svd_flip utils/extmath.py:944-952).  Y[N*F] row-major.
 * n_components <= 0 means min(N, F).  Outputs: scaler mean_/scale_/var_ [F], pca mean_ [F],
 * components_ [nc*F], explained_variance_ / _ratio_ [nc], Y_pca [N*nc] (= U S), flip_argmax [nc]
 * (index of the max-|.| entry of each component row: the svd_flip sign decision), n_sweeps.
 * Environment GPEMU_SVD_FLIP=u (read per call): the u-based decision of the scikit-learn the reference pins (ref:
 * pdm.lock:1998-1999 -> 1.3.0: per column of U; flip_argmax then holds the deciding ROW of U); default v: the rule of
 * scikit-learn >= 1.5 the goldens were made with.  Physical-space outputs do not depend on it. */
int gpemu_pca_fit(int device, int64_t N, int64_t F, const double *Y, int64_t n_components,
                  double *scaler_mean, double *scaler_scale, double *scaler_var, double *pca_mean,
                  double *components, double *explained_variance, double *explained_variance_ratio,
                  double *Y_pca, int64_t *flip_argmax, int64_t *n_sweeps);

/* ---- truncation covariance ----------------------------------------------------------------------
 * Replaces ref: emulation.py:227-251 (compute_emulator_group_cov_unexplained):
 *   cov_out[F*F] = S_{>k} diag(explained_variance_{>k}) S_{>k}^T,  S = components^T,
 * components[n_comp*F] = pca.components_, explained_variance[n_comp], k = n_pc: one F x F x (n_comp - n_pc)
 * product on the f64 matrix cores.  The reference divides it by the batch size at use (emulation.py:531-532);
 * that stays with the callers (n_div). */
int gpemu_truncation_cov(int device, int64_t n_comp, int64_t F, int64_t n_pc, const double *components,
                         const double *explained_variance, double *cov_out);

/* ---- stretch-move ensemble sampler ------------------------------------------------------------
 * Replaces ref: mcmc.py:77-107, 187-204: emcee.EnsembleSampler(n_walkers, ndim, log_posterior,
 * pool=Pool()) with its default StretchMove(a=2) and the pool.map over walkers.  The ensemble,
 * the proposals, the accept/reject and the chain live on the device; the log-posterior is the sum
 * over the given emulation groups (block-diagonal covariance, ref: emulation.py:346-406), each with
 * n_div = 1 (emcee evaluates one walker per call).
 */
int gpemu_sampler_create(gpemu_sampler **out, gpemu_model *const *groups, int n_groups, int64_t W,
                         double a, uint64_t seed);
/* n_chains INDEPENDENT ensembles of W walkers each sharing the groups' emulators -- the reference's closure loop
 * (ref: steer_analysis.py:168-183: one MCMC per validation design point) as one batched run: the proposals of all
 * chains are stacked into the same cross-kernel / triangular-GEMM / likelihood launches.  Chain c draws from its own
 * Philox key seeds[c] and is, bit for bit, the chain gpemu_sampler_create(..., W, a, seeds[c]) produces on the data
 * vector c.  State, chain and counters are laid out chain after chain: walker c W + w. */
int gpemu_sampler_create_chains(gpemu_sampler **out, gpemu_model *const *groups, int n_groups, int64_t W,
                                double a, const uint64_t *seeds, int n_chains);
int gpemu_sampler_destroy(gpemu_sampler *s);
int gpemu_sampler_set_stream(gpemu_sampler *s, void *stream); /* NULL = the first group's stream */
/* X0[W*d]; logp0[W] or NULL to evaluate it (ref: mcmc.py:88, emcee State(initial_state)) */
int gpemu_sampler_set_state(gpemu_sampler *s, const double *X0, const double *logp0);
int gpemu_sampler_get_state(gpemu_sampler *s, double *X, double *logp);
int gpemu_sampler_reset(gpemu_sampler *s); /* emcee sampler.reset(): drop chain + acceptance counts */
/* `steps` full stretch-move steps with device-side Philox randomness; returns 1 if any proposal's
 * log-probability was NaN (emcee raises ValueError). */
int gpemu_sampler_run(gpemu_sampler *s, int64_t steps, int store_chain);
/* One step with host-supplied randomness in emcee's draw order: inds[W] = the shuffled split,
 * then for split 0 (first ceil(W/2) entries) and split 1 (rest): zz = ((a-1)u+1)^2/a, rint in
 * [0, Nc), logu = log(u').  Used to replay a numpy RandomState stream. */
int gpemu_sampler_step_host_rng(gpemu_sampler *s, const int32_t *inds, const double *zz,
                                const int64_t *rint, const double *logu, int store_chain);
/* chain_out[n*W*d] (emcee get_chain()[first:first+n]), logp_out[n*W] (get_log_prob()) */
int gpemu_sampler_get_chain(gpemu_sampler *s, int64_t first, int64_t n, double *chain_out,
                            double *logp_out);
int gpemu_sampler_get_counts(gpemu_sampler *s, int64_t *naccepted /*[W]*/, int64_t *iterations,
                             int64_t *chain_len);
/* Walker-averaged normalised autocorrelation function of the stored chain rows [first, first + n_steps), walkers
 * [w0, w0 + nw) (a chain of a stacked sampler), lags [lag0, lag0 + n_lags): f_out[n_lags][d],
 *     f[l][dd] = 1/nw sum_w acf_(w,dd)[l] / acf_(w,dd)[0],  acf[l] = sum_t (x[t] - mean)(x[t + l] - mean),
 * i.e. emcee.autocorr.function_1d averaged as emcee.autocorr.integrated_time does (emcee 3.1.x, third party; call
 * site ref: mcmc.py:111-119 sampler.get_autocorr_time()).  The host asks for blocks of lags (lag0 a multiple of 16,
 * the first block at lag0 = 0) until Sokal's window closes, so the chain never has to leave HBM for it. */
int gpemu_sampler_acf(gpemu_sampler *s, int64_t first, int64_t n_steps, int64_t w0, int64_t nw, int64_t lag0,
                      int64_t n_lags, double *f_out);
/* Phases of one step for the multi-GPU driver: every rank holds the whole ensemble and draws the
 * same randomness; rank r evaluates proposals [lo, hi) of the half (the log-probabilities land in
 * dnewlp_slice[0 .. hi-lo), typically the caller's all-gather input) and the ranks all-gather them
 * (RCCL, done by the caller on the sampler's stream) before every rank accepts identically; the
 * accept kernels also record the chain row when store_chain is set.  dnewlp_* are DEVICE pointers. */
int gpemu_sampler_reserve_chain(gpemu_sampler *s, int64_t additional_steps); /* grow the chain buffer once, up front */
int gpemu_sampler_begin_step(gpemu_sampler *s);
int gpemu_sampler_half_propose_eval(gpemu_sampler *s, int half, int64_t lo, int64_t hi,
                                    double *dnewlp_slice);
int gpemu_sampler_half_accept(gpemu_sampler *s, int half, const double *dnewlp_all, int store_chain);
int gpemu_sampler_end_step(gpemu_sampler *s, int store_chain);
int gpemu_sampler_check(gpemu_sampler *s); /* synchronise; 1 if a NaN log-probability was seen */

/* ---- RCCL communicator (one rank per GPU) and the sharded run in a single call ----------------
 * Replaces the multiprocessing pool of ref: mcmc.py:77-85 across GPUs.  librccl is bound with dlopen:
 * pass the path of the copy the process already uses (torch.distributed's) or NULL for "librccl.so".
 * Rank 0 obtains a 128-byte id with gpemu_comm_unique_id and ships it to the other ranks by any means
 * (torch.distributed broadcast, a file, MPI); every rank then calls gpemu_comm_create (collective). */
typedef struct gpemu_comm gpemu_comm;
int gpemu_comm_unique_id(const char *librccl_path, char *id_out128);
int gpemu_comm_create(gpemu_comm **out, int device, int rank, int world, const char *id128,
                      const char *librccl_path);
int gpemu_comm_destroy(gpemu_comm *c);
int gpemu_comm_dims(const gpemu_comm *c, int *rank, int *world);
/* all-gather of `count` doubles per rank, DEVICE pointers, enqueued on `stream` */
int gpemu_comm_all_gather(gpemu_comm *c, const double *dsend, double *drecv, int64_t count, void *stream);
/* `steps` stretch-move steps with each half's proposals split over the communicator's ranks (contiguous
 * blocks of ceil(n/world)); one 8-byte-per-proposal all-gather per half-step, no host round trip inside
 * the loop.  Same chain as gpemu_sampler_run on every rank.  emulate_world > 0 (one-rank communicator
 * only): evaluate just the share of rank 0 of an emulate_world-rank job -- a timing aid, not a valid chain; `c` may
 * be NULL then where the fused two-launch half-step applies (GPEMU_ERR_UNSUPPORTED otherwise). */
int gpemu_sampler_run_sharded(gpemu_sampler *s, gpemu_comm *c, int64_t steps, int store_chain,
                              int emulate_world);

/* One-off check of the peer exchange before a chain depends on it: this rank stores a token into every rank's buffer
 * through the pointers of gpemu_sampler_peer_import and waits (bounded, seconds) for all ranks' tokens in its own.
 * Collective in effect: call on every rank at about the same time (after a barrier).  0 = all tokens arrived. */
int gpemu_sampler_peer_selftest(gpemu_sampler *s);

/* ---- sharded run without a collective: peer stores over xGMI ---------------------------------------------
 * Every rank owns a small "gather" buffer; after evaluating its share of a half's proposals a rank stores each new
 * log-probability (8 bytes) straight into every rank's buffer.  Setup: each rank exports a 64-byte IPC handle of its
 * buffer (gpemu_sampler_peer_export), the ranks exchange the handles by any means (torch.distributed all_gather),
 * and each imports all of them (handles[world*64], its own entry is ignored).  gpemu_sampler_run_peer then runs
 * `steps` stretch-move steps with one front launch + one triangular GEMM per emulation group per half-step and no RCCL
 * call; same chain as gpemu_sampler_run on every rank.  Takes up to 8 emulation groups of up to 64 PCs each, one chain,
 * d <= 7 parameters (else GPEMU_ERR_UNSUPPORTED from the export / import: use gpemu_sampler_run_sharded).  The ranks must enter every call
 * together (a barrier on the host side): a peer's stores may arrive as soon as it has started.  A peer that does not
 * deliver within GPEMU_PEER_TIMEOUT_MS (5000) ends the run with GPEMU_ERR_STATE on every rank.
 * Replaces ref: mcmc.py:77-85 (the pool.map over walkers). */
int gpemu_sampler_peer_export(gpemu_sampler *s, char *handle_out64);
/* How many ranks of the job run their samplers on THIS device (default 1: one process per GPU, the production layout of
 * ref: mcmc.py:77-85's pool).  Call before gpemu_sampler_peer_import.  With one rank per device a fused launch needs no
 * residency rule (its waits only target workgroups dispatched before the waiters); with several ranks on one device
 * (one-GPU rehearsals) all their launches must be resident together, which the import then checks against the runtime's
 * occupancy figure -- failing with GPEMU_ERR_UNSUPPORTED (fall back to gpemu_sampler_run_sharded) instead of a time-out. */
int gpemu_sampler_peer_share(gpemu_sampler *s, int ranks_on_device);
int gpemu_sampler_peer_import(gpemu_sampler *s, int world, int rank, const char *handles);
int gpemu_sampler_run_peer(gpemu_sampler *s, int64_t steps, int store_chain);

/* Snapshot / restore of the chain state on the device (ensemble, log-probabilities, acceptance counters, step and
 * chain counters).  A block of steps whose peer exchange was lost (GPEMU_ERR_STATE from gpemu_sampler_run_peer) is
 * rerun from the snapshot over a collective transport: the random stream is counter based, so the rerun draws what the
 * failed attempt drew and the chain is that of an unbroken run.  Replaces nothing in the reference (its pool has no
 * recovery: ref: mcmc.py:77-107); part of the sharded run that replaces the pool. */
int gpemu_sampler_snapshot(gpemu_sampler *s);
int gpemu_sampler_restore(gpemu_sampler *s);

/* Launches so far, in this process, of the one-launch cross-kernel + triangular GEMM for small emulators (at most 256
 * design points and 32 PCs per group, at most 128 rows: csrc/k_halfstep.hip; what a sampler's half-step and a small
 * batched log-posterior take there).  For tests: which path ran.  GPEMU_NO_HALFSTEP=1 switches that path off. */
int64_t gpemu_halfstep_small_launches(void);

/* Philox4x32-10 block function (host copy of the device generator; for tests) */
int gpemu_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                     uint32_t *out4);

#ifdef __cplusplus
}
#endif
#endif /* GPEMU_H */
