// C ABI of libgpemu.so (see include/gpemu.h).  Host-side glue only: argument checks, device
// memory ownership, launch sequencing.  All arithmetic is in the k_*.hip kernels.
#include <cmath>
#include <cstdarg>

#include "internal.h"
#include "kstar_host.h"
#include "gemm.h"
#include "linalg_dev.h"

namespace gpemu {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int launch_lik_setup(gpemu_model *m, const std::vector<int> &hstart, double *dA, double *dPT, double *dZ, int *dinfo,
                     hipStream_t st);
int launch_predict_full(gpemu_model *m, int64_t B, double n_div, double *dcv, double *dcov, hipStream_t st,
                        const double *dmean, const double *dvar);
int launch_loglik_exact(gpemu_model *m, int64_t B, const double *dXq, double *dout, hipStream_t st);

template <typename T>
static int dev_alloc(T **p, int64_t n) {
  *p = nullptr;
  if (n <= 0) n = 1;
  GP_HIP(hipMalloc((void **)p, sizeof(T) * (size_t)n));
  return GPEMU_OK;
}
#define GP_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != GPEMU_OK) return rc__; \
  } while (0)

static int upload(double *dst, const double *src, int64_t n, hipStream_t st) {
  GP_HIP(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
  return GPEMU_OK;
}

static void free_workspace(Workspace &w) {
  hipFree(w.Xq); hipFree(w.KS); hipFree(w.mean_part); hipFree(w.mean_part2); hipFree(w.vsq_part);
  hipFree(w.mean); hipFree(w.var); hipFree(w.logp);
  w = Workspace();
}

int ensure_workspace(gpemu_model *m, int64_t B) {
  Workspace &w = m->ws;
  int64_t need = round_up(B < 1 ? 1 : B, TILE);
  if (need <= w.Bcap) return GPEMU_OK;
  GP_HIP(hipStreamSynchronize(m->stream));
  free_workspace(w);
  const int64_t k = m->k;
  GP_TRY(dev_alloc(&w.Xq, need * DPAD));
  GP_TRY(dev_alloc(&w.KS, k * m->Npad * need));
  GP_TRY(dev_alloc(&w.mean_part, k * (m->Npad / 32) * need));   // sized for the 32-row small-batch form
  GP_TRY(dev_alloc(&w.mean_part2, k * (m->Npad / 32) * need));
  GP_TRY(dev_alloc(&w.vsq_part, k * (m->Npad / 32) * need));   // sized for the 32-row small-batch form
  GP_TRY(dev_alloc(&w.mean, need * k));
  GP_TRY(dev_alloc(&w.var, need * k));
  GP_TRY(dev_alloc(&w.logp, need));
  GP_HIP(hipMemsetAsync(w.KS, 0, sizeof(double) * (size_t)(k * m->Npad * need), m->stream));
  GP_HIP(hipMemsetAsync(w.Xq, 0, sizeof(double) * (size_t)(need * DPAD), m->stream));
  GP_HIP(hipMemsetAsync(w.vsq_part, 0, sizeof(double) * (size_t)(k * (m->Npad / 32) * need), m->stream));
  // the caller's launches may go to ANOTHER stream (a sampler over several groups runs every group on the first
  // group's stream): the zero fill must have landed before anything writes the new buffers (found by the shipped
  // three-group golden G7: the fill of groups 2 and 3 raced with their first evaluation and wiped partial sums)
  GP_HIP(hipStreamSynchronize(m->stream));
  w.Bcap = need;
  return GPEMU_OK;
}

// ---- optional per-kernel timing ---------------------------------------------------------------
static void prof_drain(gpemu_model *m) {
  // all recorded events must have completed (callers synchronise the stream first)
  for (int which = 0; which < 2; ++which) {
    auto &v = which == 0 ? m->ev_trmm : m->ev_kstar;
    for (auto &pr : v) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, m->ev_pool[pr.first], m->ev_pool[pr.second]) == hipSuccess) {
        m->prof_ms[which] += ms;
        m->prof_n[which] += 1;
      }
    }
    v.clear();
  }
  m->ev_next = 0;
}

int prof_mark(gpemu_model *m, hipStream_t st) {
  if (!m->profiling) return -1;
  if (m->ev_next >= 8192) {  // bounded pool: fold what we have into the totals and reuse
    (void)hipStreamSynchronize(st);
    prof_drain(m);
  }
  if (m->ev_next >= m->ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return -1;
    m->ev_pool.push_back(e);
  }
  int idx = (int)m->ev_next++;
  if (hipEventRecord(m->ev_pool[idx], st) != hipSuccess) return -1;
  return idx;
}

void prof_pair(gpemu_model *m, int which, int e0, int e1) {
  if (!m->profiling || e0 < 0 || e1 < 0) return;
  (which == 0 ? m->ev_trmm : m->ev_kstar).emplace_back(e0, e1);
}

static int check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device available: libgpemu has no CPU implementation");
    return GPEMU_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range (have %d)", device, n);
    return GPEMU_ERR_ARG;
  }
  return GPEMU_OK;
}

}  // namespace gpemu

namespace gpemu {
// Log-posterior of B query rows already in the padded [rows >= round_up(B,128)][DPAD] layout
// (the sampler writes its proposals in that layout).  accumulate != 0 adds to dout (multi-group).
int logpost_padded(gpemu_model *m, int64_t B, double *dXq, double *dout, int accumulate,
                   hipStream_t st, const AcceptArgs *aa, const ProposeArgs *pa) {
  if (!m->lik_ready) { set_error("gpemu_likelihood_setup has not been called"); return GPEMU_ERR_STATE; }
  int rc = ensure_workspace(m, B);
  if (rc != GPEMU_OK) return rc;
  // small emulators: cross-kernel and GEMM in one launch (k_halfstep.hip; the same bits), else the two general launches
  rc = GPEMU_ERR_UNSUPPORTED;
  if (getenv("GPEMU_NO_HALFSTEP") == nullptr && getenv("GPEMU_NO_GROUP_MERGE") == nullptr && !(aa && aa->chain_per != 0)) {
    gpemu_model *one[1] = {m};
    rc = launch_halfstep_small(one, 1, B, dXq, st, pa);
  }
  if (rc == GPEMU_ERR_UNSUPPORTED) {
    rc = launch_kstar(m, B, dXq, st, pa);
    if (rc == GPEMU_OK) rc = launch_trmm_vsq(m, B, st);
  }
  if (rc != GPEMU_OK) return rc;
  return launch_loglik_lowrank(m, B, dXq, dout, accumulate, st, aa);
}

// Log-posterior of B padded query rows summed over ng >= 2 emulation groups with ONE launch per stage (cross-kernels,
// triangular GEMMs, likelihoods + accept).  A sampler over the reference's shipped three groups (5 / 11 / 25 PCs, ~150
// design points, 200 walkers) spends its half-step in nine ~6 us launches otherwise; the kernels' work per group is the
// per-group launches' (same device functions, the groups' terms added in the same order): the same bits.
// Eligible: at most 128 columns (the small-batch GEMM) evaluated as one chain, at most 32 PCs per group, one base kernel
// and parameter count.  GPEMU_ERR_UNSUPPORTED otherwise (nothing launched): the caller goes group by group.
int logpost_groups(gpemu_model *const *ms, int ng, int64_t B, double *dXq, double *dout, hipStream_t st,
                   const AcceptArgs *aa, const ProposeArgs *pa) {
  const bool off = getenv("GPEMU_NO_GROUP_MERGE") != nullptr;     // (tests: the per-group launches; read per call)
  if (off || ng < 2 || ng > 8 || B < 1 || B > 128) return GPEMU_ERR_UNSUPPORTED;
  if (aa && aa->chain_per != 0) return GPEMU_ERR_UNSUPPORTED;
  for (int g = 0; g < ng; ++g) {
    const gpemu_model *m = ms[g];
    if (!m->lik_ready) { set_error("gpemu_likelihood_setup has not been called"); return GPEMU_ERR_STATE; }
    const int64_t Bv = m->variant_B > 0 ? m->variant_B : B;
    if (Bv > 128 || m->k > 32 || m->d != ms[0]->d || kstar_kind(m) != kstar_kind(ms[0]) || m->ksteps != ms[0]->ksteps ||
        m->device != ms[0]->device || m->profiling)
      return GPEMU_ERR_UNSUPPORTED;
  }
  for (int g = 0; g < ng; ++g) {
    const int rc = ensure_workspace(ms[g], B);
    if (rc != GPEMU_OK) return rc;
  }
  // (the GEMM's schedules first: the one step that can still say "unsupported", before anything is launched)
  int rc = prepare_trmm_vsq_small_groups(ms, ng, B, st);
  if (rc != GPEMU_OK) return rc;
  rc = launch_kstar_groups(ms, ng, B, dXq, st, pa);
  if (rc == GPEMU_OK) rc = launch_trmm_vsq_small_groups(ms, ng, B, st);
  if (rc != GPEMU_OK) return rc;
  return launch_loglik_groups(ms, ng, B, dXq, dout, 0, st, aa);
}

// Log-posterior of B <= 128 padded query rows over ng >= 1 groups of small emulators (at most 256 design points, 32 PCs
// each): cross-kernel and triangular GEMM of every group in ONE launch (k_halfstep.hip), then the likelihood launch of
// the general path on single partial sums.  The same bits as the general path.  GPEMU_NO_HALFSTEP (or
// GPEMU_NO_GROUP_MERGE: "stage by stage, group by group") switches it off (tests; read per call).
int logpost_small(gpemu_model *const *ms, int ng, int64_t B, double *dXq, double *dout, hipStream_t st,
                  const AcceptArgs *aa, const ProposeArgs *pa) {
  if (getenv("GPEMU_NO_HALFSTEP") != nullptr || getenv("GPEMU_NO_GROUP_MERGE") != nullptr) return GPEMU_ERR_UNSUPPORTED;
  if (ng < 1 || ng > 8 || B < 1 || B > 128 || (aa && aa->chain_per != 0)) return GPEMU_ERR_UNSUPPORTED;
  for (int g = 0; g < ng; ++g) {
    const gpemu_model *m = ms[g];
    if (!m->lik_ready) { set_error("gpemu_likelihood_setup has not been called"); return GPEMU_ERR_STATE; }
    if (m->Npad > 256 || m->k > 32) return GPEMU_ERR_UNSUPPORTED;
  }
  for (int g = 0; g < ng; ++g) {
    const int rc = ensure_workspace(ms[g], B);
    if (rc != GPEMU_OK) return rc;
  }
  const int rc = launch_halfstep_small(ms, ng, B, dXq, st, pa);
  if (rc != GPEMU_OK) return rc;
  if (ng == 1) return launch_loglik_lowrank(ms[0], B, dXq, dout, 0, st, aa);
  return launch_loglik_groups(ms, ng, B, dXq, dout, 0, st, aa);
}
}  // namespace gpemu

using namespace gpemu;

extern "C" {

const char *gpemu_version(void) { return "gpemu 0.1 (gfx950)"; }
const char *gpemu_last_error(void) { return g_err; }

int gpemu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int gpemu_device_name(int device, char *buf, int64_t buflen) {
  GP_TRY(check_device(device));
  GP_ARG(buf && buflen > 0, "buf");
  hipDeviceProp_t prop;
  GP_HIP(hipGetDeviceProperties(&prop, device));
  snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return GPEMU_OK;
}

int gpemu_device_bus_id(int device, char *buf, int64_t buflen) {
  GP_TRY(check_device(device));
  GP_ARG(buf && buflen >= 16, "buf");
  GP_HIP(hipDeviceGetPCIBusId(buf, (int)buflen, device));
  return GPEMU_OK;
}

int gpemu_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes) {
  GP_TRY(check_device(device));
  GP_ARG(free_bytes && total_bytes, "null pointer");
  GP_HIP(hipSetDevice(device));
  size_t f = 0, t = 0;
  GP_HIP(hipMemGetInfo(&f, &t));
  *free_bytes = (int64_t)f;
  *total_bytes = (int64_t)t;
  return GPEMU_OK;
}

int gpemu_model_create(gpemu_model **out, int device, int64_t N, int64_t d, int64_t F, int64_t k,
                       int kernel_kind, double nu, int has_const, int has_noise,
                       const double *X_train, const double *ls, const double *constv,
                       const double *noise, const double *alpha, const double *L,
                       const double *components, const double *scaler_mean,
                       const double *scaler_scale, const double *cov_unexplained) {
  GP_ARG(out, "out");
  *out = nullptr;
  GP_ARG(N > 0 && d > 0 && F > 0 && k > 0, "N, d, F, k must be positive");
  GP_ARG(d <= DPAD, "d > 8 parameters is not supported by this build");
  GP_ARG(k <= 64, "k > 64 principal components is not supported by this build");
  GP_ARG(kernel_kind == GPEMU_KERNEL_RBF || kernel_kind == GPEMU_KERNEL_MATERN, "kernel_kind");
  if (kernel_kind == GPEMU_KERNEL_MATERN)
    GP_ARG(nu == 0.5 || nu == 1.5 || nu == 2.5, "Matern nu must be 0.5, 1.5 or 2.5");
  GP_ARG(X_train && ls && alpha && L && components && scaler_mean && scaler_scale, "null array");
  GP_ARG(!has_const || constv, "constv");
  GP_ARG(!has_noise || noise, "noise");
  GP_TRY(check_device(device));
  GP_HIP(hipSetDevice(device));

  gpemu_model *m = new gpemu_model();
  m->device = device;
  m->N = N; m->d = d; m->F = F; m->k = k;
  m->Npad = round_up(N, TILE);
  m->vsq_nrb = m->Npad / 64;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      m->num_cu = prop.multiProcessorCount;
  }
  m->kernel_kind = kernel_kind; m->nu = nu;
  m->has_const = has_const ? 1 : 0; m->has_noise = has_noise ? 1 : 0;
  const int64_t Np = m->Npad;
  int rc = GPEMU_OK;
  double *dL = nullptr;
  auto fail = [&](int code) {
    hipFree(dL);
    gpemu_model_destroy(m);
    return code;
  };
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("hipStreamCreate failed");
    return fail(GPEMU_ERR_HIP);
  }
  hipStream_t st = m->stream;

  // host staging of the small padded arrays
  const bool matern05 = kernel_kind == GPEMU_KERNEL_MATERN && nu == 0.5;
  std::vector<double> hXs(matern05 ? (size_t)(k * Np * DPAD) : 0, 0.0), hls((size_t)(k * DPAD), 1.0),
      hc((size_t)k, 0.0), hkd((size_t)k, 1.0), hal((size_t)(k * Np), 0.0);
  for (int64_t p = 0; p < k; ++p) {
    for (int64_t dd = 0; dd < d; ++dd) {
      double l = ls[p * d + dd];
      if (!(l > 0.0)) { set_error("length scale must be positive"); return fail(GPEMU_ERR_ARG); }
      hls[p * DPAD + dd] = l;
    }
    if (matern05)
      for (int64_t j = 0; j < N; ++j)
        for (int64_t dd = 0; dd < d; ++dd)
          hXs[(p * Np + j) * DPAD + dd] = X_train[j * d + dd] / ls[p * d + dd];  // skl: X / length_scale
    if (has_const) { hc[p] = constv[p]; hkd[p] += constv[p]; }
    if (has_noise) hkd[p] += noise[p];
    for (int64_t j = 0; j < N; ++j) hal[p * Np + j] = alpha[p * N + j];
  }
#define GP_STEP(expr) if ((rc = (expr)) != GPEMU_OK) return fail(rc)
  GP_STEP(dev_alloc(&m->ls, k * DPAD));
  GP_STEP(dev_alloc(&m->constv, k));
  GP_STEP(dev_alloc(&m->kdiag, k));
  GP_STEP(dev_alloc(&m->alpha, k * Np));
  GP_STEP(dev_alloc(&m->Wt, k * Np * Np));
  GP_STEP(dev_alloc(&m->comp, k * F));
  GP_STEP(dev_alloc(&m->smean, F));
  GP_STEP(dev_alloc(&m->sscale, F));
  GP_STEP(dev_alloc(&m->cunexpl, F * F));
  GP_STEP(dev_alloc(&dL, k * N * N));
  GP_STEP(upload(m->ls, hls.data(), k * DPAD, st));
  {
    // the cross-kernel's operands for the matrix cores (kstar_host.h)
    KstarHost kh;
    build_kstar_operands(N, Np, d, k, kstar_kind(m), X_train, ls, alpha, kh);
    m->ksteps = kh.ksteps;
    GP_STEP(dev_alloc(&m->Xa, (int64_t)kh.Xa.size()));
    GP_STEP(dev_alloc(&m->alf, (int64_t)kh.alf.size()));
    GP_STEP(dev_alloc(&m->qsc, (int64_t)kh.qsc.size()));
    GP_STEP(dev_alloc(&m->qof, (int64_t)kh.qof.size()));
    GP_STEP(dev_alloc(&m->etab, (int64_t)kh.tab.size()));
    GP_STEP(upload(m->Xa, kh.Xa.data(), (int64_t)kh.Xa.size(), st));
    GP_STEP(upload(m->alf, kh.alf.data(), (int64_t)kh.alf.size(), st));
    GP_STEP(upload(m->qsc, kh.qsc.data(), (int64_t)kh.qsc.size(), st));
    GP_STEP(upload(m->qof, kh.qof.data(), (int64_t)kh.qof.size(), st));
    GP_STEP(upload(m->etab, kh.tab.data(), (int64_t)kh.tab.size(), st));
    std::vector<double> hinv(hls.size());
    if (matern05) {
      for (size_t i = 0; i < hls.size(); ++i) hinv[i] = 1.0 / hls[i];
      GP_STEP(dev_alloc(&m->Xs, k * Np * DPAD));
      GP_STEP(dev_alloc(&m->inv_ls, k * DPAD));
      GP_STEP(upload(m->Xs, hXs.data(), k * Np * DPAD, st));
      GP_STEP(upload(m->inv_ls, hinv.data(), k * DPAD, st));
    }
    if (hipStreamSynchronize(st) != hipSuccess) {   // the staging vectors go out of scope
      set_error("model_create: upload failed");
      return fail(GPEMU_ERR_HIP);
    }
  }
  GP_STEP(upload(m->constv, hc.data(), k, st));
  GP_STEP(upload(m->kdiag, hkd.data(), k, st));
  GP_STEP(upload(m->alpha, hal.data(), k * Np, st));
  GP_STEP(upload(m->comp, components, k * F, st));
  GP_STEP(upload(m->smean, scaler_mean, F, st));
  GP_STEP(upload(m->sscale, scaler_scale, F, st));
  if (cov_unexplained) {
    GP_STEP(upload(m->cunexpl, cov_unexplained, F * F, st));
  } else if (hipMemsetAsync(m->cunexpl, 0, sizeof(double) * (size_t)(F * F), st) != hipSuccess) {
    set_error("hipMemsetAsync failed");
    return fail(GPEMU_ERR_HIP);
  }
  GP_STEP(upload(dL, L, k * N * N, st));
  {
    // W_p = L_p^-1 by the blocked MFMA triangular inverse, written transposed into Wt[p]
    const int64_t N64 = round_up(N, 64);
    double *sA = nullptr, *sD = nullptr, *sW = nullptr, *sT = nullptr;
    rc = dev_alloc(&sA, N64 * N64);
    if (rc == GPEMU_OK) rc = dev_alloc(&sD, N64 * 64);
    if (rc == GPEMU_OK) rc = dev_alloc(&sW, N64 * N64);
    if (rc == GPEMU_OK) rc = dev_alloc(&sT, N64 * N64);
    if (rc == GPEMU_OK && hipMemsetAsync(m->Wt, 0, sizeof(double) * (size_t)(k * Np * Np), st) != hipSuccess) {
      set_error("hipMemsetAsync failed");
      rc = GPEMU_ERR_HIP;
    }
    for (int64_t p = 0; p < k && rc == GPEMU_OK; ++p)
      rc = device_invert_factor_to_Wt(dL + p * N * N, N, m->Wt + p * Np * Np, Np, sA, sD, sW, sT, st);
    if (rc == GPEMU_OK && hipStreamSynchronize(st) != hipSuccess) {
      set_error("model_create: triangular inverse failed: %s", hipGetErrorString(hipGetLastError()));
      rc = GPEMU_ERR_HIP;
    }
    hipFree(sA); hipFree(sD); hipFree(sW); hipFree(sT);
    if (rc != GPEMU_OK) return fail(rc);
  }
#undef GP_STEP
  if (hipStreamSynchronize(st) != hipSuccess) {
    set_error("model_create: device synchronisation failed: %s", hipGetErrorString(hipGetLastError()));
    return fail(GPEMU_ERR_HIP);
  }
  hipFree(dL);
  *out = m;
  return GPEMU_OK;
}

int gpemu_model_destroy(gpemu_model *m) {
  if (!m) return GPEMU_OK;
  hipSetDevice(m->device);
  if (m->stream) hipStreamSynchronize(m->stream);
  hipFree(m->Xs); hipFree(m->inv_ls); hipFree(m->ls); hipFree(m->Xa); hipFree(m->alf); hipFree(m->qsc); hipFree(m->qof);
  hipFree(m->etab); hipFree(m->constv); hipFree(m->kdiag);
  hipFree(m->alpha); hipFree(m->Wt); hipFree(m->comp); hipFree(m->smean); hipFree(m->sscale);
  hipFree(m->cunexpl); hipFree(m->yexp); hipFree(m->yerr); hipFree(m->lo); hipFree(m->hi);
  for (const gpemu_model::LikEntry &en : m->lik_cache) { hipFree(en.G); hipFree(en.g0); hipFree(en.scal); }
  hipFree(m->exact_scratch);
  hipFree(m->blk_start); hipFree(m->blk_of);
  for (const gpemu_model::SchedEntry &en : m->sched_cache) { hipFree(en.items); hipFree(en.cnt); }
  for (const gpemu_model::SchedEntry &en : m->sm_cache) { hipFree(en.items); hipFree(en.cnt); }
  (void)hipFree(m->lik_terms);
  (void)hipFree(m->lik_tickets);
  free_workspace(m->ws);
  for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
  if (m->stream) hipStreamDestroy(m->stream);
  delete m;
  return GPEMU_OK;
}

int gpemu_model_profile(gpemu_model *m, int enable) {
  GP_ARG(m, "model");
  GP_HIP(hipSetDevice(m->device));
  GP_HIP(hipDeviceSynchronize());
  prof_drain(m);
  m->profiling = enable != 0;
  m->prof_ms[0] = m->prof_ms[1] = 0.0;
  m->prof_n[0] = m->prof_n[1] = 0;
  return GPEMU_OK;
}

int gpemu_model_profile_read(gpemu_model *m, double *ms_total, int64_t *launches) {
  GP_ARG(m && ms_total && launches, "null pointer");
  GP_HIP(hipSetDevice(m->device));
  GP_HIP(hipDeviceSynchronize());
  prof_drain(m);
  for (int i = 0; i < 2; ++i) { ms_total[i] = m->prof_ms[i]; launches[i] = m->prof_n[i]; }
  return GPEMU_OK;
}

int gpemu_model_dims(const gpemu_model *m, int64_t *N, int64_t *d, int64_t *F, int64_t *k) {
  GP_ARG(m, "model");
  if (N) *N = m->N;
  if (d) *d = m->d;
  if (F) *F = m->F;
  if (k) *k = m->k;
  return GPEMU_OK;
}

int gpemu_model_device(const gpemu_model *m) { return m ? m->device : GPEMU_ERR_ARG; }

int gpemu_model_sync(gpemu_model *m) {
  GP_ARG(m, "model");
  GP_HIP(hipSetDevice(m->device));
  GP_HIP(hipStreamSynchronize(m->stream));
  return GPEMU_OK;
}

// ---- GP predict --------------------------------------------------------------------------------
constexpr int64_t MAX_CHUNK = 2048;   // rows per pass of the predict pipeline (bounds the K_* workspace)

// small_ok: the caller sums the partials in the likelihood's order (walker_mean_sd), which is the order in which the
// one-launch form for small emulators (k_halfstep.hip) has summed them already
static int gp_predict_core(gpemu_model *m, int64_t B, const double *dX, hipStream_t st, bool small_ok = false) {
  GP_TRY(ensure_workspace(m, B));
  ProposeArgs raw;                 // the cross-kernel kernel pads the caller's rows itself
  raw.raw = dX; raw.n = (int)B; raw.d = (int)m->d;
  if (small_ok && getenv("GPEMU_NO_HALFSTEP") == nullptr && getenv("GPEMU_NO_GROUP_MERGE") == nullptr) {
    gpemu_model *one[1] = {m};
    const int rc = launch_halfstep_small(one, 1, B, m->ws.Xq, st, &raw);
    if (rc != GPEMU_ERR_UNSUPPORTED) return rc;
  }
  GP_TRY(launch_kstar(m, B, m->ws.Xq, st, &raw));
  GP_TRY(launch_trmm_vsq(m, B, st));
  return GPEMU_OK;
}


int gpemu_gp_predict_dev(gpemu_model *m, int64_t B, const double *dX, double *dmean, double *dvar,
                         void *stream) {
  GP_ARG(m && dX && dmean && dvar, "null pointer");
  GP_ARG(B > 0, "B must be positive");
  GP_HIP(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  for (int64_t off = 0; off < B; off += MAX_CHUNK) {   // large batches go through in chunks
    const int64_t nb = (B - off < MAX_CHUNK) ? (B - off) : MAX_CHUNK;
    GP_TRY(gp_predict_core(m, nb, dX + off * m->d, st));
    GP_TRY(launch_reduce_mean_var(m, nb, dmean + off * m->k, dvar + off * m->k, st));
  }
  return GPEMU_OK;
}

int gpemu_gp_predict(gpemu_model *m, int64_t B, const double *X, double *mean_out, double *var_out) {
  GP_ARG(m && X && mean_out && var_out, "null pointer");
  GP_ARG(B > 0, "B must be positive");
  GP_HIP(hipSetDevice(m->device));
  hipStream_t st = m->stream;
  double *dX = nullptr, *dm = nullptr, *dv = nullptr;
  int rc = dev_alloc(&dX, B * m->d);
  if (rc == GPEMU_OK) rc = dev_alloc(&dm, B * m->k);
  if (rc == GPEMU_OK) rc = dev_alloc(&dv, B * m->k);
  if (rc == GPEMU_OK) rc = upload(dX, X, B * m->d, st);
  if (rc == GPEMU_OK) rc = gpemu_gp_predict_dev(m, B, dX, dm, dv, st);
  if (rc == GPEMU_OK) {
    hipError_t e = hipMemcpyAsync(mean_out, dm, sizeof(double) * B * m->k, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(var_out, dv, sizeof(double) * B * m->k, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("gp_predict: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  }
  hipFree(dX); hipFree(dm); hipFree(dv);
  return rc;
}

// ---- likelihood ----------------------------------------------------------------------------------
int gpemu_likelihood_setup(gpemu_model *m, const double *y_exp, const double *y_err,
                           const double *lo, const double *hi, double n_div, int64_t n_blocks,
                           const int64_t *block_start) {
  return gpemu_likelihood_setup_chains(m, 1, y_exp, y_err, lo, hi, n_div, n_blocks, block_start);
}

int gpemu_likelihood_setup_chains(gpemu_model *m, int n_chains, const double *y_exp, const double *y_err,
                                  const double *lo, const double *hi, double n_div, int64_t n_blocks,
                                  const int64_t *block_start) {
  GP_ARG(m && y_exp && y_err && lo && hi, "null pointer");
  GP_ARG(n_div >= 1.0, "n_div must be >= 1");
  GP_ARG(n_chains >= 1 && n_chains <= 4096, "n_chains must be in [1, 4096]");
  const int64_t NC = n_chains;
  GP_HIP(hipSetDevice(m->device));
  hipStream_t st = m->stream;
  const int64_t F = m->F, k = m->k;
  // observable blocks (default: one block = the whole group)
  std::vector<int> hstart, hof((size_t)F, 0);
  if (n_blocks <= 0 || !block_start) {
    hstart = {0, (int)F};
  } else {
    GP_ARG(block_start[0] == 0 && block_start[n_blocks] == F, "block_start must cover [0, F]");
    for (int64_t o = 0; o <= n_blocks; ++o) {
      if (o > 0) GP_ARG(block_start[o] > block_start[o - 1], "block_start must be increasing");
      hstart.push_back((int)block_start[o]);
    }
  }
  const int64_t nblk = (int64_t)hstart.size() - 1;
  for (int64_t o = 0; o < nblk; ++o)
    for (int f = hstart[o]; f < hstart[o + 1]; ++f) hof[f] = (int)o;
  // same data as the cached constants belong to?  then an n_div seen before is a pointer swap
  std::vector<double> key;
  key.push_back((double)NC);
  key.insert(key.end(), y_exp, y_exp + NC * F);
  key.insert(key.end(), y_err, y_err + F);
  key.insert(key.end(), lo, lo + m->d);
  key.insert(key.end(), hi, hi + m->d);
  for (int v : hstart) key.push_back((double)v);
  const bool same_data = m->lik_ready && key.size() == m->lik_host.size() &&
                         memcmp(key.data(), m->lik_host.data(), sizeof(double) * key.size()) == 0;
  if (same_data) {
    for (const gpemu_model::LikEntry &en : m->lik_cache)
      if (en.n_div == n_div) {
        GP_HIP(hipStreamSynchronize(st));
        m->G = en.G; m->g0 = en.g0; m->scal = en.scal; m->n_div = n_div;
        return GPEMU_OK;
      }
  } else {
    GP_HIP(hipStreamSynchronize(st));
    for (const gpemu_model::LikEntry &en : m->lik_cache) { (void)hipFree(en.G); (void)hipFree(en.g0); (void)hipFree(en.scal); }
    m->lik_cache.clear();
    m->G = m->g0 = m->scal = nullptr;
  }
  if (m->lik_cache.size() >= 64) {           // bounded: drop the oldest entry
    GP_HIP(hipStreamSynchronize(st));
    const gpemu_model::LikEntry en = m->lik_cache.front();
    (void)hipFree(en.G); (void)hipFree(en.g0); (void)hipFree(en.scal);
    m->lik_cache.erase(m->lik_cache.begin());
  }
  GP_HIP(hipStreamSynchronize(st));
  if (!m->yerr) {
    GP_TRY(dev_alloc(&m->yerr, F));
    GP_TRY(dev_alloc(&m->lo, DPAD)); GP_TRY(dev_alloc(&m->hi, DPAD));
    GP_TRY(dev_alloc(&m->blk_of, F));
  }
  (void)hipFree(m->yexp);
  m->yexp = nullptr;
  GP_TRY(dev_alloc(&m->yexp, NC * F));
  m->lik_chains = n_chains;
  (void)hipFree(m->blk_start);
  m->blk_start = nullptr;
  gpemu_model::LikEntry en{n_div, nullptr, nullptr, nullptr};
  GP_TRY(dev_alloc(&en.G, nblk * k * k)); GP_TRY(dev_alloc(&en.g0, NC * nblk * k));
  GP_TRY(dev_alloc(&en.scal, NC * 2 * nblk)); GP_TRY(dev_alloc(&m->blk_start, nblk + 1));
  m->G = en.G; m->g0 = en.g0; m->scal = en.scal;
  m->lik_cache.push_back(en);
  m->nblk = nblk;
  GP_HIP(hipMemcpyAsync(m->blk_start, hstart.data(), sizeof(int) * (nblk + 1), hipMemcpyHostToDevice, st));
  GP_HIP(hipMemcpyAsync(m->blk_of, hof.data(), sizeof(int) * F, hipMemcpyHostToDevice, st));
  m->lik_ready = false;
  m->lik_host.clear();
  m->n_div = n_div;
  double hlo[DPAD], hhi[DPAD];
  for (int i = 0; i < DPAD; ++i) { hlo[i] = i < m->d ? lo[i] : -INFINITY; hhi[i] = i < m->d ? hi[i] : INFINITY; }
  GP_TRY(upload(m->yexp, y_exp, NC * F, st)); GP_TRY(upload(m->yerr, y_err, F, st));
  GP_TRY(upload(m->lo, hlo, DPAD, st)); GP_TRY(upload(m->hi, hhi, DPAD, st));
  GP_HIP(hipStreamSynchronize(st));  // hlo/hhi are stack buffers
  double *dA = nullptr, *dPT = nullptr, *dZ = nullptr;
  int *dinfo = nullptr;
  int rc = dev_alloc(&dA, F * F);
  if (rc == GPEMU_OK) rc = dev_alloc(&dPT, nblk * chol_scratch_size(F));
  if (rc == GPEMU_OK) rc = dev_alloc(&dZ, F * (k + NC));
  if (rc == GPEMU_OK) rc = dev_alloc(&dinfo, nblk);
  std::vector<int> info((size_t)nblk, 0);
  if (rc == GPEMU_OK && hipMemsetAsync(dinfo, 0, sizeof(int) * nblk, st) != hipSuccess) rc = GPEMU_ERR_HIP;
  if (rc == GPEMU_OK) rc = launch_lik_setup(m, hstart, dA, dPT, dZ, dinfo, st);
  if (rc == GPEMU_OK) {
    hipError_t e = hipMemcpyAsync(info.data(), dinfo, sizeof(int) * nblk, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("likelihood_setup: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  }
  hipFree(dA); hipFree(dPT); hipFree(dZ); hipFree(dinfo);
  if (rc != GPEMU_OK) return rc;
  for (int64_t o = 0; o < nblk; ++o) {
    if (info[o] != 0) {
      set_error("likelihood_setup: A = C_unexpl/n o ss^T + diag(y_err^2) is not positive definite "
                "(observable block %lld, pivot %d)", (long long)o, info[o]);
      return hstart[o] + info[o];
    }
  }
  m->lik_ready = true;
  m->lik_host = key;
  return GPEMU_OK;
}

int gpemu_logpost_dev(gpemu_model *m, int64_t B, const double *dX, double *dout, int mode, void *stream) {
  GP_ARG(m, "null pointer");
  GP_ARG(B >= 0, "B must not be negative");
  if (B == 0) return GPEMU_OK;   /* empty batch -> empty result (ref: log_posterior.py:59,68) */
  GP_ARG(dX && dout, "null pointer");
  GP_ARG(mode == GPEMU_LOGPOST_LOWRANK || mode == GPEMU_LOGPOST_EXACT, "mode");
  if (!m->lik_ready) { set_error("gpemu_likelihood_setup has not been called"); return GPEMU_ERR_STATE; }
  GP_HIP(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  for (int64_t off = 0; off < B; off += MAX_CHUNK) {
    const int64_t nb = (B - off < MAX_CHUNK) ? (B - off) : MAX_CHUNK;
    GP_TRY(gp_predict_core(m, nb, dX + off * m->d, st, mode == GPEMU_LOGPOST_LOWRANK));
    if (mode == GPEMU_LOGPOST_LOWRANK) GP_TRY(launch_loglik_lowrank(m, nb, m->ws.Xq, dout + off, 0, st));
    else GP_TRY(launch_loglik_exact(m, nb, m->ws.Xq, dout + off, st));
  }
  return GPEMU_OK;
}

int gpemu_logpost(gpemu_model *m, int64_t B, const double *X, double *out, int mode) {
  GP_ARG(m, "null pointer");
  GP_ARG(B >= 0, "B must not be negative");
  if (B == 0) return GPEMU_OK;
  GP_ARG(X && out, "null pointer");
  GP_HIP(hipSetDevice(m->device));
  hipStream_t st = m->stream;
  double *dX = nullptr, *dout = nullptr;
  int rc = dev_alloc(&dX, B * m->d);
  if (rc == GPEMU_OK) rc = dev_alloc(&dout, B);
  if (rc == GPEMU_OK) rc = upload(dX, X, B * m->d, st);
  if (rc == GPEMU_OK) rc = gpemu_logpost_dev(m, B, dX, dout, mode, st);
  if (rc == GPEMU_OK) {
    hipError_t e = hipMemcpyAsync(out, dout, sizeof(double) * B, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("logpost: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  }
  hipFree(dX); hipFree(dout);
  return rc;
}

// ---- truncation covariance -----------------------------------------------------------------------
}  // extern "C"
namespace gpemu {
// rows n_pc.. of the components, zero padded to [Kp][Fp]: A plain, Bm scaled by the row's explained variance
__global__ void trunc_pack_kernel(const double *__restrict__ comp, const double *__restrict__ ev, double *__restrict__ A,
                                  double *__restrict__ Bm, int F, int Fp, int K, int Kp, int n_pc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Kp * Fp) return;
  const int r = (int)(idx / Fp), f = (int)(idx - (int64_t)r * Fp);
  double v = 0.0, lam = 0.0;
  if (r < K && f < F) { v = comp[(int64_t)(n_pc + r) * F + f]; lam = ev[n_pc + r]; }
  A[idx] = v;
  Bm[idx] = lam * v;
}
}  // namespace gpemu
extern "C" {

int gpemu_truncation_cov(int device, int64_t n_comp, int64_t F, int64_t n_pc, const double *components,
                         const double *explained_variance, double *cov_out) {
  GP_ARG(components && explained_variance && cov_out, "null pointer");
  GP_ARG(n_comp > 0 && F > 0 && n_pc >= 0 && n_pc <= n_comp, "n_comp, F, n_pc");
  GP_TRY(check_device(device));
  GP_HIP(hipSetDevice(device));
  const int64_t K = n_comp - n_pc;
  if (K == 0) {
    memset(cov_out, 0, sizeof(double) * (size_t)(F * F));
    return GPEMU_OK;
  }
  const int64_t Fp = round_up(F, 64), Kp = round_up(K, 32);
  double *dcomp = nullptr, *dev_ = nullptr, *dA = nullptr, *dB = nullptr, *dC = nullptr;
  hipStream_t st = nullptr;   // a one-off setup product: the null stream
  int rc = dev_alloc(&dcomp, n_comp * F);
  if (rc == GPEMU_OK) rc = dev_alloc(&dev_, n_comp);
  if (rc == GPEMU_OK) rc = dev_alloc(&dA, Kp * Fp);
  if (rc == GPEMU_OK) rc = dev_alloc(&dB, Kp * Fp);
  if (rc == GPEMU_OK) rc = dev_alloc(&dC, Fp * Fp);
  if (rc == GPEMU_OK) rc = upload(dcomp, components, n_comp * F, st);
  if (rc == GPEMU_OK) rc = upload(dev_, explained_variance, n_comp, st);
  if (rc == GPEMU_OK) {
    const int64_t n = Kp * Fp;
    hipLaunchKernelGGL(trunc_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dcomp, dev_, dA, dB,
                       (int)F, (int)Fp, (int)K, (int)Kp, (int)n_pc);
    GemmArgs g;                       // C[f][g] = sum_r A[r][f] * (lambda_r A[r][g]): both operands k-major
    g.A = dA; g.B = dB; g.C = dC;
    g.lda = Fp; g.ldb = Fp; g.ldc = Fp;
    g.M = (int)Fp; g.N = (int)Fp; g.K = (int)Kp;
    rc = launch_gemm(g, true, true, 1, st);
  }
  if (rc == GPEMU_OK) {
    hipError_t e = hipMemcpy2DAsync(cov_out, sizeof(double) * F, dC, sizeof(double) * Fp, sizeof(double) * F, (size_t)F,
                                    hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("truncation_cov: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  }
  hipFree(dcomp); hipFree(dev_); hipFree(dA); hipFree(dB); hipFree(dC);
  return rc;
}

// ---- full predict ----------------------------------------------------------------------------------
}  // extern "C"
namespace gpemu {
}  // namespace gpemu
extern "C" {

int gpemu_predict_full_dev(gpemu_model *m, int64_t B, const double *dX, double n_div, double *dcv,
                           double *dcov, void *stream) {
  GP_ARG(m && dX && dcv && dcov, "null pointer");
  GP_ARG(B > 0 && n_div >= 1.0, "B, n_div");
  GP_HIP(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  const int64_t F = m->F;
  for (int64_t off = 0; off < B; off += MAX_CHUNK) {
    const int64_t nb = (B - off < MAX_CHUNK) ? (B - off) : MAX_CHUNK;
    GP_TRY(gp_predict_core(m, nb, dX + off * m->d, st));
    // no mean / variance arrays: the writer sums the GP stage's partials itself (one launch less)
    GP_TRY(launch_predict_full(m, nb, n_div, dcv + off * F, dcov + off * F * F, st, nullptr, nullptr));
  }
  return GPEMU_OK;
}

int gpemu_predict_full(gpemu_model *m, int64_t B, const double *X, double n_div, double *cv_out,
                       double *cov_out) {
  GP_ARG(m && X && cv_out && cov_out, "null pointer");
  GP_ARG(B > 0, "B must be positive");
  GP_HIP(hipSetDevice(m->device));
  hipStream_t st = m->stream;
  const int64_t F = m->F;
  double *dX = nullptr, *dcv = nullptr, *dcov = nullptr;
  int rc = dev_alloc(&dX, B * m->d);
  if (rc == GPEMU_OK) rc = dev_alloc(&dcv, B * F);
  if (rc == GPEMU_OK) rc = dev_alloc(&dcov, B * F * F);
  if (rc == GPEMU_OK) rc = upload(dX, X, B * m->d, st);
  if (rc == GPEMU_OK) rc = gpemu_predict_full_dev(m, B, dX, n_div, dcv, dcov, st);
  if (rc == GPEMU_OK) {
    hipError_t e = hipMemcpyAsync(cv_out, dcv, sizeof(double) * B * F, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
      e = hipMemcpyAsync(cov_out, dcov, sizeof(double) * B * F * F, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("predict_full: %s", hipGetErrorString(e)); rc = GPEMU_ERR_HIP; }
  }
  hipFree(dX); hipFree(dcv); hipFree(dcov);
  return rc;
}

}  // extern "C"
