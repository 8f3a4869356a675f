// Probe for VERDICT r4 items 1 and 2 (profiles/r05_share_probe.txt).  Compiles the product's cross-kernel and both
// triangular GEMMs (k_predict.hip, k_trmm_small.hip) into the tool and times them on the C3 shape:
//
//  (1) "model-sharded" share of an 8-rank half-step: instead of 64 of the 512 proposals against the whole model
//      (walker sharding: every rank re-streams all of W for 64 columns), rank r evaluates ALL 512 columns against one
//      eighth of the (PC, row block) pairs, contiguous in (PC, row block) and balanced by their k-tile counts.  Timed
//      on the rank's item list: (a) the large-batch kernel, LPT over whole 64 x 128 items, (b) the same with every
//      item halved to 64 columns, (c) the small-batch kernel (32 x 32 items, K split over the waves) at 512 columns;
//      beside them today's walker-sharded launch (small-batch kernel, 64 columns, all pairs) and the full launch.
//  (2) the cross-kernel BESIDE the triangular GEMM: both on the device together from two streams, unsynchronised
//      (what a producer / consumer overlap inside a half-step could gain at best), and as gated pairs (one cross-kernel
//      and one GEMM started together, the next pair behind both: the two-stream form with its cross-stream events).
//   usage: share_probe [world = 8]
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../k_predict.hip"
#include "../k_trmm_small.hip"

namespace gpemu {
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
int prof_mark(gpemu_model *, hipStream_t) { return -1; }
void prof_pair(gpemu_model *, int, int, int) {}
}  // namespace gpemu

using namespace gpemu;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static int g_world = 8, g_rank = 0, g_rows = 64;      // the share being filtered: rows per block of the builder in use
static int64_t g_Npad = 1024;
static int g_k = 10;
// (PC, row block) pairs in (PC, row block) order, cut into `world` runs of equal k-tile cost
static bool keep_pair(int p, int rb) {
  const int nrb = (int)(g_Npad / g_rows);
  auto cost = [&](int r) { return (double)((int64_t)(r + 1) * g_rows + 31) / 32; };
  double total = 0.0, before = 0.0;
  for (int r = 0; r < nrb; ++r) total += cost(r);
  total *= g_k;
  for (int q = 0; q < p; ++q)
    for (int r = 0; r < nrb; ++r) before += cost(r);
  for (int r = 0; r < rb; ++r) before += cost(r);
  const double mid = before + 0.5 * cost(rb);
  return (int)(mid / total * g_world) == g_rank;
}

template <typename F>
static double time_us(F &&launch, hipStream_t st, int warm, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < warm; ++i) launch();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms * 1e3 / reps;
}

int main(int argc, char **argv) {
  g_world = argc > 1 ? atoi(argv[1]) : 8;
  const int64_t N = 1000, Npad = 1024, k = 10, d = 6, B = 512, Bcap = 512;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  // a model by hand: what the launchers read
  gpemu_model m;
  m.N = N; m.d = d; m.F = 500; m.k = k; m.Npad = Npad; m.num_cu = prop.multiProcessorCount; m.vsq_nrb = Npad / 64;
  CK(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
  hipStream_t sA = m.stream, sB;
  CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
  {
    const double lo[8] = {0.1, 1, 0.0067, 0.0067, 0, 0.05, 0.3, 2}, hi[8] = {0.5, 10, 10, 10, 1.5, 100, 0.9, 7};
    std::vector<double> X(N * d), ls(k * d), al(k * N), Q(Bcap * 8, 0.0);
    for (int64_t j = 0; j < N; ++j) for (int64_t dd = 0; dd < d; ++dd) X[j * d + dd] = lo[dd] + (hi[dd] - lo[dd]) * rnd();
    for (int64_t p = 0; p < k; ++p) for (int64_t dd = 0; dd < d; ++dd) ls[p * d + dd] = (hi[dd] - lo[dd]) * 0.5;
    for (auto &v : al) v = 2.0 * rnd() - 1.0;
    for (int64_t b = 0; b < B; ++b) for (int64_t dd = 0; dd < d; ++dd) Q[b * 8 + dd] = lo[dd] + (hi[dd] - lo[dd]) * rnd();
    KstarHost kh;
    build_kstar_operands(N, Npad, d, k, 0, X.data(), ls.data(), al.data(), kh);
    m.ksteps = kh.ksteps;
    auto up = [&](double **dst, const std::vector<double> &h) {
      CK(hipMalloc((void **)dst, sizeof(double) * h.size()));
      CK(hipMemcpy(*dst, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    };
    up(&m.Xa, kh.Xa); up(&m.alf, kh.alf); up(&m.qsc, kh.qsc); up(&m.qof, kh.qof); up(&m.etab, kh.tab);
    up(&m.ws.Xq, Q);
    std::vector<double> c(k, 0.0);
    up(&m.constv, c);
    // W: a lower-triangular random factor inverse stands in (the kernels' time does not depend on the values, the clocks
    // the part holds do: random data, not zeros)
    std::vector<double> W((size_t)k * Npad * Npad);
    for (auto &v : W) v = rnd() - 0.5;
    up(&m.Wt, W);
  }
  Workspace &w = m.ws;
  w.Bcap = Bcap;
  CK(hipMalloc((void **)&w.KS, sizeof(double) * k * Npad * Bcap));
  CK(hipMalloc((void **)&w.mean_part, sizeof(double) * k * (Npad / 32) * Bcap));
  CK(hipMalloc((void **)&w.vsq_part, sizeof(double) * k * (Npad / 32) * Bcap));
  CK(hipMemset(w.vsq_part, 0, sizeof(double) * k * (Npad / 32) * Bcap));
  if (launch_kstar(&m, B, w.Xq, sA, nullptr) != GPEMU_OK) { printf("kstar failed\n"); return 1; }     // K_*^T: real values
  CK(hipStreamSynchronize(sA));
  printf("share_probe: C3 shape N = %lld (Npad %lld), k = %lld, %lld columns, %d CUs, world %d\n", (long long)N,
         (long long)Npad, (long long)k, (long long)B, m.num_cu, g_world);

  // ---- reference launches --------------------------------------------------------------------------------------
  const double t_full = time_us([&] { launch_trmm_vsq(&m, B, sA); }, sA, 300, 300);
  const double t_kstar = time_us([&] { launch_kstar(&m, B, w.Xq, sA, nullptr); }, sA, 300, 300);
  const double t_64 = time_us([&] { launch_trmm_vsq(&m, 64, sA); }, sA, 300, 300);
  const double t_kstar64 = time_us([&] { launch_kstar(&m, 64, w.Xq, sA, nullptr); }, sA, 300, 300);
  printf("reference: large-batch GEMM at 512 columns %.2f us, cross-kernel %.2f us; walker-sharded share of 8 ranks (64 columns): "
         "small-batch GEMM %.2f us, cross-kernel %.2f us\n", t_full, t_kstar, t_64, t_kstar64);

  // ---- (1) model-sharded share: the rank's (PC, row block) pairs at all 512 columns ----------------------------
  g_Npad = Npad; g_k = (int)k;
  const int ncb = (int)(B / TILE);
  auto upload_sched = [&](const void *flat, size_t fbytes, const std::vector<int> &cnt, void **ditems, int **dcnt) {
    CK(hipMalloc(ditems, fbytes));
    CK(hipMalloc((void **)dcnt, sizeof(int) * cnt.size()));
    CK(hipMemcpy(*ditems, flat, fbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(*dcnt, cnt.data(), sizeof(int) * cnt.size(), hipMemcpyHostToDevice));
  };
  double worst[3] = {0, 0, 0};
  for (g_rank = 0; g_rank < g_world; ++g_rank) {
    double t[3] = {0, 0, 0};
    int nit[3] = {0, 0, 0}, nwk[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int variant = 0; variant < 2; ++variant) {            // (a) whole items, (b) every item halved
      g_rows = 64;
      std::vector<TrmmItem> flat;
      std::vector<int> cnt;
      int max_items = 0, nworkers = 0;
      build_trmm_schedule(&m, ncb, flat, cnt, max_items, nworkers, keep_pair, variant);
      if (nworkers == 0 || max_items > TRMM_MAX_ITEMS) continue;
      void *ditems = nullptr; int *dcnt = nullptr;
      upload_sched(flat.data(), sizeof(TrmmItem) * flat.size(), cnt, &ditems, &dcnt);
      t[variant] = time_us([&] {
        hipLaunchKernelGGL(trmm_vsq_dma_kernel, dim3((unsigned)nworkers), dim3(512), 0, sA, m.Wt, w.KS, w.vsq_part,
                           (const TrmmItem *)ditems, dcnt, max_items, Npad, Bcap, (int)k, (int)m.vsq_nrb);
      }, sA, 200, 300);
      for (int c : cnt) nit[variant] += c;
      nwk[variant] = nworkers; mx[variant] = max_items;
      CK(hipFree(ditems)); CK(hipFree(dcnt));
    }
    {                                                          // (c) small-batch kernel, 32 x 32 items, 512 columns
      g_rows = 32;
      std::vector<SmallItem> flat;
      std::vector<int> cnt;
      int max_items = 0, nworkers = 0;
      build_small_schedule(&m, (int)(B / ST_N), flat, cnt, max_items, nworkers, 2 * m.num_cu, keep_pair);
      void *ditems = nullptr; int *dcnt = nullptr;
      upload_sched(flat.data(), sizeof(SmallItem) * flat.size(), cnt, &ditems, &dcnt);
      t[2] = time_us([&] {
        hipLaunchKernelGGL((trmm_vsq_small_kernel<4, 4>), dim3((unsigned)nworkers), dim3(512), 0, sA, m.Wt, w.KS, w.vsq_part,
                           (const SmallItem *)ditems, dcnt, max_items, Npad, Bcap, (int)k, (int)(Npad / ST_M));
      }, sA, 200, 300);
      for (int c : cnt) nit[2] += c;
      nwk[2] = nworkers; mx[2] = max_items;
      CK(hipFree(ditems)); CK(hipFree(dcnt));
    }
    printf("rank %d of %d, its (PC, row block) pairs x 512 columns: (a) large-batch kernel, whole items %6.2f us (%d items, %d workers, "
           "<= %d each); (b) all halved %6.2f us (%d items); (c) small-batch kernel 32 x 32 %6.2f us (%d items, %d workers)\n",
           g_rank, g_world, t[0], nit[0], nwk[0], mx[0], t[1], nit[1], t[2], nit[2], nwk[2]);
    for (int v = 0; v < 3; ++v) worst[v] = std::max(worst[v], t[v]);
  }
  printf("model-sharded share, slowest rank: (a) %.2f us  (b) %.2f us  (c) %.2f us   against %.2f us for the walker-sharded share "
         "(and %.2f / %.2f us for the cross-kernel of 512 / 64 columns: the model-sharded rank forms 1/%d of the 512-column rows)\n",
         worst[0], worst[1], worst[2], t_64, t_kstar, t_kstar64, g_world);

  // ---- (2) cross-kernel beside the GEMM ---------------------------------------------------------------------------
  {
    const int n_g = 300, per = 4;
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    for (int i = 0; i < 100; ++i) { launch_trmm_vsq(&m, B, sA); launch_kstar(&m, B, w.Xq, sB, nullptr); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a0, sA)); CK(hipEventRecord(b0, sB));
    for (int i = 0; i < n_g; ++i) {
      launch_trmm_vsq(&m, B, sA);
      for (int j = 0; j < per; ++j) launch_kstar(&m, B, w.Xq, sB, nullptr);
    }
    CK(hipEventRecord(a1, sA)); CK(hipEventRecord(b1, sB));
    CK(hipDeviceSynchronize());
    float msA = 0, msB = 0;
    CK(hipEventElapsedTime(&msA, a0, a1)); CK(hipEventElapsedTime(&msB, b0, b1));
    printf("unsynchronised, two streams: %d GEMMs beside %d cross-kernels: GEMM %.2f us each (alone %.2f), cross-kernel %.2f us each "
           "(alone %.2f); stream spans %.2f / %.2f ms\n", n_g, n_g * per, msA * 1e3 / n_g, t_full, msB * 1e3 / (n_g * per), t_kstar,
           msA, msB);
    // serial pair on one stream, and gated pairs on two
    const double t_serial = time_us([&] { launch_kstar(&m, B, w.Xq, sA, nullptr); launch_trmm_vsq(&m, B, sA); }, sA, 200, 300);
    std::vector<hipEvent_t> ea(8), eb(8);
    for (auto &e : ea) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : eb) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int it = 0;
    const double t_gated = time_us([&] {
      const int c = it & 7, pv = (it + 7) & 7;
      if (it > 0) { (void)hipStreamWaitEvent(sB, ea[pv], 0); (void)hipStreamWaitEvent(sA, eb[pv], 0); }
      launch_kstar(&m, B, w.Xq, sB, nullptr);
      launch_trmm_vsq(&m, B, sA);
      (void)hipEventRecord(eb[c], sB);
      (void)hipEventRecord(ea[c], sA);
      ++it;
    }, sA, 200, 300);
    CK(hipDeviceSynchronize());
    printf("pairs: cross-kernel then GEMM on one stream %.2f us per pair; started together on two streams, the next pair behind both "
           "(two cross-stream events per pair) %.2f us per pair\n", t_serial, t_gated);
  }
  return 0;
}
