// Triangular GEMM with fused column sum-of-squares for SMALL batches (B <= 128 columns per launch: one rank's
// share of the proposing half on a 4- or 8-GPU run, ref: mcmc.py:77-85 replaced by walker sharding).
//
//   V[i][b] = sum_{j<=i} Wt[p][j][i] * KS[p][j][b],   out[b][p][rb] = sum_{i in 32-row block rb} V[i][b]^2
//
// With 64 columns and 10 PCs a rank has 0.64 GFLOP per launch: 2.5 MFLOP per CU, less than ONE 64 x 128 item of
// the large-batch kernel.  What bounds a small batch is therefore (a) the longest item -- an item's time is its
// FLOPs over one CU's rate whatever its shape -- and (b) the operand traffic, which grows as the tile shrinks.
// Tile 32 rows x 32 columns: the longest item (K = 1024) is 2.1 MFLOP = 6.8 us of one CU at peak, W is read once
// per 32-column block (second read from the XCD's L2, see the schedule) and K_*^T once per 32-row block.
//
// One 512-thread workgroup per CU, persistent over an LPT list of (PC, row block, column block) items.  The eight
// waves split K, not the tile: wave w owns k-rows 4 w .. 4 w + 3 of every 32-deep k-tile and accumulates the whole
// 32 x 32 tile (2 x 2 f64 MFMA tiles: four independent accumulator chains).  So no operand is shared between waves
// and NOTHING goes through LDS in the k loop: a wave loads its MFMA fragments straight from L2 into registers --
// one global_load_dwordx4 per operand and k-step, lane (q, lk) fetching W[k0 + lk][i0 + 2 q .. 2 q + 1]: the two
// adjacent values feed the two row tiles, which therefore hold the even and the odd rows (the row order inside a
// block is irrelevant to a column sum of squares), likewise the even / odd columns -- with a register ring four
// k-tiles deep (eight gave the same time: the loop is not latency bound), no barrier, no LDS bandwidth.  (History, C3 at B = 64: LDS-staged 32 x 64 items 31 us; persistent
// 32 x 32 items with LDS-direct loads 24.5 us -- the LDS-DMA path fills at most ~28 B/clk per CU, the whole budget
// of this tile shape; this form: see DESIGN.md.)  The eight K-slices are summed through LDS in the item epilogue in
// a fixed order, so results are deterministic.
#include <algorithm>

#include "internal.h"

namespace gpemu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int ST_M = 32, ST_N = 32, ST_K = 32;
constexpr int ST_MAX_ITEMS = 256;
// persistent workers per CU for `nitems` items: what bounds this kernel is each wave's own load -> MFMA dependency, not the
// operand bytes (profiles/r04_small_gemm_64col_negative.txt), so more waves per SIMD hide more of it -- as long as every
// worker still gets ~1.6 items for the LPT schedule to balance (measured at C3: 64 columns = 640 items: 22.9 / 21.6 / 27.4
// us with 1 / 2 / 3 workers per CU; 128 columns = 1280 items: 36.4 / 34.5 / 31.9 us)
static inline int small_workers_per_cu(int64_t nitems, int num_cu) {
  if (nitems >= (int64_t)5 * num_cu) return 3;
  if (nitems >= (int64_t)2 * num_cu) return 2;
  return 1;
}

struct SmallItem {
  int p, rb, col0, pad;
};

// ST_RING: k-tiles in flight per wave (two 16-byte loads each); WPE: waves per SIMD the register budget is cut for
// (4: two workgroups per CU, ring of 4; 6: three per CU, 80 VGPRs, ring of 3).  Same arithmetic in both.
// `worker`: index of this workgroup's item list in the schedule
template <int ST_RING>
__device__ __forceinline__ void trmm_vsq_small_body(
    const double *__restrict__ Wt, const double *__restrict__ KS, double *__restrict__ out,
    const SmallItem *__restrict__ sched, const int *__restrict__ sched_cnt, int max_items, int64_t Npad,
    int64_t Bcap, int k, int nrb, const int worker) {
  __shared__ __attribute__((aligned(16))) double scratch[4 * 4 * 64 * 4];   // K-slice partial tiles, 32 KiB
  __shared__ SmallItem s_items[ST_MAX_ITEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, lk = lane >> 4;

  const int nitems = __builtin_amdgcn_readfirstlane(sched_cnt[worker]);
  if (tid < nitems) s_items[tid] = sched[(int64_t)worker * max_items + tid];
  __syncthreads();
  if (nitems == 0) return;
  // item fields as wave-uniform scalars: every branch on them is a scalar branch
  auto item_p = [&](int i) { return __builtin_amdgcn_readfirstlane(s_items[i].p); };
  auto item_rb = [&](int i) { return __builtin_amdgcn_readfirstlane(s_items[i].rb); };
  auto item_col0 = [&](int i) { return __builtin_amdgcn_readfirstlane(s_items[i].col0); };

  // ---- load cursor: ST_RING k-tiles ahead of the compute cursor, across item boundaries ----
  const int64_t laneA = (int64_t)(4 * wave + lk) * Npad + 2 * q;     // this lane's k-row and row pair in a k-tile
  const int64_t laneB = (int64_t)(4 * wave + lk) * Bcap + 2 * q;
  const int64_t astep = (int64_t)ST_K * Npad, bstep = (int64_t)ST_K * Bcap;
  int l_item = 0, l_t = 0, l_nt = 1;
  const double *l_pa = Wt, *l_pb = KS;
  auto l_open = [&]() {
    const int ip = item_p(l_item), irb = item_rb(l_item), ic0 = item_col0(l_item);
    l_nt = irb + 1;                                 // k-tiles of 32: rows j <= i0 + 31
    l_pa = Wt + (int64_t)ip * Npad * Npad + (int64_t)irb * ST_M + laneA;
    l_pb = KS + (int64_t)ip * Npad * Bcap + ic0 + laneB;
    l_t = 0;
  };
  auto issue = [&](d2 &ra, d2 &rb) {
    ra = *reinterpret_cast<const d2 *>(l_pa);
    rb = *reinterpret_cast<const d2 *>(l_pb);
    if (l_item < nitems) {
      if (++l_t == l_nt) {
        if (++l_item < nitems) l_open();            // else: stay on the last k-tile (harmless re-read)
      } else {
        l_pa += astep;
        l_pb += bstep;
      }
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  d2 ra[ST_RING], rb[ST_RING];
#pragma unroll
  for (int u = 0; u < ST_RING; ++u) ra[u] = rb[u] = d2{0.0, 0.0};
  l_open();
#pragma unroll
  for (int u = 0; u < ST_RING; ++u) issue(ra[u], rb[u]);

  // one accumulator set: four independent MFMA chains per wave (a second set, taken in turn by successive k-tiles, made
  // no difference in time and costs 32 VGPRs; with one set every variant of this kernel sums in the same order)
  d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
  int c_item = 0, c_t = 0;
  int cur_p = item_p(0), cur_rb = item_rb(0), cur_col0 = item_col0(0);
  int c_nt = cur_rb + 1;

  for (;;) {
    bool done = false;
#pragma unroll
    for (int u = 0; u < ST_RING; ++u) {
      const d2 av = ra[u], bv = rb[u];
      issue(ra[u], rb[u]);                          // the slot's next occupant: k-tile + ST_RING
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
      if (++c_t == c_nt) {
        // Item finished: sum the eight K-slices (fixed order: deterministic) and reduce V^2 over the 32 rows.
        // Slot of a value: [wave][tile x + 2 y][lane][reg]  (8 KiB per wave).  Tile (x, y), lane (lr, lk'), register r
        // is row i0 + 2 (lk' + 4 r) + x, column col0 + 2 lr + y.
        lds_barrier();                              // the previous item's scratch reads are over
        if (wave >= 4) {
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
              *reinterpret_cast<d4 *>(&scratch[(((wave - 4) * 4 + x + 2 * y) * 64 + lane) * 4]) = acc[x][y];
        }
        lds_barrier();
        if (wave < 4) {
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
              d4 *slot = reinterpret_cast<d4 *>(&scratch[((wave * 4 + x + 2 * y) * 64 + lane) * 4]);
              *slot = acc[x][y] + *slot;            // slices w and w + 4
            }
        }
        lds_barrier();
        {
          // all 512 threads: wave w takes columns 4 w .. 4 w + 3, lane = (16 row groups) x (4 columns); a row group
          // is (row tile x, accumulator lane group lk', register pair h)
          const int col = 4 * wave + (lane & 3), g = lane >> 2;
          const int x = g >> 3, lkp = (g >> 1) & 3, h = g & 1;
          const int y = col & 1, lrp = col >> 1;
          const int off = ((x + 2 * y) * 64 + lkp * 16 + lrp) * 4 + 2 * h;
          d2 v = *reinterpret_cast<const d2 *>(&scratch[off]);
#pragma unroll
          for (int w2 = 1; w2 < 4; ++w2) v = v + *reinterpret_cast<const d2 *>(&scratch[w2 * 1024 + off]);
          double sq = fma(v[0], v[0], v[1] * v[1]);
          sq += __shfl_xor(sq, 4);
          sq += __shfl_xor(sq, 8);
          sq += __shfl_xor(sq, 16);
          sq += __shfl_xor(sq, 32);
          if (lane < 4) out[(((int64_t)cur_col0 + col) * k + cur_p) * nrb + cur_rb] = sq;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
        if (++c_item == nitems) { done = true; break; }
        cur_p = item_p(c_item); cur_rb = item_rb(c_item); cur_col0 = item_col0(c_item);
        c_nt = cur_rb + 1;
        c_t = 0;
      }
    }
    if (done) break;
  }
}

template <int ST_RING, int WPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void trmm_vsq_small_kernel(
    const double *__restrict__ Wt, const double *__restrict__ KS, double *__restrict__ out,
    const SmallItem *__restrict__ sched, const int *__restrict__ sched_cnt, int max_items, int64_t Npad,
    int64_t Bcap, int k, int nrb) {
  trmm_vsq_small_body<ST_RING>(Wt, KS, out, sched, sched_cnt, max_items, Npad, Bcap, k, nrb, (int)blockIdx.x);
}

// Several emulation groups in one launch: workgroups [start[g], start[g + 1]) are the workers of group g's own schedule
// (each worker's items belong to one group, so the body is the single-group kernel's -- the same bits)
constexpr int SMALL_GROUPS_MAX = 8;
struct SmallGroup {
  const double *Wt, *KS;
  double *out;
  const SmallItem *sched;
  const int *cnt;
  int64_t Npad, Bcap;
  int max_items, k, nrb;
};
struct SmallGroups {
  SmallGroup g[SMALL_GROUPS_MAX];
  int start[SMALL_GROUPS_MAX + 1];
  int ng;
};
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void trmm_vsq_small_groups_kernel(SmallGroups sg) {
  int gi = 0;
  while (gi + 1 < sg.ng && (int)blockIdx.x >= sg.start[gi + 1]) ++gi;
  const SmallGroup &g = sg.g[gi];
  trmm_vsq_small_body<4>(g.Wt, g.KS, g.out, g.sched, g.cnt, g.max_items, g.Npad, g.Bcap, g.k, g.nrb,
                         (int)blockIdx.x - sg.start[gi]);
}

// host side: LPT schedule, XCD-aware.  Workgroups are dispatched round-robin over the 8 XCDs (worker w on XCD
// w % 8, each with its own 4 MiB L2).  The items of one (PC, column block) group read the same K_*^T block and
// the column blocks of one (PC, row block) the same W strip, so whole groups are dealt to XCDs in (PC, column
// block) order -- adjacent column blocks of a PC land on the same XCD, equally long, and start together -- and
// the groups left over when the count is not a multiple of 8 are spread over every XCD.
// worker_cap > 0: that many workers at most (several groups sharing one launch: k_trmm_small.hip, ..._groups)
// keep (probes only, tools/share_probe.hip): restrict the launch to the (PC, 32-row block) pairs it accepts
static void build_small_schedule(const gpemu_model *m, int ncb, std::vector<SmallItem> &flat, std::vector<int> &cnt,
                                 int &max_items, int &nworkers, int worker_cap = 0, bool (*keep)(int p, int rb) = nullptr) {
  const int nrb = (int)(m->Npad / ST_M), k = (int)m->k;
  struct It { double cost; SmallItem it; int group; };
  std::vector<It> items;
  const double ov = 1.5;             // per-item epilogue, in k-tiles of 32
  for (int rb = 0; rb < nrb; ++rb) {
    const double nt = (double)(rb + 1);
    for (int p = 0; p < k; ++p)
      for (int cb = 0; cb < ncb; ++cb)
        if (!keep || keep(p, rb)) items.push_back({nt + ov, SmallItem{p, rb, cb * ST_N, 0}, p * ncb + cb});
  }
  std::stable_sort(items.begin(), items.end(), [](const It &a, const It &b) { return a.cost > b.cost; });
  const int ncu = worker_cap > 0 ? worker_cap : m->num_cu * small_workers_per_cu((int64_t)items.size(), m->num_cu);
  nworkers = ncu < (int)items.size() ? ncu : (int)items.size();
  std::vector<std::vector<SmallItem>> per(nworkers);
  std::vector<double> load(nworkers, 0.0);
  const int nxcd = 8, ngroups = k * ncb;
  const bool use_xcd = !keep && nworkers == ncu && nworkers % nxcd == 0 && ngroups >= nxcd;
  const int whole = use_xcd ? (ngroups / nxcd) * nxcd : 0;     // groups [0, whole) live on one XCD each
  const int gper = use_xcd ? ngroups / nxcd : 1;
  for (const It &x : items) {
    int best;
    if (x.group < whole) {
      const int xcd = x.group / gper;
      best = xcd;
      for (int w = xcd; w < nworkers; w += nxcd)
        if (load[w] < load[best]) best = w;
    } else {
      best = 0;
      for (int w = 1; w < nworkers; ++w)
        if (load[w] < load[best]) best = w;
    }
    per[best].push_back(x.it);
    load[best] += x.cost;
  }
  max_items = 1;
  for (auto &v : per) max_items = v.size() > (size_t)max_items ? (int)v.size() : max_items;
  flat.assign((size_t)nworkers * max_items, SmallItem{0, 0, 0, 0});
  cnt.assign(nworkers, 0);
  for (int w = 0; w < nworkers; ++w) {
    cnt[w] = (int)per[w].size();
    for (size_t i = 0; i < per[w].size(); ++i) flat[(size_t)w * max_items + i] = per[w][i];
  }
}

// XCD on which build_small_schedule places the items of (PC p, the 32-column block holding column col), or -1 (the
// left-over groups, or no XCD-aware placement): the cross-kernel writes those rows from the same XCD (k_front.hip)
int small_trmm_xcd_of(const gpemu_model *m, int64_t B, int p, int64_t col) {
  const int nrb = (int)(m->Npad / ST_M), k = (int)m->k, ncb = (int)(round_up(B, ST_N) / ST_N);
  const int64_t nitems = (int64_t)nrb * k * ncb;
  const int ncu = m->num_cu * small_workers_per_cu(nitems, m->num_cu), nxcd = 8, ngroups = k * ncb;
  if (nitems < ncu || ncu % nxcd != 0 || ngroups < nxcd) return -1;
  const int gper = ngroups / nxcd, g = p * ncb + (int)(col / ST_N);
  return g < gper * nxcd ? g / gper : -1;
}

// returns GPEMU_ERR_UNSUPPORTED (without setting an error) when the shape needs more items per worker than the
// kernel holds; the caller then uses the general small-batch kernel
// the schedule for B columns on the device (m->sm_*); GPEMU_ERR_UNSUPPORTED as below
static int small_schedule_ready(gpemu_model *m, int64_t B, hipStream_t st, int worker_cap = 0) {
  const int ncb = (int)(round_up(B, ST_N) / ST_N);
  const int cap = worker_cap > 0 ? -worker_cap : m->num_cu;      // (key of the one cached schedule)
  if (m->sm_ncb != ncb || m->sm_cap != cap) {
    const gpemu_model::SchedEntry *hit = nullptr;
    for (const auto &e : m->sm_cache)
      if (e.ncb == ncb && e.cap == cap) hit = &e;
    if (!hit) {
      std::vector<SmallItem> flat;
      std::vector<int> cnt;
      int max_items = 0, nworkers = 0;
      build_small_schedule(m, ncb, flat, cnt, max_items, nworkers, worker_cap);
      if (max_items > ST_MAX_ITEMS) return GPEMU_ERR_UNSUPPORTED;     // (nothing cached or replaced: the current one stays)
      if (m->sm_cache.size() >= 16) {                                  // bounded: drop the oldest once nothing reads it
        GP_HIP(hipStreamSynchronize(st));
        GP_HIP(hipStreamSynchronize(m->stream));
        (void)hipFree(m->sm_cache.front().items);
        (void)hipFree(m->sm_cache.front().cnt);
        m->sm_cache.erase(m->sm_cache.begin());
      }
      gpemu_model::SchedEntry e{ncb, cap, nullptr, nullptr, max_items, nworkers};
      GP_HIP(hipMalloc(&e.items, sizeof(SmallItem) * flat.size()));
      GP_HIP(hipMalloc((void **)&e.cnt, sizeof(int) * cnt.size()));
      GP_HIP(hipMemcpy(e.items, flat.data(), sizeof(SmallItem) * flat.size(), hipMemcpyHostToDevice));
      GP_HIP(hipMemcpy(e.cnt, cnt.data(), sizeof(int) * cnt.size(), hipMemcpyHostToDevice));
      m->sm_cache.push_back(e);
      hit = &m->sm_cache.back();
    }
    m->sm_items = hit->items; m->sm_cnt = hit->cnt;
    m->sm_ncb = ncb; m->sm_cap = cap; m->sm_max_items = hit->max_items; m->sm_workers = hit->workers;
  }
  return GPEMU_OK;
}

int launch_trmm_vsq_small(gpemu_model *m, int64_t B, hipStream_t st) {
  Workspace &w = m->ws;
  const int nrb = (int)(m->Npad / ST_M);
  const int rc = small_schedule_ready(m, B, st);
  if (rc != GPEMU_OK) return rc;
  w.cur_nrb = nrb;
  const int pe0 = prof_mark(m, st);
  if (m->sm_workers > 2 * m->num_cu)
    hipLaunchKernelGGL((trmm_vsq_small_kernel<3, 6>), dim3((unsigned)m->sm_workers), dim3(512), 0, st, m->Wt, w.KS, w.vsq_part,
                       (const SmallItem *)m->sm_items, m->sm_cnt, m->sm_max_items, m->Npad, w.Bcap, (int)m->k, nrb);
  else
    hipLaunchKernelGGL((trmm_vsq_small_kernel<4, 4>), dim3((unsigned)m->sm_workers), dim3(512), 0, st, m->Wt, w.KS, w.vsq_part,
                       (const SmallItem *)m->sm_items, m->sm_cnt, m->sm_max_items, m->Npad, w.Bcap, (int)m->k, nrb);
  GP_HIP(hipGetLastError());
  prof_pair(m, 0, pe0, prof_mark(m, st));
  return GPEMU_OK;
}

// the small-batch GEMMs of ng groups (B <= 128 columns each) in one launch; GPEMU_ERR_UNSUPPORTED (no error set, nothing
// launched) when a group's shape does not fit the kernel: the caller launches group by group
// All groups' workers resident together: two 512-thread workgroups per CU in all, dealt to the groups by their share
// of the k-tiles (left to themselves three groups of the shipped size bring 888 workers for 512 slots: 17.6 us against
// 7.4 - 10.4 us for each group alone).  Multiples of 8, so that worker w of a group still sits on XCD w % 8.
int prepare_trmm_vsq_small_groups(gpemu_model *const *ms, int ng, int64_t B, hipStream_t st) {
  if (ng > SMALL_GROUPS_MAX) return GPEMU_ERR_UNSUPPORTED;
  const int ncb = (int)(round_up(B, ST_N) / ST_N);
  double cost[SMALL_GROUPS_MAX], total = 0.0;
  for (int g = 0; g < ng; ++g) {
    const double nrb = (double)(ms[g]->Npad / ST_M);
    cost[g] = (double)ms[g]->k * ncb * (0.5 * nrb * (nrb + 1.0) + 1.5 * nrb);
    total += cost[g];
  }
  const int slots = 2 * ms[0]->num_cu;
  for (int g = 0; g < ng; ++g) {
    int cap = (int)(slots * cost[g] / total) / 8 * 8;
    if (cap < 8) cap = 8;
    const int rc = small_schedule_ready(ms[g], B, st, cap);
    if (rc != GPEMU_OK) return rc;
  }
  return GPEMU_OK;
}

int launch_trmm_vsq_small_groups(gpemu_model *const *ms, int ng, int64_t B, hipStream_t st) {
  const int rc0 = prepare_trmm_vsq_small_groups(ms, ng, B, st);
  if (rc0 != GPEMU_OK) return rc0;
  SmallGroups sg;
  sg.ng = ng;
  sg.start[0] = 0;
  for (int g = 0; g < ng; ++g) {
    gpemu_model *m = ms[g];
    Workspace &w = m->ws;
    const int nrb = (int)(m->Npad / ST_M);
    w.cur_nrb = nrb;
    sg.g[g] = SmallGroup{m->Wt, w.KS, w.vsq_part, (const SmallItem *)m->sm_items, m->sm_cnt, m->Npad, w.Bcap,
                         m->sm_max_items, (int)m->k, nrb};
    sg.start[g + 1] = sg.start[g] + m->sm_workers;
  }
  hipLaunchKernelGGL(trmm_vsq_small_groups_kernel, dim3((unsigned)sg.start[ng]), dim3(512), 0, st, sg);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
