// GP predictive mean / variance for a batch of parameter points, all PCs of one emulation group.
//
// Replaces  ref: emulation.py:494-499  ->  skl _gpr.py:441-494  (GaussianProcessRegressor.predict
// with return_std=True, k times):
//     K_* = kernel_(X, X_train);  mean = K_* . alpha_;  V = L_^-1 K_*^T;  var = diag - sum_i V_i^2
// Device form: W = L_^-1 is formed once at model creation, so V = W K_*^T is a lower-triangular
// GEMM on the f64 matrix cores (v_mfma_f64_16x16x4_f64) with the column sum of squares fused into
// the epilogue -- V is never written to memory.
//
//   kstar_kernel          K_*^T[p][j][b] (HBM/L2 workspace) + partial means: the pairwise squared distance as a
//                         rank-8 product on the matrix cores (predict_dev.h), table exponential, HBM-write bound
//   trmm_vsq_dma_kernel   sum_i (W_p K_*^T)[i][b]^2 per 64-row block: persistent, LDS-direct loads, XCD-aware LPT
//                         schedule (MFMA f64); batches of at most 128 columns: k_trmm_small.hip
//   reduce_mean_var_kernel  sums the partials, var = kdiag - vsq, clip, std^2
#include <algorithm>

#include "internal.h"
#include "kstar_host.h"
#include "predict_dev.h"

namespace gpemu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// Cross-kernel.  grid (Bcap / 64, Npad / rows per workgroup, k), block 256 = 4 waves on a tile of
// WR JTW x 16 training rows x 64 queries (predict_dev.h: kstar_mfma_block).  The workgroup first puts its 64 raw query
// rows into LDS -- read from the padded buffer, padded on the fly from the caller's rows, or formed as the stretch
// proposal of the sampler -- two components per thread.
struct KstarArgs {
  double *Xq;                          // [Bcap][DPAD] padded query rows (read, or written once per column block)
  const double *Xa, *alf, *qsc, *qof;  // matrix-core operands (kstar_host.h)
  const double *etab, *constv;
  const double *Xs, *inv_ls;           // Matern-0.5 only: row-major scaled training rows for the direct distance
  double *KS, *mean_part;
  int64_t N, Npad, Bcap;
  int has_const, d;
  int k, nchunk, ncb64;                // grid = ncb64 * nchunk * k workgroups (1-D)
  int gper, ncbp;                      // gper > 0: XCD-aware placement, gper of the k ncbp (PC, 128-column block) groups
                                       // of each piece of ncbp column blocks per XCD
};

// bidx: the workgroup's index within its group's grid; store_rows: this group's first workgroups keep the padded query
// rows / stretch factors (the first group of a launch that serves several: the others form the same rows and store nothing)
template <int KIND, int KS, int JTW, int NBW>
__device__ __forceinline__ void kstar_body(const KstarArgs &ka, const ProposeArgs &pa, const int bidx, const bool store_rows) {
  __shared__ double s_tab[1 << KSTAR_TB];
  __shared__ __attribute__((aligned(16))) double s_q[64 * DPAD];
  __shared__ double s_red[4 * 64];
  constexpr int WC = 4 / NBW, WR = 4 / WC, JT = WR * JTW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Workgroups are dispatched round-robin over the 8 XCDs (workgroup i runs on XCD i % 8).  The triangular GEMM that
  // consumes K_*^T deals whole (PC, 128-column block) groups to XCDs (build_trmm_schedule: group g -> XCD g / gper), and
  // the L2s are kept coherent by hardware: a store to a line that ANOTHER XCD's L2 still holds from the GEMM's reads has
  // to invalidate it there first (measured: this kernel takes 13.5 us without the GEMM in between, 19.6 us behind it).
  // So each group's rows are written by the XCD that will read them: same L2, no cross-XCD traffic.
  int p, chunk, cb;
  const int nchunk = ka.nchunk;
  if (ka.gper > 0) {
    // the GEMM runs one launch per piece of ncbp 128-column blocks (launch_trmm_vsq: at most 512 columns each), every
    // launch dealing its k ncbp groups to the XCDs in (PC, column block) order, gper per XCD
    const int xcd = bidx & 7, slot = bidx >> 3;
    const int half = slot & 1, rest = slot >> 1;
    chunk = rest % nchunk;
    const int rest2 = rest / nchunk;
    const int piece = rest2 / ka.gper, g = xcd * ka.gper + rest2 % ka.gper;
    p = g / ka.ncbp;
    cb = 2 * (piece * ka.ncbp + g % ka.ncbp) + half;
  } else {
    cb = bidx % ka.ncb64;
    const int rest = bidx / ka.ncb64;
    chunk = rest % nchunk;
    p = rest / nchunk;
  }
  const int64_t b0 = (int64_t)cb * 64;
  const int64_t b = b0 + lane;
  const int c0 = wave, c1 = wave + 4;                // this thread's two components of query b
  const int64_t njt = ka.Npad / 16;
  KstarFrags<KS, JTW> fr;
  if (threadIdx.x < (1 << KSTAR_TB)) s_tab[threadIdx.x] = ka.etab[threadIdx.x];
  const bool keeper = store_rows && chunk == 0 && p == 0;   // the workgroup that stores the padded rows of its columns
  double q0 = 0.0, q1 = 0.0;
  if (pa.enabled) {
    // stretch-move proposal for column b (every workgroup recomputes it; one of them stores it)
    if (b < pa.n) {
      const int w = pa.idx_s[b], j = pa.partner[b];
      const double z = pa.zz[b];
      if (c0 < pa.d) {
        const double cj = pa.X[(int64_t)j * DPAD + c0], sw = pa.X[(int64_t)w * DPAD + c0];
        q0 = cj - (cj - sw) * z;                     // emcee moves/stretch.py get_proposal
      }
      if (c1 < pa.d) {
        const double cj = pa.X[(int64_t)j * DPAD + c1], sw = pa.X[(int64_t)w * DPAD + c1];
        q1 = cj - (cj - sw) * z;
      }
      if (keeper && wave == 0) pa.factors[b] = (pa.d - 1.0) * log(z);
    }
  } else if (pa.raw) {
    // caller rows [n][d]: padded on the fly
    if (b < pa.n) {
      if (c0 < pa.d) q0 = pa.raw[b * pa.d + c0];
      if (c1 < pa.d) q1 = pa.raw[b * pa.d + c1];
    }
  } else {
    q0 = ka.Xq[b * DPAD + c0];
    q1 = ka.Xq[b * DPAD + c1];
  }
  if (keeper && (pa.enabled || pa.raw)) {
    ka.Xq[b * DPAD + c0] = q0;
    ka.Xq[b * DPAD + c1] = q1;
  }
  s_q[lane * DPAD + c0] = q0;
  s_q[lane * DPAD + c1] = q1;
  __syncthreads();
  // (requesting the fragments ahead of the proposal's loads was measured: 16.45 vs 16.1 us, slower)
  kstar_load_frags<KS, JTW, NBW>(fr, ka.Xa + (int64_t)p * njt * KS * 64, ka.alf + (int64_t)p * njt * 16, (int64_t)chunk * JT,
                                 lane, wave);
  const double c = ka.has_const ? ka.constv[p] : 0.0;
  KstarDirect dir{nullptr, nullptr};
  if (KIND == 1) dir = KstarDirect{ka.Xs + (int64_t)p * ka.Npad * DPAD, ka.inv_ls + p * DPAD};
  const double sum = kstar_mfma_block<KIND, KS, JTW, NBW, KSTAR_TB>(
      s_q, s_tab, s_red, fr, ka.qsc + p * 4 * KS, ka.qof + p * 4 * KS, c, ka.d, (int64_t)chunk * JT, ka.N, ka.KS + (int64_t)p * ka.Npad * ka.Bcap + b0, ka.Bcap, dir,
      lane, wave);
  if (wave == 0) ka.mean_part[(b * ka.k + p) * nchunk + chunk] = sum;
}

template <int KIND, int KS, int JTW, int NBW>
__global__ __launch_bounds__(256, 2) void kstar_kernel(KstarArgs ka, ProposeArgs pa) {
  kstar_body<KIND, KS, JTW, NBW>(ka, pa, (int)blockIdx.x, true);
}

// The cross-kernels of several emulation groups in ONE launch (a sampler over the shipped three groups spends its
// half-step in nine ~6 us launches otherwise): workgroups [start[g], start[g + 1]) are group g's grid.  Every group
// forms the stretch proposal itself (the same arithmetic, the same rows); the first one stores it.
constexpr int GROUPS_MAX = 8;
struct KstarGroups {
  KstarArgs g[GROUPS_MAX];
  int start[GROUPS_MAX + 1];
  int ng;
};
template <int KIND, int KS, int JTW, int NBW>
__global__ __launch_bounds__(256, 2) void kstar_groups_kernel(KstarGroups kg, ProposeArgs pa) {
  int gi = 0;
  while (gi + 1 < kg.ng && (int)blockIdx.x >= kg.start[gi + 1]) ++gi;
  kstar_body<KIND, KS, JTW, NBW>(kg.g[gi], pa, (int)blockIdx.x - kg.start[gi], gi == 0);
}

// Columns per launch of the large-batch triangular GEMM.  More than 512 columns (emulation.predict on a large batch) go
// one launch per 512 columns: K_*^T of 1024 columns is 82 MB, and with W_p it no longer streams through the XCDs' L2s the
// way the 512-column schedule is built for (one launch of 1024 columns takes 228 us, two of 512 take 2 x 93 us).  Only
// when the pieces are equally wide (one cached schedule); else, and for stacked chains, the whole batch in one launch.
static int64_t trmm_piece_cols(const gpemu_model *m, int64_t B) {
  constexpr int64_t max_cols = 512;
  if (m->variant_B == 0 && B > max_cols) {
    const int64_t n = (B + max_cols - 1) / max_cols, per = round_up((B + n - 1) / n, TILE), last = B - (n - 1) * per;
    if (last > 0 && round_up(last, TILE) == per) return per;
  }
  return round_up(B, TILE);
}

// XCD whose workers will read rows (PC p, column col) of K_*^T in launch_trmm_vsq(m, B), or -1 if that launch places
// nothing XCD-aware: mirrors build_trmm_schedule / build_small_schedule (k_trmm_small.hip)
int trmm_xcd_of(const gpemu_model *m, int64_t B, int p, int64_t col) {
  const int64_t Bv = m->variant_B > 0 ? m->variant_B : B;
  if (Bv <= 128) return small_trmm_xcd_of(m, B, p, col);
  const int ncbp = (int)(trmm_piece_cols(m, B) / TILE), ngroups = (int)m->k * ncbp;
  if (ngroups % 8 != 0 || m->num_cu % 8 != 0) return -1;
  return (p * ncbp + (int)(col / TILE) % ncbp) / (ngroups / 8);
}

// arguments and grid of one group's cross-kernel for a batch of B columns (sets the workspace's chunk count)
static KstarArgs kstar_setup(gpemu_model *m, int64_t B, double *dXq, int &nwg, bool &small) {
  // only the column tiles that hold real queries; without a proposal / raw rows, the rows of dXq up to
  // round_up(B, TILE) must be finite (the sampler's proposal buffer zeroes them)
  Workspace &w = m->ws;
  const int64_t Bv = m->variant_B > 0 ? m->variant_B : B;   // a chain stacked with others is evaluated as it would be alone
  small = Bv <= KSTAR_SMALL_MAX;                     // few columns: more, shorter workgroups
  const int rows_per_wg = small ? KSTAR_ROWS_SMALL : KSTAR_ROWS_BIG;
  w.cur_nchunk = (int)(m->Npad / rows_per_wg);
  // column blocks of 64 queries: whole 128-column tiles for the triangular GEMM, except that a batch of at most
  // 64 (always served by the small-batch kernel's 64-column items) needs only its first block
  const int64_t ncols = (B <= 64) ? 64 : round_up(B, TILE);
  const int ncb64 = (int)(ncols / 64);
  nwg = ncb64 * w.cur_nchunk * (int)m->k;
  // XCD-aware placement where the large-batch GEMM's schedule is (whole groups per XCD: build_trmm_schedule), piece by
  // piece of the columns as launch_trmm_vsq will cut them
  const int ncbp = (int)(trmm_piece_cols(m, B) / TILE);
  const int ngroups = (int)m->k * ncbp;
  const bool xcd_aware = Bv > 128 && ncb64 % 2 == 0 && (ncb64 / 2) % ncbp == 0 && ngroups % 8 == 0 && m->num_cu % 8 == 0;
  const int gper = xcd_aware ? ngroups / 8 : 0;
  return KstarArgs{dXq, m->Xa, m->alf, m->qsc, m->qof, m->etab, m->constv, m->Xs, m->inv_ls, w.KS, w.mean_part,
                   m->N, m->Npad, w.Bcap, m->has_const, (int)m->d, (int)m->k, w.cur_nchunk, ncb64, gper, ncbp};
}

#define GP_KSTAR_DISPATCH(kind, ksteps, small, LAUNCH)                  \
  do {                                                                   \
    if ((ksteps) == 2) {                                                 \
      switch (kind) {                                                    \
        case 0: if (small) LAUNCH(0, 2, 1); else LAUNCH(0, 2, 2); break; \
        case 1: if (small) LAUNCH(1, 2, 1); else LAUNCH(1, 2, 2); break; \
        case 2: if (small) LAUNCH(2, 2, 1); else LAUNCH(2, 2, 2); break; \
        default: if (small) LAUNCH(3, 2, 1); else LAUNCH(3, 2, 2); break; \
      }                                                                  \
    } else {                                                             \
      switch (kind) {                                                    \
        case 0: if (small) LAUNCH(0, 3, 1); else LAUNCH(0, 3, 2); break; \
        case 1: if (small) LAUNCH(1, 3, 1); else LAUNCH(1, 3, 2); break; \
        case 2: if (small) LAUNCH(2, 3, 1); else LAUNCH(2, 3, 2); break; \
        default: if (small) LAUNCH(3, 3, 1); else LAUNCH(3, 3, 2); break; \
      }                                                                  \
    }                                                                    \
  } while (0)

int launch_kstar(gpemu_model *m, int64_t B, double *dXq, hipStream_t st, const ProposeArgs *pa) {
  const ProposeArgs pargs = pa ? *pa : ProposeArgs();
  int nwg = 0;
  bool small = false;
  const KstarArgs ka = kstar_setup(m, B, dXq, nwg, small);
  const dim3 grid((unsigned)nwg), block(256);
  const int pe0 = prof_mark(m, st);
#define GP_LAUNCH_ONE(KD, KSV, JT) hipLaunchKernelGGL((kstar_kernel<KD, KSV, JT, 2>), grid, block, 0, st, ka, pargs)
  GP_KSTAR_DISPATCH(kstar_kind(m), m->ksteps, small, GP_LAUNCH_ONE);
#undef GP_LAUNCH_ONE
  GP_HIP(hipGetLastError());
  prof_pair(m, 1, pe0, prof_mark(m, st));
  return GPEMU_OK;
}

// one launch for the cross-kernels of ng groups on the same B query rows (same base kernel and parameter count in all
// of them: the caller checks); `pa`: the stretch proposal, formed by every group, stored by the first
int launch_kstar_groups(gpemu_model *const *ms, int ng, int64_t B, double *dXq, hipStream_t st, const ProposeArgs *pa) {
  const ProposeArgs pargs = pa ? *pa : ProposeArgs();
  KstarGroups kg;
  kg.ng = ng;
  kg.start[0] = 0;
  bool small = false;
  for (int g = 0; g < ng; ++g) {
    int nwg = 0;
    kg.g[g] = kstar_setup(ms[g], B, dXq, nwg, small);
    kg.start[g + 1] = kg.start[g] + nwg;
  }
  const dim3 grid((unsigned)kg.start[ng]), block(256);
#define GP_LAUNCH_GROUPS(KD, KSV, JT) hipLaunchKernelGGL((kstar_groups_kernel<KD, KSV, JT, 2>), grid, block, 0, st, kg, pargs)
  GP_KSTAR_DISPATCH(kstar_kind(ms[0]), ms[0]->ksteps, small, GP_LAUNCH_GROUPS);
#undef GP_LAUNCH_GROUPS
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

// ------------------------------------------------------------------------------------------
// Triangular GEMM with fused column sum-of-squares.
//   V[i][b] = sum_{j<=i} Wt[j][i] * KS[j][b]      (per PC p; both operands are k-major)
//   out[b][p][rb] = sum_{i in 64-row block rb} V[i][b]^2
// MFMA v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][n = l&15];
// D[reg] is row (l>>4) + 4*reg, column l&15 (verified by tools/mfma_f64_peak).
//
// One 512-thread workgroup per CU: 8 waves = 2 per SIMD, arranged 2 (M) x 4 (N), 32 x 32 per wave
// (2 x 2 MFMA tiles, 16 accumulator registers) on a 64 x 128 tile with K step 32.  Persistent: one workgroup per CU
// walks a host-built list of items (LPT schedule over the known K extents, so the triangular work is balanced to within
// one small item) as ONE software-pipelined stream of k-tiles.  Items of the first quarter of the rows (short K) are
// split into two 64-column halves to give the schedule small pieces for its tail.  Measured history (C3, B = 512):
// 128x128 tiles / 4 waves 182 us; 64x128 / 2 workgroups per CU 123 us; one item per workgroup, 8 waves 117 us;
// persistent + register-staged 108 us; XCD-aware 106.5; LDS-direct 104.7; loads interleaved with the MFMA groups 95.5
// (the superseded kernels and the work-queue variant are recorded in profiles/, not kept here).
constexpr int TM = 64;      // rows per item
constexpr int KT = 32;      // K step
struct TrmmItem {
  int p, rb, col0, half;  // PC, 64-row block, first column, 1 = 64-column item
};
constexpr int TRMM_MAX_ITEMS = 64;  // per worker; the schedule falls back to more workers' worth otherwise

// ------------------------------------------------------------------------------------------
// LDS-direct variant of the persistent kernel: the k-tiles are streamed from L2 into LDS by
// global_load_lds_dwordx4 (each wave-load lands as one contiguous 1 KiB chunk, lane l -> bytes 16 l .. 16 l + 15),
// so there is no register staging and no ds_write on the waves' instruction streams.  The LDS tiles are
// unpadded, k-major ([k][64] for W, [k][128] for K_*^T); bank conflicts of the fragment reads are avoided by
// storing the odd k-rows with their 16-double blocks swapped (position = index ^ 16) -- the swizzle is applied
// on the GLOBAL side (which pair a lane fetches), the LDS side of a DMA load is always contiguous.  Two LDS
// buffers as distinct arrays (so that hipcc's waitcnt insertion can tell the buffer being filled from the
// one being read); three buffers: the loads of k-tile s+2 are issued at the start of k-tile s and the barrier
// that ends k-tile s waits (s_waitcnt vmcnt(6), written by hand) only for the loads of k-tile s+1.
typedef const __attribute__((address_space(1))) void *gas_ptr;
typedef __attribute__((address_space(3))) void *las_ptr;

// the six loads of k-tile s+2 go out one by one between the MFMA groups of k-tile s
__global__ __launch_bounds__(512, 2) void trmm_vsq_dma_kernel(
    const double *__restrict__ Wt, const double *__restrict__ KS, double *__restrict__ out,
    const TrmmItem *__restrict__ sched, const int *__restrict__ sched_cnt, int max_items,
    int64_t Npad, int64_t Bcap, int k, int nrb
#ifdef GPEMU_TRMM_STAMPS        // diagnostic build (tools/trmm_balance.py): per-worker time stamps of a launch
    , unsigned long long *__restrict__ stamps
#endif
    ) {
  constexpr int BUFD = KT * TM + KT * TILE;            // one k-tile: [W tile | K_*^T tile], 48 KiB
  __shared__ __attribute__((aligned(16))) double L0[BUFD];
  __shared__ __attribute__((aligned(16))) double L1[BUFD];
  __shared__ __attribute__((aligned(16))) double L2[BUFD];
  __shared__ double red[2][TILE];
  __shared__ TrmmItem s_items[TRMM_MAX_ITEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lk = lane >> 4;

#ifdef GPEMU_TRMM_STAMPS
  if (stamps && tid == 0) stamps[blockIdx.x * 16] = __builtin_amdgcn_s_memrealtime();
#endif
  const int nitems = sched_cnt[blockIdx.x];
  if (tid < nitems) s_items[tid] = sched[(int64_t)blockIdx.x * max_items + tid];
  __syncthreads();
  if (nitems == 0) return;
  const TrmmItem *my = s_items;

  // per-lane source offsets (doubles) of the wave's six 1 KiB chunks of a k-tile
  int offA[2], offB[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kk = 2 * (wave * 2 + j) + (lane >> 5);          // chunk = two k-rows of the W tile
    offA[j] = kk * (int)Npad + ((2 * (lane & 31)) ^ ((kk & 1) << 4));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = wave * 4 + r;                              // chunk = one k-row of the K_*^T tile
    offB[r] = kk * (int)Bcap + ((2 * lane) ^ ((kk & 1) << 4));
  }
  const int64_t astep = (int64_t)KT * Npad, bstep = (int64_t)KT * Bcap;

  // ---- load cursor (one k-tile ahead of the compute cursor) ----
  int l_item = 0, l_t = 0, l_nt = 1;
  const double *l_pa = Wt, *l_pb = KS;
  auto l_open = [&]() {
    const TrmmItem it = my[l_item];
    const int64_t i0 = (int64_t)it.rb * TM;
    l_nt = (int)((i0 + TM + KT - 1) / KT);
    l_pa = Wt + (int64_t)it.p * Npad * Npad + i0;
    l_pb = KS + (int64_t)it.p * Npad * Bcap + (it.col0 & ~(TILE - 1));
    l_t = 0;
  };
  // Issued as inline assembly: with the builtin, hipcc's waitcnt insertion treats every later LDS read as a
  // possible reader of the in-flight destination and drains vmcnt to 0, which defeats the two-tile lookahead;
  // the waits for these loads are the hand-written s_waitcnt vmcnt(6) of tile_barrier().
  auto dma1 = [&](const double *src, double *dst_wave_uniform) {
    const unsigned lds_off = (unsigned)(uintptr_t)((las_ptr)dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off"
                 :
                 : "s"(lds_off), "v"(src)
                 : "memory");
  };
  // one of the wave's six loads of a k-tile (part compile-time after unrolling); the cursor moves on after the last
  auto dma_part = [&](double *dA, int part) {
    double *dB = dA + KT * TM;
    if (part < 2) dma1(l_pa + offA[part], dA + (wave * 2 + part) * 128);
    else dma1(l_pb + offB[part - 2], dB + (wave * 4 + part - 2) * 128);
    if (part == 5 && l_item < nitems) {
      if (++l_t == l_nt) {
        if (++l_item < nitems) l_open();   // else: stay on the last k-tile (harmless re-read)
      } else {
        l_pa += astep;
        l_pb += bstep;
      }
    }
  };
  auto dma = [&](double *dA) {
#pragma unroll
    for (int part = 0; part < 6; ++part) dma_part(dA, part);
  };

  // vmcnt(6) lgkmcnt(0): everything but this wave's six newest loads has landed, every LDS read has returned
  auto tile_barrier = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0076);
    __builtin_amdgcn_s_barrier();
  };
  l_open();
  dma(L0);
  dma(L1);
  tile_barrier();
#ifdef GPEMU_TRMM_STAMPS
  if (stamps && tid == 0) stamps[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime();
#endif

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  int c_item = 0, c_t = 0;
  TrmmItem cur = my[0];
  int c_nt = (int)(((int64_t)cur.rb * TM + TM + KT - 1) / KT);
  const int sw = (lk & 1) << 4;                                 // this lane's k-rows are odd <=> lk odd
  const int ia0 = lk * TM + ((wm * 32 + lr) ^ sw), ia1 = lk * TM + ((wm * 32 + 16 + lr) ^ sw);

  auto step = [&](const double *cA, double *nA) -> bool {
    const double *cB = cA + KT * TM;
    if (cur.half) {
      const int ib = lk * TILE + (((cur.col0 & 64) + wn * 16 + lr) ^ sw);
      double a[2][2], b[2];
      a[0][0] = cA[ia0];
      a[0][1] = cA[ia1];
      b[0] = cB[ib];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (ks + 1 < KT / 4) {
          a[nx][0] = cA[(ks + 1) * 4 * TM + ia0];
          a[nx][1] = cA[(ks + 1) * 4 * TM + ia1];
          b[nx] = cB[(ks + 1) * 4 * TILE + ib];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cu][mi], b[cu], acc[mi][0], 0, 0, 0);
        if (ks < 6) dma_part(nA, ks);   // k-tile +2 -> the buffer k-tile -1 was read from
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const int ib0 = lk * TILE + ((wn * 32 + lr) ^ sw), ib1 = lk * TILE + ((wn * 32 + 16 + lr) ^ sw);
      double a[2][2], b[2][2];
      a[0][0] = cA[ia0];
      a[0][1] = cA[ia1];
      b[0][0] = cB[ib0];
      b[0][1] = cB[ib1];
#pragma unroll
      for (int ks = 0; ks < KT / 4; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (ks + 1 < KT / 4) {
          a[nx][0] = cA[(ks + 1) * 4 * TM + ia0];
          a[nx][1] = cA[(ks + 1) * 4 * TM + ia1];
          b[nx][0] = cB[(ks + 1) * 4 * TILE + ib0];
          b[nx][1] = cB[(ks + 1) * 4 * TILE + ib1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cu][mi], b[cu][ni], acc[mi][ni], 0, 0, 0);
        if (ks < 6) dma_part(nA, ks);   // k-tile +2 -> the buffer k-tile -1 was read from
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    tile_barrier();
    if (++c_t == c_nt) {
      const int ncols = cur.half ? 64 : TILE;
      const int wcol = cur.half ? wn * 16 : wn * 32;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (ni == 0 || !cur.half) {
          double sq = 0.0;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq = fma(acc[mi][ni][r], acc[mi][ni][r], sq);
          sq += __shfl_xor(sq, 16);
          sq += __shfl_xor(sq, 32);
          if (lk == 0) red[wm][wcol + ni * 16 + lr] = sq;
        }
      }
      __syncthreads();
      if (tid < ncols)
        out[(((int64_t)cur.col0 + tid) * k + cur.p) * nrb + cur.rb] = red[0][tid] + red[1][tid];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};
#ifdef GPEMU_TRMM_STAMPS
      if (stamps && tid == 0 && c_item < 13) stamps[blockIdx.x * 16 + 2 + c_item] = __builtin_amdgcn_s_memrealtime();
#endif
      if (++c_item == nitems) return true;
      cur = my[c_item];
      c_nt = (int)(((int64_t)cur.rb * TM + TM + KT - 1) / KT);
      c_t = 0;
    }
    return false;
  };
  for (;;) {
    if (step(L0, L2)) break;
    if (step(L1, L0)) break;
    if (step(L2, L1)) break;
  }
}

// host side: LPT schedule of the items of one launch shape (cached per model and column-tile count)
// keep (probes only, tools/share_probe.hip): restrict the launch to the (PC, 64-row block) pairs it accepts
static void build_trmm_schedule(gpemu_model *m, int ncb, std::vector<TrmmItem> &flat, std::vector<int> &cnt,
                                int &max_items, int &nworkers, bool (*keep)(int p, int rb) = nullptr, int split_all = 0) {
  const int nrb = (int)m->vsq_nrb, k = (int)m->k;
  struct It { double cost; TrmmItem it; };
  std::vector<It> items;
  // row blocks with short K are issued as two 64-column halves; with one or two column blocks (B <= 256) there
  // are too few items for 256 workers unless every row block is
  const int split_below = (ncb <= 2 || split_all) ? nrb : nrb / 4;
  // per-item cost of the epilogue, in k-tiles.  Measured (in-kernel stamps, profiles/r03_trmm_balance.txt): workers with
  // 2 / 3 / 5 items finish at 85.3 / 87.6 / 88.5 us, but raising it to 0.4-0.9 (3 / 4 items everywhere) leaves the slowest
  // worker at 91 us: the spread is the XCDs' (means 86.6-88.2 us), not the cost model's
  const double ov = 0.15;
  for (int rb = 0; rb < nrb; ++rb) {
    const double nt = (double)(((int64_t)rb * TM + TM + KT - 1) / KT);
    for (int p = 0; p < k; ++p)
      for (int cb = 0; cb < ncb; ++cb) {
        if (keep && !keep(p, rb)) continue;
        if (rb < split_below) {
          items.push_back({0.5 * nt + ov, TrmmItem{p, rb, cb * TILE, 1}});
          items.push_back({0.5 * nt + ov, TrmmItem{p, rb, cb * TILE + 64, 1}});
        } else {
          items.push_back({nt + ov, TrmmItem{p, rb, cb * TILE, 0}});
        }
      }
  }
  std::stable_sort(items.begin(), items.end(), [](const It &a, const It &b) { return a.cost > b.cost; });
  const int ncu = m->num_cu;
  nworkers = ncu < (int)items.size() ? ncu : (int)items.size();
  std::vector<std::vector<TrmmItem>> per(nworkers);
  std::vector<double> load(nworkers, 0.0);
  // XCD-aware placement.  Workgroups are dispatched round-robin over the 8 XCDs (worker w runs on XCD w % 8),
  // each with its own 4 MiB L2.  All items of one (PC, column block) group read the same K_*^T block and the
  // groups of one PC read the same W_p; every group costs the same, so when the groups divide evenly they are
  // dealt to the XCDs in (PC, column block) order -- an XCD then works on at most two or three PCs, and the
  // column-block items of one (PC, row block), equally long, start together and stream the same W rows through
  // that L2 -- and LPT runs within each XCD's workers.  Otherwise: plain LPT over all workers.
  const int nxcd = 8, ngroups = k * ncb;
  if (!keep && nworkers == ncu && nworkers % nxcd == 0 && ngroups % nxcd == 0) {
    const int gper = ngroups / nxcd;
    for (const It &x : items) {
      const int g = x.it.p * ncb + x.it.col0 / TILE;
      const int xcd = g / gper;
      int best = xcd;
      for (int w = xcd; w < nworkers; w += nxcd)
        if (load[w] < load[best]) best = w;
      per[best].push_back(x.it);
      load[best] += x.cost;
    }
  } else {
    for (const It &x : items) {  // longest processing time first onto the least loaded worker
      int best = 0;
      for (int w = 1; w < nworkers; ++w)
        if (load[w] < load[best]) best = w;
      per[best].push_back(x.it);
      load[best] += x.cost;
    }
  }
  max_items = 1;
  for (auto &v : per) max_items = v.size() > (size_t)max_items ? (int)v.size() : max_items;
  flat.assign((size_t)nworkers * max_items, TrmmItem{0, 0, 0, 0});
  cnt.assign(nworkers, 0);
  for (int w = 0; w < nworkers; ++w) {
    cnt[w] = (int)per[w].size();
    for (size_t i = 0; i < per[w].size(); ++i) flat[(size_t)w * max_items + i] = per[w][i];
  }
}

int launch_trmm_vsq(gpemu_model *m, int64_t B, hipStream_t st) {
  Workspace &w = m->ws;
  constexpr int64_t smallb_max = 128;
  const int64_t Bv = m->variant_B > 0 ? m->variant_B : B;   // a chain stacked with others is evaluated as it would be alone
  const int64_t per = trmm_piece_cols(m, B);
  if (per < round_up(B, TILE)) {     // one launch per piece; the column partials of a piece land where a single launch would put them
    double *const KS0 = w.KS, *const V0 = w.vsq_part;
    int rc = GPEMU_OK;
    for (int64_t c0 = 0; c0 < B && rc == GPEMU_OK; c0 += per) {
      w.KS = KS0 + c0;
      w.vsq_part = V0 + c0 * m->k * m->vsq_nrb;
      rc = launch_trmm_vsq(m, std::min(per, B - c0), st);
    }
    w.KS = KS0;
    w.vsq_part = V0;
    return rc;
  }
  if (Bv <= smallb_max) {  // small batch: persistent 32 x 32 items, operands straight into registers
    const int rc = launch_trmm_vsq_small(m, B, st);
    if (rc != GPEMU_ERR_UNSUPPORTED) return rc;
    if (m->variant_B > 0) {
      set_error("stacked chains: the small-batch triangular GEMM cannot take %lld columns", (long long)B);
      return rc;
    }
    // a shape with more items per worker than the small-batch kernel holds: the large-batch kernel with every item halved
  }
  w.cur_nrb = (int)m->vsq_nrb;
  const int nrb = (int)m->vsq_nrb;
  const int ncb = (int)(round_up(B, TILE) / TILE);  // only the column tiles that hold real queries
  const int cap = m->num_cu;
  if (m->sched_ncb != ncb || m->sched_cap != cap) {
    const gpemu_model::SchedEntry *hit = nullptr;
    for (const auto &e : m->sched_cache)
      if (e.ncb == ncb && e.cap == cap) hit = &e;
    if (!hit) {
      std::vector<TrmmItem> flat;
      std::vector<int> cnt;
      int max_items = 0, nworkers = 0;
      build_trmm_schedule(m, ncb, flat, cnt, max_items, nworkers);
      if (max_items > TRMM_MAX_ITEMS) {
        set_error("triangular GEMM schedule needs %d items per worker (limit %d): batch too large for one launch",
                  max_items, TRMM_MAX_ITEMS);
        return GPEMU_ERR_UNSUPPORTED;
      }
      // schedules are kept (a few KB each): launches in flight keep reading the one they were given
      gpemu_model::SchedEntry e{ncb, cap, nullptr, nullptr, max_items, nworkers};
      GP_HIP(hipMalloc(&e.items, sizeof(TrmmItem) * flat.size()));
      GP_HIP(hipMalloc((void **)&e.cnt, sizeof(int) * cnt.size()));
      GP_HIP(hipMemcpy(e.items, flat.data(), sizeof(TrmmItem) * flat.size(), hipMemcpyHostToDevice));
      GP_HIP(hipMemcpy(e.cnt, cnt.data(), sizeof(int) * cnt.size(), hipMemcpyHostToDevice));
      m->sched_cache.push_back(e);
      hit = &m->sched_cache.back();
    }
    m->sched_items = hit->items; m->sched_cnt = hit->cnt;
    m->sched_ncb = ncb; m->sched_cap = cap; m->sched_max_items = hit->max_items; m->sched_workers = hit->workers;
  }
  const int pe0 = prof_mark(m, st);
#ifdef GPEMU_TRMM_STAMPS
  // diagnostic build only (make STAMPS=1; tools/trmm_balance.py): per-worker time stamps of launch 600, dumped to the
  // file GPEMU_TRMM_STAMP_FILE names.  The product build has neither the state nor the getenv in its launch path.
  static const char *stamp_path = getenv("GPEMU_TRMM_STAMP_FILE");
  static unsigned long long *dstamps = nullptr;
  static int stamp_calls = 0;
  if (stamp_path && !dstamps) {
    GP_HIP(hipMalloc((void **)&dstamps, sizeof(unsigned long long) * 16 * 1024));
    GP_HIP(hipMemset(dstamps, 0, sizeof(unsigned long long) * 16 * 1024));
  }
  hipLaunchKernelGGL(trmm_vsq_dma_kernel, dim3((unsigned)m->sched_workers), dim3(512), 0, st, m->Wt,
                     w.KS, w.vsq_part, (const TrmmItem *)m->sched_items, m->sched_cnt, m->sched_max_items,
                     m->Npad, w.Bcap, (int)m->k, nrb, dstamps);
  GP_HIP(hipGetLastError());
  prof_pair(m, 0, pe0, prof_mark(m, st));
  if (stamp_path && ++stamp_calls == 600) {
    GP_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(16 * 1024);
    GP_HIP(hipMemcpy(h.data(), dstamps, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    std::vector<int> cnt(m->sched_workers);
    GP_HIP(hipMemcpy(cnt.data(), m->sched_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
    if (FILE *f = fopen(stamp_path, "w")) {
      unsigned long long t0 = ~0ull;
      for (int wk = 0; wk < m->sched_workers; ++wk) t0 = std::min(t0, h[wk * 16]);
      for (int wk = 0; wk < m->sched_workers; ++wk) {
        fprintf(f, "%d %d", wk, cnt[wk]);
        for (int i = 0; i < 2 + std::min(cnt[wk], 12); ++i) fprintf(f, " %.2f", (double)(h[wk * 16 + i] - t0) / 100.0);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#else
  hipLaunchKernelGGL(trmm_vsq_dma_kernel, dim3((unsigned)m->sched_workers), dim3(512), 0, st, m->Wt,
                     w.KS, w.vsq_part, (const TrmmItem *)m->sched_items, m->sched_cnt, m->sched_max_items,
                     m->Npad, w.Bcap, (int)m->k, nrb);
  GP_HIP(hipGetLastError());
  prof_pair(m, 0, pe0, prof_mark(m, st));
#endif
  return GPEMU_OK;
}

// ------------------------------------------------------------------------------------------
__global__ void reduce_mean_var_kernel(const double *__restrict__ mean_part,
                                       const double *__restrict__ vsq_part,
                                       const double *__restrict__ kdiag, double *__restrict__ mean,
                                       double *__restrict__ var, int64_t B, int64_t Bcap, int k,
                                       int nchunk, int nrb) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * k) return;
  int p = (int)(idx / B);
  int64_t b = idx - (int64_t)p * B;
  double mu = 0.0, vs = 0.0;
  for (int c = 0; c < nchunk; ++c) mu += mean_part[(b * k + p) * nchunk + c];
  for (int r = 0; r < nrb; ++r) vs += vsq_part[(b * k + p) * nrb + r];
  double v = kdiag[p] - vs;
  if (v < 0.0) v = 0.0;     // skl _gpr.py:479-485
  double sd = sqrt(v);      // predict returns std (skl _gpr.py:494) ...
  mean[b * k + p] = mu;
  var[b * k + p] = sd * sd; // ... and the reference squares it again (ref: emulation.py:499)
}

int launch_reduce_mean_var(gpemu_model *m, int64_t B, double *dmean, double *dvar, hipStream_t st) {
  const Workspace &w = m->ws;
  int64_t n = B * m->k;
  hipLaunchKernelGGL(reduce_mean_var_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     w.mean_part, w.vsq_part, m->kdiag, dmean, dvar, B, w.Bcap, (int)m->k,
                     w.cur_nchunk, w.cur_nrb);
  GP_HIP(hipGetLastError());
  return GPEMU_OK;
}

}  // namespace gpemu
