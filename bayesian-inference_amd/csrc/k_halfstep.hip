// Small emulators (at most 256 design points): cross-kernel AND triangular GEMM of a block of proposals in ONE workgroup.
//
// The reference's own analysis (ref: config/jet_substructure.yaml:243-271) has ~150 design points, emulation groups of
// 5 / 11 / 25 PCs and 100 - 200 walkers.  There a stretch-move half-step is three launches of 10 - 16 us each whose
// arithmetic would take 2 us: every stage is a chain of latencies (launch, the proposal's dependent loads, K_*^T through
// memory, partial sums through memory), and the general kernels pad N to 128 (150 -> 256: more than half of the small
// GEMM's k-tiles multiply zeros).  For N <= 256 the whole K dimension of a (PC, 32 proposals) pair fits one workgroup:
//
//   workgroup (PC p of group g, column block cb):   512 threads, one per CU
//     0  requests that need no proposal: W_p^T fragments of the first k-tiles, the cross-kernel's training fragments
//     1  stretch proposal of the block's 32 walkers (ref: emcee moves/stretch.py) -> LDS
//     2  K_*^T[j][b] for the N real rows on the matrix cores (predict_dev.h), kept in LDS -- never written to memory;
//        per 16-row tile the partial mean  sum_j alpha_j K_*[j][b]
//     3  V = W_p K_*^T by 32-row blocks, k-tiles up to the diagonal, K split over the waves in the eight slices of
//        k_trmm_small.hip (two per wave, one exchange per block); per block the column sums of V^2
//     4  the partial sums are added IN THE ORDER the three-launch path adds them (cross-kernel: two 16-row tiles per
//        32-row chunk, then the likelihood's walker_mean_sd order; GEMM: 32-row blocks, same order): mean and
//        ||W k_*||^2 of (b, p) are final and go to mean_part / vsq_part as ONE chunk / ONE row block
//   then the likelihood launch of the three-launch path, unchanged, reading single partials.
//
// Every value is formed by the device functions and in the order of kstar_kernel<., 2, 1, 2> / trmm_vsq_small_kernel /
// walker_mean_sd; rows and k-tiles that are padding contribute exact zeros there and are skipped here: the chain is the
// three-launch path's BIT FOR BIT (tests/test_gpu_sampler.py::test_small_emulators_..., test_gpu_shipped.py).  No
// workgroup waits for another, so nothing depends on how many are resident (several ranks rehearsing on one device).
//
// Shipped shape (150 design points, 5 + 11 + 25 PCs, 200 walkers): 84.2 -> 54.7 us per step.  What was measured on the
// way -- six forms of the GEMM phase, and the likelihood + accept behind tickets in the SAME launch (slower: signalling
// between XCDs goes through memory) -- is in profiles/r05_halfstep_small.txt and DESIGN.md 4.17.
#include <algorithm>
#include <atomic>

#include "internal.h"
#include "kstar_host.h"
#include "predict_dev.h"

namespace gpemu {
static std::atomic<int64_t> g_halfstep_launches{0};
#ifdef GPEMU_HS_STAMPS          // diagnostic build (make HS_STAMPS=1; tools/hs_stamps.py): clock64 at the phase boundaries, wave 0 of 3 workgroups
__device__ long long g_hs_stamps[3][32];
__device__ long long g_hs_wall[512][4];     // per workgroup: wall_clock64 (100 MHz, one counter for the chip) at start / - / end
#define HS_WALL(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_hs_wall[blockIdx.x][i] = wall_clock64(); } while (0)
#define HS_STAMP(i) do { if (lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 64 || blockIdx.x == gridDim.x - 9)) g_hs_stamps[blockIdx.x == 0 ? 0 : (blockIdx.x == 64 ? 1 : 2)][i] = clock64(); } while (0)
#else
#define HS_STAMP(i) do { } while (0)
#define HS_WALL(i) do { } while (0)
#endif

typedef double hd4 __attribute__((ext_vector_type(4)));
typedef double hd2 __attribute__((ext_vector_type(2)));

constexpr int HS_COLS = 32;          // proposals per workgroup (the small-batch GEMM's column block)
constexpr int HS_LDK = 34;           // leading dimension of K_*^T in LDS (doubles): 272-byte rows, 16-byte aligned
constexpr int HS_RING = 4;           // k-tiles of W^T fragments in flight per wave
constexpr int HS_NMAX = 256;         // design points at most
constexpr int HS_GROUPS_MAX = 8;

struct HsGroup {
  const double *Xa, *alf, *qsc, *qof, *constv, *Xs, *inv_ls, *Wt;
  double *mean_part, *vsq_part;      // [Bcap][k]: written as one chunk / one row block per (proposal, PC)
  int64_t N, Npad;
  int k, has_const, pc0;             // pc0: index of the group's first PC in the launch's list of all PCs
  int subs;                          // lanes that share a PC's partial sums in walker_mean_sd: 4 for k <= 16, else 2
};
struct HsArgs {
  HsGroup g[HS_GROUPS_MAX];
  int ng, ktot, ncb, d;
  int64_t B;
  const double *etab;
  double *Xq;                        // [..][DPAD] padded query rows (read, or written by the first PC's workgroups)
};

template <int KIND>
__global__ __launch_bounds__(512) void halfstep_small_kernel(HsArgs ha, ProposeArgs pa) {
  constexpr int KS = 2, TB = KSTAR_TB;
  HS_WALL(0);
  extern __shared__ __attribute__((aligned(16))) double s_K[];              // [32 ceil(N / 32)][HS_LDK]
  __shared__ __attribute__((aligned(16))) double s_ex[2 * 2 * 3 * 2 * 64 * 4];   // K-slice sums on their way to wave g = 0: two buffers of 24 KiB
  __shared__ double s_sq[(HS_NMAX / 32) * 4 * 64];                          // per 32-row block, tile and lane: V^2 summed over the lane's rows
  __shared__ double s_tab[1 << TB];
  __shared__ __attribute__((aligned(16))) double s_q[HS_COLS * DPAD];
  __shared__ double s_t[(HS_NMAX / 16) * HS_COLS];                          // partial mean per 16-row tile
  __shared__ double s_vsq[(HS_NMAX / 32) * HS_COLS];                        // sum of V^2 per 32-row block

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup i runs on XCD i % 8: the column blocks of one PC share an XCD, whose L2 then holds W_p^T once and keeps it
  // from half-step to half-step (41 PCs x 150^2 / 2 doubles = 3.7 MB over eight 4 MB L2s)
  HS_STAMP(0);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int P = (slot / ha.ncb) * 8 + xcd, cb = slot % ha.ncb;
  if (P >= ha.ktot) return;
  int gi = 0;
  while (gi + 1 < ha.ng && P >= ha.g[gi + 1].pc0) ++gi;
  const HsGroup &gr = ha.g[gi];
  const int p = P - gr.pc0;
  const int64_t N = gr.N, Npad = gr.Npad;
  const int njt = (int)((N + 15) / 16), nt32 = (int)((N + 31) / 32);
  const int64_t njt_pad = Npad / 16;
  const int q = lane & 15, lk = lane >> 4;

  // ---- 0: what needs no proposal ----------------------------------------------------------------------------------
  // Triangular GEMM, 32-row blocks: wave w = (y, g) = (w & 1, w >> 1) owns the column tile y (columns 2 n + y) of both row
  // tiles (rows 32 rb + 2 i + x) for the K-slices g and g + 4 of k_trmm_small.hip (slice s = k-rows 32 t + 4 s .. + 3 of
  // k-tile t: that kernel's wave s, the same MFMA sequences) -- so "slice g + slice g + 4", that kernel's first sum, is
  // formed in registers, and one exchange through LDS per block (waves g > 0 -> wave g = 0) finishes the other three.
  // W^T fragments three k-tiles ahead: lane (q, lk) takes W^T[32 t + 4 s + lk][32 rb + 2 q, + 1] for its two slices.
  const int gy = wave & 1, gg = wave >> 1;
  const double *Wp = gr.Wt + (int64_t)p * Npad * Npad + (int64_t)(4 * gg + lk) * Npad + 2 * q;
  // (every call loads -- beyond the last k-tile the last one again -- so that the number of loads in flight is known at
  // compile time; the cursor is three 32-bit scalars moved by selects: the k-tile's ~60 scalar instructions of address
  // arithmetic in 64 bits were what the loop spent its time on, 350 of 750 clocks per k-tile with the MFMAs taken out)
  const int npad = (int)Npad;
  const double *Wp2 = Wp + 16 * npad;
  int l_rb = 0, l_t = 0, l_off = 0;
  auto issue = [&](hd2 (&r)[2]) {
    r[0] = *reinterpret_cast<const hd2 *>(Wp + l_off);
    r[1] = *reinterpret_cast<const hd2 *>(Wp2 + l_off);
    const bool in_block = l_t < l_rb, next_block = !in_block && (l_rb + 1 < nt32);
    l_rb = next_block ? l_rb + 1 : l_rb;
    l_t = in_block ? l_t + 1 : (next_block ? 0 : l_t);
    l_off = in_block ? l_off + 32 * npad : (next_block ? 32 * l_rb : l_off);
  };
  hd2 ring[HS_RING][2];
#pragma unroll
  for (int u = 0; u < HS_RING; ++u) issue(ring[u]);
  // cross-kernel: wave w forms the tiles (j-tile (w >> 1) + 4 u, column tile w & 1), u < 4
  const int bt = wave & 1, jt0 = wave >> 1;
  double a[4][KS];
  kd4 al[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int jt = jt0 + 4 * u;
#pragma unroll
    for (int s = 0; s < KS; ++s) a[u][s] = 0.0;
    al[u] = kd4{0.0, 0.0, 0.0, 0.0};
    if (jt < njt) {
#pragma unroll
      for (int s = 0; s < KS; ++s) a[u][s] = gr.Xa[(((int64_t)p * njt_pad + jt) * KS + s) * 64 + lane];
      al[u] = *reinterpret_cast<const kd4 *>(gr.alf + ((int64_t)p * njt_pad + jt) * 16 + lk * 4);
    }
  }
  double sc[KS], of[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) { sc[s] = gr.qsc[p * 4 * KS + 4 * s + lk]; of[s] = gr.qof[p * 4 * KS + 4 * s + lk]; }
  const double cst = gr.has_const ? gr.constv[p] : 0.0;

  // ---- 1: the block's query rows -----------------------------------------------------------------------------------
  if (tid < (1 << TB)) s_tab[tid] = ha.etab[tid];
  for (int i = tid; i < (HS_NMAX / 16) * HS_COLS; i += 512) s_t[i] = 0.0;
  if (tid < (HS_NMAX / 32) * HS_COLS) s_vsq[tid] = 0.0;
  for (int i = njt * 16 * HS_LDK + tid; i < nt32 * 32 * HS_LDK; i += 512) s_K[i] = 0.0;   // rows a k-tile reads beyond the last j-tile
  if (tid < HS_COLS * DPAD) {
    const int col = tid & (HS_COLS - 1), comp = tid >> 5;
    const int64_t b = (int64_t)cb * HS_COLS + col;
    const bool keeper = (P == 0);
    double qv = 0.0;
    if (pa.enabled) {
      if (b < pa.n) {
        const int w = pa.idx_s[b], j = pa.partner[b];
        const double z = pa.zz[b];
        if (comp < pa.d) {
          const double cj = pa.X[(int64_t)j * DPAD + comp], sw = pa.X[(int64_t)w * DPAD + comp];
          qv = cj - (cj - sw) * z;                     // emcee moves/stretch.py get_proposal
        }
        if (keeper && comp == 0) pa.factors[b] = (pa.d - 1.0) * log(z);
      }
    } else if (pa.raw) {
      if (b < pa.n && comp < pa.d) qv = pa.raw[b * pa.d + comp];
    } else {
      qv = ha.Xq[b * DPAD + comp];
    }
    if (keeper && (pa.enabled || pa.raw)) ha.Xq[b * DPAD + comp] = qv;
    s_q[col * DPAD + comp] = qv;
  }
  __syncthreads();
  HS_STAMP(1);

  // ---- 2: K_*^T tiles into LDS, partial means per tile (kstar_mfma_block with one j-tile per wave: the same values) --
  {
    const int col = bt * 16 + q;
    double bq[KS];
    double part = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int comp = 4 * s + lk;
      const double qv = (comp < 8) ? s_q[col * 8 + comp] : 0.0;
      const double v = fma(qv, sc[s], of[s]);
      bq[s] = v;
      part = (comp < ha.d) ? fma(v, v, part) : part;
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    const double hq = (KIND == 0) ? -0.5 * part : part;
    KstarDirect dir{nullptr, nullptr};
    if (KIND == 1) dir = KstarDirect{gr.Xs + (int64_t)p * Npad * DPAD, gr.inv_ls + p * DPAD};
    // the products of all of the wave's tiles first: their matrix-core latency hides behind the first tile's ~100 vector
    // instructions (padding tiles multiply zeros and are dropped below)
    kd4 accs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) accs[u] = kstar_tile_product<KS>(a[u], bq);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jt = jt0 + 4 * u;
      if (jt < njt) {                                       // (wave-uniform)
        const int64_t row0 = (int64_t)jt * 16 + lk;
        kd4 v = kstar_value4<KIND, TB>(accs[u], hq, s_tab, dir, s_q, row0, col);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += cst;
        if ((jt + 1) * 16 > N) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + 4 * r >= N) v[r] = 0.0;
        }
        double macc = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s_K[(row0 + 4 * r) * HS_LDK + col] = v[r];
          macc = fma(al[u][r], v[r], macc);
        }
        double s = macc;
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (hq != hq) s = hq;                               // a NaN in the query: the column's mean says so
        if (lk == 0) s_t[jt * HS_COLS + col] = s;
      }
    }
  }
  __syncthreads();
  HS_STAMP(2);

  // ---- 3: V = W_p K_*^T by 32-row blocks; per block the column sums of V^2 (trmm_vsq_small_body: the same order) -----
  {
    const double *Kl = s_K + (4 * gg + lk) * HS_LDK + 2 * q + gy;
    hd4 acc[2][2];                                           // [slice g / g + 4][row tile x]
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int x = 0; x < 2; ++x) acc[sl][x] = hd4{0.0, 0.0, 0.0, 0.0};
    hd4 pt0, pt1;
    auto finish_block = [&](int rb, const hd4 &t0, const hd4 &t1) {
      const hd4 *ex = reinterpret_cast<const hd4 *>(s_ex) + (size_t)(rb & 1) * (2 * 3 * 2 * 64);
      hd4 v0 = t0, v1 = t1;
#pragma unroll
      for (int g2 = 0; g2 < 3; ++g2) {
        v0 = v0 + ex[((gy * 3 + g2) * 2 + 0) * 64 + lane];
        v1 = v1 + ex[((gy * 3 + g2) * 2 + 1) * 64 + lane];
      }
      s_sq[(rb * 4 + 0 + 2 * gy) * 64 + lane] = fma(v0[0], v0[0], v0[1] * v0[1]) + fma(v0[2], v0[2], v0[3] * v0[3]);
      s_sq[(rb * 4 + 1 + 2 * gy) * 64 + lane] = fma(v1[0], v1[0], v1[1] * v1[1]) + fma(v1[2], v1[2], v1[3] * v1[3]);
    };
    const int nsteps = nt32 * (nt32 + 1) / 2;                // k-tiles (rb, t), t <= rb, in the order the fragments were requested
    int c_rb = 0, c_t = 0, c_boff = 0;
    double b0 = Kl[0], b1 = Kl[16 * HS_LDK];
    for (int s0 = 0; s0 < nsteps; s0 += HS_RING) {
#pragma unroll
      for (int u = 0; u < HS_RING; ++u) {
        // (the ring slots are fixed registers: a slot is read here and refilled at once -- rotating the fragments through
        // register moves made every move wait for the newest load: 830 instead of ~520 clocks per k-tile)
        const hd2 a0 = ring[u][0], a1 = ring[u][1];
        issue(ring[u]);
        if (s0 + u < nsteps) {                               // (wave-uniform)
          const double c0 = b0, c1 = b1;
          const bool last_t = (c_t == c_rb);
          c_boff = last_t ? 0 : c_boff + 32 * HS_LDK;        // the next k-tile's K_*^T fragments while the matrix cores work
          b0 = Kl[c_boff];
          b1 = Kl[c_boff + 16 * HS_LDK];
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            acc[0][x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[x], c0, acc[0][x], 0, 0, 0);
            acc[1][x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[x], c1, acc[1][x], 0, 0, 0);
          }
          if (last_t) {
            // slices g and g + 4 here; waves g > 0 hand their sums to wave g = 0 of the same column tile, which adds them
            // in turn (k_trmm_small.hip: ((t0 + t1) + t2) + t3), squares, and sums the register pair and the two pairs of
            // a lane (that kernel's xor-4 step); the lane groups and the two row tiles follow in phase 4.  Two exchange
            // buffers in turn: one barrier per block.  (Wave g = 0 finishing the block BEHIND the next k-tile's MFMAs was
            // measured: slower -- double-precision vector instructions beside another wave's f64 MFMAs on one SIMD halve
            // both, profiles/r01_fp64_rates.txt.)
            hd4 *ex = reinterpret_cast<hd4 *>(s_ex) + (size_t)(c_rb & 1) * (2 * 3 * 2 * 64);
            pt0 = acc[0][0] + acc[1][0];
            pt1 = acc[0][1] + acc[1][1];
            if (gg > 0) {
              ex[((gy * 3 + gg - 1) * 2 + 0) * 64 + lane] = pt0;
              ex[((gy * 3 + gg - 1) * 2 + 1) * 64 + lane] = pt1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (gg == 0) finish_block(c_rb, pt0, pt1);
            HS_STAMP(8 + c_rb);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
              for (int x = 0; x < 2; ++x) acc[sl][x] = hd4{0.0, 0.0, 0.0, 0.0};
            c_t = 0;
            ++c_rb;
          } else {
            ++c_t;
          }
        }
      }
    }
  }
  __syncthreads();
  HS_STAMP(3);

  // ---- 4: the partial sums in the order of the three-launch path (kstar_kernel's chunks of two tiles; walker_mean_sd) --
  // ||W k_*||^2 per 32-row block first: over the four lane groups (the xor-8 and xor-16 steps of k_trmm_small.hip), then
  // over the two row tiles (xor 32); thread (rb, column)
  if (tid < nt32 * HS_COLS) {
    const int rb = tid >> 5, col = tid & 31, y = col & 1, n = col >> 1;
    const double *t0 = s_sq + ((rb * 4 + 2 * y) * 64) + n, *t1 = t0 + 64;
    const double a0 = (t0[0] + t0[16]) + (t0[32] + t0[48]), a1 = (t1[0] + t1[16]) + (t1[32] + t1[48]);
    s_vsq[rb * HS_COLS + col] = a0 + a1;
  }
  __syncthreads();
  // wave 0: the means, wave 1: ||W k_*||^2; lane = (half, column): half h sums the partials of lanes sub = h and h + 2 of
  // walker_mean_sd's order, the halves meet by one shuffle.  Chunks / blocks beyond the real rows are +0.0: skipped.
  if (wave < 2) {
    const int col = lane & 31, half = lane >> 5;
    const int64_t b = (int64_t)cb * HS_COLS + col;
    const int subs = gr.subs;
    const int nreal = (wave == 0) ? (njt + 1) / 2 : nt32;      // chunks of two 16-row tiles / 32-row blocks with real rows
    const double *src = (wave == 0) ? s_t + col : s_vsq + col;
    const int st0 = (wave == 0) ? 2 * HS_COLS : HS_COLS, st1 = (wave == 0) ? HS_COLS : 0;
    auto part = [&](int i) { return (wave == 0) ? src[i * st0] + src[i * st0 + st1] : src[i * st0]; };
    double total;
    if (subs == 4) {
      double sa = 0.0, sb = 0.0;
      for (int i = half; i < nreal; i += 4) sa += part(i);
      for (int i = half + 2; i < nreal; i += 4) sb += part(i);
      sa += __shfl_xor(sa, 32);
      sb += __shfl_xor(sb, 32);
      total = sa + sb;                                          // (m0 + m1) + (m2 + m3)
    } else {
      double sa = 0.0;
      for (int i = half; i < nreal; i += 2) sa += part(i);
      total = sa + __shfl_xor(sa, 32);                          // m0 + m1
    }
    if (half == 0 && b < ha.B) {
      if (wave == 0) gr.mean_part[b * gr.k + p] = total;
      else gr.vsq_part[b * gr.k + p] = total;
    }
  }
  HS_STAMP(4);
  HS_WALL(2);
}

// cross-kernel + triangular GEMM of B <= 128 query rows for ng groups of at most 256 design points in one launch; the
// groups' workspaces then hold ONE partial per (row, PC) (cur_nchunk = cur_nrb = 1) for the likelihood launch.
// GPEMU_ERR_UNSUPPORTED (nothing launched, no error set) where the shape does not fit: the caller takes the general path.
int launch_halfstep_small(gpemu_model *const *ms, int ng, int64_t B, double *dXq, hipStream_t st, const ProposeArgs *pa) {
  if (ng < 1 || ng > HS_GROUPS_MAX || B < 1 || B > 128) return GPEMU_ERR_UNSUPPORTED;
  const gpemu_model *m0 = ms[0];
  int nt32max = 0;
  for (int g = 0; g < ng; ++g) {
    const gpemu_model *m = ms[g];
    const int64_t Bv = m->variant_B > 0 ? m->variant_B : B;
    if (m->Npad > HS_NMAX || m->k > 32 || m->ksteps != 2 || Bv > 128 || m->d != m0->d || kstar_kind(m) != kstar_kind(m0) ||
        m->device != m0->device || m->profiling)
      return GPEMU_ERR_UNSUPPORTED;
    nt32max = std::max(nt32max, (int)((m->N + 31) / 32));
  }
  // few (PC, block) pairs: the general path spreads the same work over more, shorter workgroups (one group of 11 PCs, 200
  // walkers: 44 pairs, 41.1 us per step here against 38.6; 25 PCs: 100 pairs, 53.3 against 63.5)
  {
    int ktot = 0;
    for (int g = 0; g < ng; ++g) ktot += (int)ms[g]->k;
    const char *e = getenv("GPEMU_HALFSTEP_MIN_PAIRS");
    const int min_pairs = e ? atoi(e) : 64;
    if (ktot * (int)((B + HS_COLS - 1) / HS_COLS) < min_pairs) return GPEMU_ERR_UNSUPPORTED;
  }
  HsArgs ha;
  memset(&ha, 0, sizeof(ha));
  ha.ng = ng;
  ha.d = (int)m0->d;
  ha.B = B;
  ha.etab = m0->etab;
  ha.Xq = dXq;
  ha.ncb = (int)((B + HS_COLS - 1) / HS_COLS);
  int pc0 = 0;
  for (int g = 0; g < ng; ++g) {
    gpemu_model *m = ms[g];
    Workspace &w = m->ws;
    ha.g[g] = HsGroup{m->Xa, m->alf, m->qsc, m->qof, m->constv, m->Xs, m->inv_ls, m->Wt, w.mean_part, w.vsq_part,
                      m->N, m->Npad, (int)m->k, m->has_const, pc0, m->k <= 16 ? 4 : 2};
    pc0 += (int)m->k;
    w.cur_nchunk = 1;
    w.cur_nrb = 1;
  }
  ha.ktot = pc0;
  const int nprod = (ha.ktot + 7) / 8 * 8 * ha.ncb;
  const ProposeArgs pargs = pa ? *pa : ProposeArgs();
  const size_t shm = sizeof(double) * (size_t)nt32max * 32 * HS_LDK;
  const dim3 grid((unsigned)nprod), block(512);
#define GP_LAUNCH_HS(KD)                                                                                                  \
  do {                                                                                                                    \
    static bool allowed[64] = {false};       /* per device: the kernel's LDS goes beyond the default 64 KiB */             \
    if (!allowed[m0->device & 63]) {                                                                                      \
      GP_HIP(hipFuncSetAttribute((const void *)halfstep_small_kernel<KD>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                 (int)(sizeof(double) * HS_NMAX * HS_LDK)));                                              \
      allowed[m0->device & 63] = true;                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL(halfstep_small_kernel<KD>, grid, block, shm, st, ha, pargs);                                      \
  } while (0)
  switch (kstar_kind(m0)) {
    case 0: GP_LAUNCH_HS(0); break;
    case 1: GP_LAUNCH_HS(1); break;
    case 2: GP_LAUNCH_HS(2); break;
    default: GP_LAUNCH_HS(3); break;
  }
#undef GP_LAUNCH_HS
  GP_HIP(hipGetLastError());
  g_halfstep_launches.fetch_add(1, std::memory_order_relaxed);
  return GPEMU_OK;
}

}  // namespace gpemu

#ifdef GPEMU_HS_STAMPS
extern "C" int gpemu_debug_hs_wall(long long *out2048) {
  return hipMemcpyFromSymbol(out2048, HIP_SYMBOL(gpemu::g_hs_wall), sizeof(long long) * 2048) == hipSuccess ? 0 : 1;
}
extern "C" int gpemu_debug_hs_stamps(long long *out96) {
  return hipMemcpyFromSymbol(out96, HIP_SYMBOL(gpemu::g_hs_stamps), sizeof(long long) * 96) == hipSuccess ? 0 : 1;
}
#endif
extern "C" int64_t gpemu_halfstep_small_launches(void) { return gpemu::g_halfstep_launches.load(std::memory_order_relaxed); }
