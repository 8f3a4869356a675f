// Sampler state shared by k_sampler.hip (stretch move, phase API, RCCL run) and k_front.hip (fused run).
#pragma once
#include <vector>

#include "internal.h"

namespace gpemu {
constexpr int RNG_RING = 32;        // steps of randomness kept on the device
constexpr int RNG_BATCH = 16;       // steps generated per launch (half the ring: the previous step's draws stay
                                    // readable while the next batch is written)
constexpr int GATHER_SLOTS = 3;
// a log-probability that has not arrived yet: a quiet NaN with a payload no computation produces
constexpr unsigned long long GATHER_EMPTY = 0x7FF8DEADBEEF0001ull;
}  // namespace gpemu

struct gpemu_sampler {
  int device = 0;
  std::vector<gpemu_model *> groups;
  int64_t W = 0, d = 0;
  int64_t ns[2] = {0, 0};      // set sizes: ceil(W/2), floor(W/2)
  int64_t qcap = 0;            // rows of q (>= ns[0] rounded up to 128, + 128)
  double a = 2.0;
  uint64_t seed = 0;
  int nchains = 1;             // independent chains stacked in this sampler (W = nchains x walkers per chain)
  unsigned long long *seeds = nullptr;   // [nchains] Philox keys, on the device
  uint64_t step_counter = 0;   // RNG counter, never reset
  int64_t iterations = 0;      // steps since the last reset
  hipStream_t stream = nullptr;
  double *X = nullptr;         // [W][DPAD]  current positions: one of the two halves of Xbuf
  double *logp = nullptr;      // [W]        current log-probabilities: one of the two halves of lpbuf
  double *Xbuf = nullptr;      // [2][W][DPAD]  the fused run writes the accepted state into the other half
  double *lpbuf = nullptr;     // [2][W]
  int cur = 0;                 // which half X / logp point at
  // per-step randomness, ring of RNG_RING steps (slot = step_counter % RNG_RING)
  int *inds = nullptr;         // [RING][W] split of each walker
  int *idx = nullptr;          // [RING][2][W] members of each set, ascending walker index
  double *zz = nullptr;        // [RING][2][W]
  double *logu = nullptr;      // [RING][2][W]
  int *rint = nullptr;         // [RING][2][W]  partner walker of each proposal: c[randint(nc)] as a walker index
  double *fac = nullptr;       // [RING][2][W]  (d - 1) log zz
  int *pos = nullptr;          // [RING][W]     position of each walker in its set's list
  uint64_t rng_ready_until = 0; // steps [.., rng_ready_until) of the device stream are in the ring
  double *q = nullptr;         // [qcap][DPAD]
  double *factors = nullptr;   // [W]
  double *newlp = nullptr;     // [qcap]
  long long *naccept = nullptr;  // [W]
  int *flags = nullptr;        // [1] count of NaN log-probabilities seen
  double *chain = nullptr;     // [chain_cap][W][d]
  double *lpchain = nullptr;   // [chain_cap][W]
  int64_t chain_cap = 0, chain_len = 0;
  // multi-GPU: this rank's slice / the gathered log-probabilities of each half ([per] / [per*world])
  double *gmine[2] = {nullptr, nullptr};
  double *gfull[2] = {nullptr, nullptr};
  int64_t gper[2] = {0, 0};
  int gworld = 0;
  // fused run (k_front.hip): proposals of the half in flight and of the one before, log-probability exchange
  double *q2 = nullptr;        // [2][qcap][DPAD]
  double *gather = nullptr;    // [GATHER_SLOTS][ns[0]]  this rank's copy of every proposal's new log-probability
  bool gather_uncached = false;
  double **peers = nullptr;    // device array [peer_world]: every rank's gather buffer (own one included)
  std::vector<void *> peer_opened;   // IPC mappings to close
  int peer_world = 0, peer_rank = 0;
  int device_share = 1;        // ranks of the job whose samplers run on this device (gpemu_sampler_peer_share)
  uint64_t front_count = 0;    // fused launches so far (gather slot and buffer parity)
  // XCD-aware order of the front kernel's cross-kernel workgroups, per (share size, group): device tables (k_front.hip)
  struct FrontPerm { int64_t cnt; int group, wg0; int *dperm; };
  std::vector<FrontPerm> front_perms;
  // snapshot of the chain state (gpemu_sampler_snapshot / _restore): a block of steps that failed -- a lost peer
  // exchange -- is rerun from here over another transport and gives the chain of an unbroken run
  double *snapX = nullptr, *snaplp = nullptr;        // [W][DPAD], [W]
  long long *snapacc = nullptr;                      // [W]
  bool snap_valid = false;
  uint64_t snap_step_counter = 0;
  int64_t snap_iterations = 0, snap_chain_len = 0;
  // autocorrelation estimate (k_acf.hip): scratch kept between the lag blocks of one estimate
  double *acf_part = nullptr, *acf_acf = nullptr, *acf_mean = nullptr, *acf_acf0 = nullptr;
  size_t acf_part_bytes = 0, acf_acf_bytes = 0, acf_mean_bytes = 0;   // capacity of acf_part / acf_acf / acf_mean + acf_acf0
  int64_t acf_first = -1, acf_n = -1, acf_w0 = -1, acf_nw = -1;       // the estimate the scratch belongs to
};

namespace gpemu {
// k_sampler.hip
int sampler_launch_rng(gpemu_sampler *s, hipStream_t st, int64_t ahead);
int sampler_ensure_chain(gpemu_sampler *s, int64_t need);
int sampler_check_nan(gpemu_sampler *s);
// k_front.hip
void front_release(gpemu_sampler *s);                         // frees the gather buffer and the peer mappings
bool front_eligible(const gpemu_sampler *s);
bool front_eligible_for(const gpemu_sampler *s, int world);   // ... and, with several ranks on this device, their launches fit on it together
// `steps` stretch-move steps with two launches per half-step (fused front kernel + triangular GEMM); world / rank:
// how the proposing half is split (world = 1: everything here); emulate: evaluate rank 0's share of a `world`-rank job
int front_run(gpemu_sampler *s, int64_t steps, int store_chain, int world, int rank, bool emulate);
}  // namespace gpemu
