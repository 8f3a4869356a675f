"""-m gpu tests of the device-resident stretch-move sampler against the CPU restatement."""
import numpy as np
import pytest

import golden_util as GU
from oracle import gp_oracle as O
from oracle import sampler_oracle as SO

pytestmark = pytest.mark.gpu


def _setup(name="g1_rbf_noise"):
    g = GU.load(name)
    model = GU.group_model(g)
    dm = GU.device_model(model)
    dm.likelihood_setup(g["y_exp"], g["y_err"], g["lo"], g["hi"], 1.0)

    def oracle_lp(X):
        X = np.atleast_2d(X)
        return np.array([O.log_posterior(x, {"g": model}, g["lo"], g["hi"], g["y_exp"], g["y_err"])[0] for x in X])
    return g, model, dm, oracle_lp


def _compare_chains(chain, lps, ochain, olps):
    """Identical accept decisions -> identical positions; a decision can only differ when
    lnpdiff - log u is within rounding of 0, which does not happen on these seeds."""
    np.testing.assert_allclose(chain, ochain, rtol=1e-12, atol=1e-12)
    fin = np.isfinite(olps)
    assert np.array_equal(fin, np.isfinite(lps))
    np.testing.assert_allclose(lps[fin], olps[fin], rtol=1e-8)


@pytest.mark.parametrize("W,steps", [(12, 12), (24, 12), (33, 12), (130, 8), (1025, 4), (8192, 2)])
def test_device_philox_chain_equals_oracle(W, steps):
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    g, model, dm, oracle_lp = _setup()
    X0 = synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"])
    ds = DeviceSampler([dm], W, a=2.0, seed=0xC0FFEE12345)
    ds.set_state(X0)
    X, lp0 = ds.get_state()
    np.testing.assert_array_equal(X, X0)
    np.testing.assert_allclose(lp0, oracle_lp(X0), rtol=1e-8)
    ds.run(steps)
    chain, lps = ds.get_chain()
    ochain, olps, onacc = SO.run(X0, oracle_lp, SO.PhiloxStream(0xC0FFEE12345), steps)
    _compare_chains(chain, lps, ochain, olps)
    nacc, iters, clen = ds.counts()
    assert iters == steps and clen == steps
    np.testing.assert_array_equal(nacc, onacc)
    # reset() clears chain and counters but keeps the state; the RNG stream continues
    ds.reset()
    assert ds.counts()[1:] == (0, 0)
    ds.close()
    dm.close()


def test_host_rng_replay_equals_emcee_stream_oracle():
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    g, model, dm, oracle_lp = _setup("g1_matern15_noise")
    W = 20
    X0 = synthetic.make_walkers(W, seed=5, lo=g["lo"], hi=g["hi"])
    ds = DeviceSampler([dm], W)
    ds.set_state(X0)
    stream = SO.EmceeStream(2024)
    for _ in range(10):
        ds.step_host_rng(*stream.draw(W))
    chain, lps = ds.get_chain()
    ochain, olps, _ = SO.run(X0, oracle_lp, SO.EmceeStream(2024), 10)
    _compare_chains(chain, lps, ochain, olps)
    ds.close()
    dm.close()


def test_multigroup_sampler_sums_groups():
    from gpemu.sampler import DeviceSampler
    g = GU.load("g5_multigroup")
    models = {grp: GU.group_model(g, prefix=grp + "_") for grp in ("g1", "g2")}
    mapping = {"A": ("g1", slice(0, 10), slice(0, 10)), "B": ("g2", slice(10, 18), slice(0, 8)),
               "C": ("g1", slice(18, 30), slice(10, 22))}
    dms = []
    for grp, cols, bs in (("g1", g["cols_g1"], [0, 10, 22]), ("g2", g["cols_g2"], [0, 8])):
        dm = GU.device_model(models[grp])
        dm.likelihood_setup(g["y_exp"][cols], g["y_err"][cols], g["lo"], g["hi"], 1.0, block_start=bs)
        dms.append(dm)
    W = 16
    ds = DeviceSampler(dms, W, seed=7)
    ds.set_state(g["Xq"])
    _, lp0 = ds.get_state()
    np.testing.assert_allclose(lp0, g["logpost_per_walker"], rtol=1e-8)   # the reference's merged value

    def oracle_lp(X):
        return np.array([O.log_posterior(x, models, g["lo"], g["hi"], g["y_exp"], g["y_err"], mapping)[0]
                         for x in np.atleast_2d(X)])
    ds.run(5)
    chain, lps = ds.get_chain()
    ochain, olps, _ = SO.run(g["Xq"], oracle_lp, SO.PhiloxStream(7), 5)
    _compare_chains(chain, lps, ochain, olps)
    ds.close()
    for dm in dms:
        dm.close()


def test_c3_sampler_bookkeeping_and_statistics():
    """BASELINE config 3 size (1024 walkers): stored log-probs equal a fresh batched evaluation of the
    stored positions (bit-identical), walkers stay in the box, acceptance is sane."""
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    model, prob, _ = GU.fixed_theta_model(1000, 500, 10, seed=0)
    dm = GU.device_model(model)
    dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    W = 1024
    ds = DeviceSampler([dm], W, seed=1)
    ds.set_state(synthetic.make_walkers(W, seed=1))
    ds.run(40)
    chain, lps = ds.get_chain()
    assert chain.shape == (40, W, 6) and lps.shape == (40, W)
    np.testing.assert_array_equal(dm.logpost(chain[-1]), lps[-1])
    np.testing.assert_array_equal(dm.logpost(chain[17]), lps[17])
    assert np.all(chain > prob["lo"]) and np.all(chain < prob["hi"])
    assert np.all(np.isfinite(lps))
    nacc, iters, _ = ds.counts()
    af = nacc / iters
    assert 0.05 < af.mean() < 0.95
    # log-probability of the ensemble increases during burn-in
    assert lps[-1].mean() > lps[0].mean()
    ds.close()
    dm.close()


# ---- sharded device sampler: 2 ranks on the one GPU, log-probabilities exchanged over gloo ----------
def _sharded_worker(rank, world, port, out_dir, transport, W=26):
    import os
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    g = GU.load("g1_rbf_noise")
    dm = GU.device_model(GU.group_model(g))
    dm.likelihood_setup(g["y_exp"], g["y_err"], g["lo"], g["hi"], 1.0)
    # W = 26: halves of 13 (ragged shards);  W = 66 at 4 / 5 ranks: shares of 9, 9, 9, 6 and 7, 7, 7, 7, 5
    ds = DeviceSampler([dm], W, seed=99)
    ds.set_state(synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"]))
    # many short back-to-back runs (9 steps in all): the exchange slots and their hand-back carry over from run to
    # run, and a rank may re-enter while its peer is still finishing the run before (the ranks pass a
    # barrier before every fused run)
    for n_steps in (1, 1, 2, 1, 3, 1):
        ds.run_sharded(n_steps, transport=transport)
    if transport == "peer":
        assert ds._peer_ok and all(ds._peer_ok.values()), "the peer transport was not taken"
    chain, lps = ds.get_chain()
    np.save(os.path.join(out_dir, f"chain_{rank}.npy"), chain)
    np.save(os.path.join(out_dir, f"lp_{rank}.npy"), lps)
    nacc, it, _ = ds.counts()
    np.save(os.path.join(out_dir, f"nacc_{rank}.npy"), nacc)
    dist.barrier()
    dist.destroy_process_group()
    ds.close()
    dm.close()


@pytest.mark.parametrize("transport", ["peer", "torch"])
def test_sharded_device_sampler_two_ranks_equals_single(tmp_path, transport):
    """Two ranks (two processes on the one GPU, rendezvous over gloo) shard the proposals; "peer": the fused run,
    each rank storing its log-probabilities into the other's gather buffer through an IPC mapping; "torch": the
    per-phase API with the values staged through the host."""
    import os
    import torch.multiprocessing as mp
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    port = 29600 + (os.getpid() % 2000) + (7 if transport == "peer" else 0)
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path), transport), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "chain_0.npy"), np.load(tmp_path / "chain_1.npy")
    np.testing.assert_array_equal(c0, c1)                       # every rank holds the same ensemble
    np.testing.assert_array_equal(np.load(tmp_path / "lp_0.npy"), np.load(tmp_path / "lp_1.npy"))
    g, model, dm, _ = _setup()
    W = 26
    ds = DeviceSampler([dm], W, seed=99)
    ds.set_state(synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"]))
    ds.run(9)
    chain, lps = ds.get_chain()
    np.testing.assert_array_equal(chain, c0)                    # and it is the single-GPU chain, bit for bit
    np.testing.assert_array_equal(lps, np.load(tmp_path / "lp_0.npy"))
    np.testing.assert_array_equal(ds.counts()[0], np.load(tmp_path / "nacc_0.npy"))
    ds.close()
    dm.close()


@pytest.mark.parametrize("world", [4, 5])
@pytest.mark.parametrize("transport", ["peer", "torch"])
def test_sharded_device_sampler_many_ranks_equals_single(tmp_path, transport, world):
    """More than two ranks (every sharded test had world = 2): 4 and 5 processes on the one GPU -- the pool
    allows at most 6 processes on a card including this one, so the 8-way arithmetic is covered by the slice test
    below -- with ragged shares (66 walkers: halves of 33 -> 9, 9, 9, 6 and 7, 7, 7, 7, 5), fused peer-store run and
    per-phase run: every rank ends with the single-GPU chain, bit for bit."""
    import os
    import torch.multiprocessing as mp
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    W = 66
    port = 29800 + (os.getpid() % 1500) + 11 * world + (5 if transport == "peer" else 0)
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path), transport, W), nprocs=world, join=True)
    chains = [np.load(tmp_path / f"chain_{r}.npy") for r in range(world)]
    lps = [np.load(tmp_path / f"lp_{r}.npy") for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(chains[0], chains[r])
        np.testing.assert_array_equal(lps[0], lps[r])
    g, model, dm, _ = _setup()
    ds = DeviceSampler([dm], W, seed=99)
    ds.set_state(synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"]))
    ds.run(9)
    chain, lp1 = ds.get_chain()
    np.testing.assert_array_equal(chain, chains[0])
    np.testing.assert_array_equal(lp1, lps[0])
    np.testing.assert_array_equal(ds.counts()[0], np.load(tmp_path / "nacc_0.npy"))
    ds.close()
    dm.close()


@pytest.mark.parametrize("world", [8, 5])
def test_phase_api_slices_of_eight_ranks_equal_single(world):
    """The 8-way share arithmetic (shard_bounds / the C side's share_of: ceil(n / world) per rank, the last ranks short
    or EMPTY) through the per-phase C ABI: one process plays every rank of an 8-rank (and a 5-rank) job in turn --
    gpemu_sampler_half_propose_eval on each rank's slice, the "all-gather" a concatenation -- and must reproduce the
    undivided chain bit for bit.  70 walkers: halves of 35 -> 8 ranks: 5, 5, 5, 5, 5, 5, 5, 0."""
    import ctypes as C
    import torch
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler, shard_bounds
    L = _lib.lib()
    g, model, dm, _ = _setup()
    W, steps = 70, 6
    X0 = synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"])
    ref = DeviceSampler([dm], W, seed=31)
    ref.set_state(X0)
    ref.run(steps)
    cref, lref = ref.get_chain()
    a = DeviceSampler([dm], W, seed=31)
    a.set_state(X0)
    dev = torch.device("cuda", a.device)
    _lib.check(L.gpemu_sampler_reserve_chain(a._h, steps))
    sizes = []
    for _ in range(steps):
        _lib.check(L.gpemu_sampler_begin_step(a._h))
        for h in (0, 1):
            n = a.ns[h]
            per = shard_bounds(n, world, 0)[2]
            full = torch.zeros(per * world, dtype=torch.float64, device=dev)
            for r in range(world):
                lo, hi, _ = shard_bounds(n, world, r)
                sizes.append(hi - lo)
                mine = torch.zeros(per, dtype=torch.float64, device=dev)
                torch.cuda.synchronize()      # (the fill runs on torch's stream, the evaluation on the sampler's)
                _lib.check(L.gpemu_sampler_half_propose_eval(a._h, h, lo, hi, C.c_void_p(mine.data_ptr())))
                torch.cuda.synchronize()
                full[r * per:(r + 1) * per] = mine
            torch.cuda.synchronize()          # (the copies into `full` ran on torch's stream, the accept runs on the sampler's)
            _lib.check(L.gpemu_sampler_half_accept(a._h, h, C.c_void_p(full.data_ptr()), 1))
        _lib.check(L.gpemu_sampler_end_step(a._h, 1))
    assert L.gpemu_sampler_check(a._h) == 0
    assert min(sizes) == 0 if world == 8 else min(sizes) > 0          # the 8-rank job has an empty share
    ca, la = a.get_chain()
    # shares and the undivided half (35 proposals) are all served by the small-batch kernel family: same bits
    np.testing.assert_array_equal(la, lref)
    np.testing.assert_array_equal(ca, cref)
    a.close(); ref.close(); dm.close()


def test_groups_in_one_launch_per_stage_equal_the_per_group_launches(monkeypatch):
    """A sampler over several emulation groups at the shipped size (~150 design points, 200 walkers, 5 / 11 / 25 PCs)
    runs ONE cross-kernel, ONE triangular-GEMM and ONE likelihood launch per half-step for all groups
    (gpemu_api.hip: logpost_groups); with GPEMU_NO_GROUP_MERGE the nine per-group launches: the same chain, bit for bit
    -- also for two groups (two walkers per likelihood workgroup), through the per-phase API, and against the oracle."""
    import ctypes as C
    import torch
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler, shard_bounds
    L = _lib.lib()
    models, dms, probs = [], [], []
    for gi, (F, k) in enumerate([(40, 5), (60, 11), (90, 25)]):
        model, prob, _ = GU.fixed_theta_model(150, F, k, seed=gi)
        dm = GU.device_model(model)
        dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
        models.append(model); dms.append(dm); probs.append(prob)
    lo, hi = probs[0]["lo"], probs[0]["hi"]
    for ng, W in ((3, 200), (2, 37)):
        X0 = synthetic.make_walkers(W, seed=21, lo=lo, hi=hi)
        out = {}
        for merged in (True, False):
            if merged:
                monkeypatch.delenv("GPEMU_NO_GROUP_MERGE", raising=False)
            else:
                monkeypatch.setenv("GPEMU_NO_GROUP_MERGE", "1")
            ds = DeviceSampler(dms[:ng], W, seed=8)
            ds.set_state(X0)
            ds.run(6)
            out[merged] = ds.get_chain() + (ds.counts()[0],)
            ds.close()
        monkeypatch.delenv("GPEMU_NO_GROUP_MERGE", raising=False)
        for a, b in zip(out[True], out[False]):
            np.testing.assert_array_equal(a, b)
        # the oracle's log-posterior of the stored positions (sum over the groups)
        chain, lps = out[True][0], out[True][1]
        for w in (0, W // 2, W - 1):
            ref = sum(O.log_posterior(chain[-1, w], {"g": models[g]}, lo, hi, probs[g]["y_exp"], probs[g]["y_err"])[0]
                      for g in range(ng))
            np.testing.assert_allclose(lps[-1, w], ref, rtol=1e-9)
        # per-phase API, the half cut into three slices (each a merged launch chain of its own)
        a = DeviceSampler(dms[:ng], W, seed=8)
        a.set_state(X0)
        dev = torch.device("cuda", a.device)
        _lib.check(L.gpemu_sampler_reserve_chain(a._h, 6))
        for _ in range(6):
            _lib.check(L.gpemu_sampler_begin_step(a._h))
            for h in (0, 1):
                n = a.ns[h]
                per = shard_bounds(n, 3, 0)[2]
                full = torch.zeros(per * 3, dtype=torch.float64, device=dev)
                for r in range(3):
                    l0, h0, _ = shard_bounds(n, 3, r)
                    mine = torch.zeros(per, dtype=torch.float64, device=dev)
                    torch.cuda.synchronize()      # (the fill runs on torch's stream, the evaluation on the sampler's)
                    _lib.check(L.gpemu_sampler_half_propose_eval(a._h, h, l0, h0, C.c_void_p(mine.data_ptr())))
                    torch.cuda.synchronize()
                    full[r * per:(r + 1) * per] = mine
                torch.cuda.synchronize()          # (the copies into `full` ran on torch's stream, the accept runs on the sampler's)
                _lib.check(L.gpemu_sampler_half_accept(a._h, h, C.c_void_p(full.data_ptr()), 1))
            _lib.check(L.gpemu_sampler_end_step(a._h, 1))
        ca, la = a.get_chain()
        np.testing.assert_array_equal(ca, chain)
        np.testing.assert_array_equal(la, lps)
        a.close()
    for dm in dms:
        dm.close()


@pytest.mark.parametrize("N,pcs,W,kind,nu", [
    (150, (5, 11, 25), 200, "rbf", np.inf),      # the shipped shape (ref: config/jet_substructure.yaml:243-271)
    (150, (5, 11, 25), 100, "matern", 1.5),      # ... with the shipped kernel and walker count
    (33, (3,), 10, "rbf", np.inf),               # two 32-row blocks, the second with one real row; five proposals per half
    (97, (16, 17), 37, "matern", 2.5),           # both orders of the likelihood's partial sums (k <= 16, k > 16); ragged halves
    (256, (32,), 256, "matern", 0.5),            # the largest shape the kernel takes: 256 rows, 32 PCs, 128 proposals
    (128, (10,), 64, "rbf", np.inf),
    (12, (3,), 8, "matern", 1.5),                # fewer design points than one 16-row tile
])
def test_small_emulators_cross_kernel_and_gemm_in_one_launch_have_the_bits_of_the_general_path(N, pcs, W, kind, nu, monkeypatch):
    """Emulators of at most 256 design points (the reference's shipped analysis has ~150): cross-kernel and triangular
    GEMM of every group in ONE launch per half-step, K_*^T never leaving the workgroup (csrc/k_halfstep.hip), then the
    likelihood launch.  Every partial sum is formed and added in the order of the three-launch path: with
    GPEMU_NO_HALFSTEP the same chain, bit for bit -- positions, log-probabilities, acceptance counts -- and the stored
    log-probabilities are the oracle's."""
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler
    L = _lib.lib()
    K = O.RBF if kind == "rbf" else O.MATERN
    models, dms, probs = [], [], []
    for gi, k in enumerate(pcs):
        model, prob, _ = GU.fixed_theta_model(N, 40 + 10 * gi, k, seed=gi, kind=K, nu=nu)
        dm = GU.device_model(model)
        dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
        models.append(model); dms.append(dm); probs.append(prob)
    lo, hi = probs[0]["lo"], probs[0]["hi"]
    X0 = synthetic.make_walkers(W, seed=5, lo=lo, hi=hi)
    monkeypatch.setenv("GPEMU_HALFSTEP_MIN_PAIRS", "0")      # (the kernel also where the general path is the faster one)
    out, launches = {}, {}
    for form in ("small", "general"):
        monkeypatch.delenv("GPEMU_NO_HALFSTEP", raising=False)
        if form == "general":
            monkeypatch.setenv("GPEMU_NO_HALFSTEP", "1")
        n0 = L.gpemu_halfstep_small_launches()
        ds = DeviceSampler(dms, W, seed=12)
        ds.set_state(X0)
        ds.run(7)
        ds.run(3)
        out[form] = ds.get_chain() + (ds.counts()[0], ds.get_state()[0], ds.get_state()[1])
        launches[form] = L.gpemu_halfstep_small_launches() - n0
        ds.close()
    monkeypatch.delenv("GPEMU_NO_HALFSTEP", raising=False)
    # which path ran (20 half-steps; set_state's evaluation of the start positions takes it too)
    assert launches["small"] >= 20 and launches["general"] == 0, launches
    for a, b in zip(out["small"], out["general"]):
        np.testing.assert_array_equal(a, b)
    assert out["general"][2].sum() > 0                                     # (moves were accepted)
    chain, lps = out["small"][0], out["small"][1]
    for w in (0, W - 1):
        ref = sum(O.log_posterior(chain[-1, w], {"g": models[g]}, lo, hi, probs[g]["y_exp"], probs[g]["y_err"])[0]
                  for g in range(len(pcs)))
        np.testing.assert_allclose(lps[-1, w], ref, rtol=1e-9)
    for dm in dms:
        dm.close()


@pytest.mark.parametrize("name", ["g1_matern25_const_noise", "g1_rbf_only", "g2_rbf_noise"])
def test_small_emulator_launch_on_the_reference_fits(name, monkeypatch):
    """The same comparison on models the REFERENCE fitted (goldens G1 / G2: Matern-2.5 + Const + White with the constant
    kernel's offset in every cross-kernel value, RBF without noise, N = 50 and 200): chain of the one-launch cross-kernel +
    GEMM equal to the general path's bit for bit, log-probabilities the oracle's."""
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler
    L = _lib.lib()
    g, model, dm, oracle_lp = _setup(name)
    monkeypatch.setenv("GPEMU_HALFSTEP_MIN_PAIRS", "0")
    W = 46
    X0 = synthetic.make_walkers(W, seed=9, lo=g["lo"], hi=g["hi"])
    out = {}
    for form in ("small", "general"):
        monkeypatch.delenv("GPEMU_NO_HALFSTEP", raising=False)
        if form == "general":
            monkeypatch.setenv("GPEMU_NO_HALFSTEP", "1")
        n0 = L.gpemu_halfstep_small_launches()
        ds = DeviceSampler([dm], W, seed=77)
        ds.set_state(X0)
        ds.run(8)
        out[form] = ds.get_chain() + (ds.counts()[0],)
        assert (L.gpemu_halfstep_small_launches() - n0 >= 16) == (form == "small")
        ds.close()
    monkeypatch.delenv("GPEMU_NO_HALFSTEP", raising=False)
    for a, b in zip(out["small"], out["general"]):
        np.testing.assert_array_equal(a, b)
    chain, lps = out["small"][0], out["small"][1]
    ref = oracle_lp(chain[-1, :6])
    fin = np.isfinite(ref)
    np.testing.assert_allclose(lps[-1, :6][fin], ref[fin], rtol=1e-8)
    dm.close()


# ---- BASELINE configs[3] ("C4") at its full size: the C3 model, 1024 walkers, shares of 128 / 64 proposals ----------
def _c4_reference(dm, prob, steps):
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    W = 1024
    ref = DeviceSampler([dm], W, seed=5)
    ref.set_state(synthetic.make_walkers(W, seed=1, lo=prob["lo"], hi=prob["hi"]))
    ref.run(steps)
    c, l = ref.get_chain()
    nacc = ref.counts()[0]
    ref.close()
    return c, l, nacc


def _c4_compare(chain, lps, cref, lref):
    """A rank's share (64 or 128 columns) goes through the small-batch GEMM, the undivided half (512 columns) through
    the large-batch one: different summation order of the 64-row partial sums, log-probabilities equal to ~1e-13
    relative.  The positions depend on them only through the accept decisions: identical."""
    np.testing.assert_array_equal(chain, cref)
    np.testing.assert_allclose(lps, lref, rtol=1e-11, atol=0)


def test_c4_full_size_eight_slices_equal_single_gpu_chain():
    """1024 walkers of the C3 model cut into the 8 shares of configs[3] (64 proposals per rank and half-step) through the
    per-phase C ABI, one process playing the eight ranks in turn: the chain of the undivided run."""
    import ctypes as C
    import torch
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler, shard_bounds
    L = _lib.lib()
    model, prob, _ = GU.fixed_theta_model(1000, 500, 10, seed=0)
    dm = GU.device_model(model)
    dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    W, steps, world = 1024, 4, 8
    cref, lref, _ = _c4_reference(dm, prob, steps)
    a = DeviceSampler([dm], W, seed=5)
    a.set_state(synthetic.make_walkers(W, seed=1, lo=prob["lo"], hi=prob["hi"]))
    dev = torch.device("cuda", a.device)
    _lib.check(L.gpemu_sampler_reserve_chain(a._h, steps))
    for _ in range(steps):
        _lib.check(L.gpemu_sampler_begin_step(a._h))
        for h in (0, 1):
            n = a.ns[h]
            per = shard_bounds(n, world, 0)[2]
            assert per == 64
            full = torch.zeros(per * world, dtype=torch.float64, device=dev)
            for r in range(world):
                lo, hi, _ = shard_bounds(n, world, r)
                mine = torch.zeros(per, dtype=torch.float64, device=dev)
                torch.cuda.synchronize()      # (the fill runs on torch's stream, the evaluation on the sampler's)
                _lib.check(L.gpemu_sampler_half_propose_eval(a._h, h, lo, hi, C.c_void_p(mine.data_ptr())))
                torch.cuda.synchronize()
                full[r * per:(r + 1) * per] = mine
            torch.cuda.synchronize()          # (the copies into `full` ran on torch's stream, the accept runs on the sampler's)
            _lib.check(L.gpemu_sampler_half_accept(a._h, h, C.c_void_p(full.data_ptr()), 1))
        _lib.check(L.gpemu_sampler_end_step(a._h, 1))
    assert L.gpemu_sampler_check(a._h) == 0
    ca, la = a.get_chain()
    _c4_compare(ca, la, cref, lref)
    a.close(); dm.close()


def _c4_worker(rank, world, port, out_dir, transport):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    model, prob, _ = GU.fixed_theta_model(1000, 500, 10, seed=0)
    dm = GU.device_model(model)
    dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    W = 1024
    ds = DeviceSampler([dm], W, seed=5)
    ds.set_state(synthetic.make_walkers(W, seed=1, lo=prob["lo"], hi=prob["hi"]))
    for n_steps in (1, 3):
        ds.run_sharded(n_steps, transport=transport)
    chain, lps = ds.get_chain()
    np.save(os.path.join(out_dir, f"chain_{rank}.npy"), chain)
    np.save(os.path.join(out_dir, f"lp_{rank}.npy"), lps)
    np.save(os.path.join(out_dir, f"nacc_{rank}.npy"), ds.counts()[0])
    with open(os.path.join(out_dir, f"transport_{rank}.txt"), "w") as fh:
        fh.write(ds.last_transport)
    dist.barrier()
    dist.destroy_process_group()
    ds.close()
    dm.close()


@pytest.mark.parametrize("transport", ["torch", "peer"])
def test_c4_full_size_four_ranks_equal_single_gpu_chain(tmp_path, transport):
    """configs[3] as processes: 4 ranks (the pool's limit of GPU processes per card is 6) x 128 proposals of the
    1024-walker C3 job on the one GPU.  "torch": per-phase C ABI + all-gather.  "peer": four full-size front launches do
    not fit one device together, so the ranks must agree on the fall-back (no time-out) and still end with the
    same chain -- on a node with one rank per GPU the fused run is taken instead."""
    import os
    import torch.multiprocessing as mp
    world = 4
    port = 30200 + (os.getpid() % 1500) + (3 if transport == "peer" else 0)
    mp.spawn(_c4_worker, args=(world, port, str(tmp_path), transport), nprocs=world, join=True)
    chains = [np.load(tmp_path / f"chain_{r}.npy") for r in range(world)]
    lps = [np.load(tmp_path / f"lp_{r}.npy") for r in range(world)]
    taken = [(tmp_path / f"transport_{r}.txt").read_text() for r in range(world)]
    assert len(set(taken)) == 1, taken                         # every rank ran the same transport
    for r in range(1, world):
        np.testing.assert_array_equal(chains[0], chains[r])    # and holds the same ensemble, bit for bit
        np.testing.assert_array_equal(lps[0], lps[r])
    model, prob, _ = GU.fixed_theta_model(1000, 500, 10, seed=0)
    dm = GU.device_model(model)
    dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    cref, lref, nacc = _c4_reference(dm, prob, 4)
    _c4_compare(chains[0], lps[0], cref, lref)
    np.testing.assert_array_equal(nacc, np.load(tmp_path / "nacc_0.npy"))
    dm.close()


def _rccl_worker(rank, port, out_dir):
    import os
    import torch
    import torch.distributed as dist
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    g, model, dm, _ = _setup()
    W = 24
    X0 = synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"])
    b = DeviceSampler([dm], W, seed=5)
    b.set_state(X0)
    b.run(6)
    cb, lb = b.get_chain()
    ok = []
    for transport in ("rccl", "torch"):     # library-owned communicator / torch.distributed's
        a = DeviceSampler([dm], W, seed=5)
        a.set_state(X0)
        a.run_sharded(4, force=True, transport=transport)
        a.run_sharded(2, force=True, transport=transport)
        ca, la = a.get_chain()
        ok += [np.array_equal(ca, cb), np.array_equal(la, lb)]
        if transport == "rccl":             # the communicator's stand-alone all-gather entry point
            import ctypes as C
            from gpemu import _lib
            src = torch.arange(5, dtype=torch.float64, device="cuda")
            dst = torch.zeros(5, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            comm = a._rccl_comm_agreed(None)
            _lib.check(_lib.lib().gpemu_comm_all_gather(comm, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()),
                                                        5, None))
            torch.cuda.synchronize()
            ok.append(bool(torch.equal(src, dst)))
        a.close()
    np.save(os.path.join(out_dir, "ok.npy"), np.array(ok))
    b.close(); dm.close()
    dist.destroy_process_group()


def test_sharded_path_over_rccl_world1(tmp_path):
    """The RCCL (backend "nccl") plumbing of the sharded run -- the library's own communicator
    (gpemu_comm_* / gpemu_sampler_run_sharded) and the torch.distributed transport -- exercised with a
    single-rank group (one GPU is all this box has)."""
    import os
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(29700 + (os.getpid() % 2000), str(tmp_path)), nprocs=1, join=True)
    assert np.load(tmp_path / "ok.npy").all()


def test_fused_run_world1_equals_three_launch_run():
    """gpemu_sampler_run_peer with a one-rank "world" (the two-launch half-step: likelihood + exchange + accept +
    proposal + cross-kernel in one kernel) against gpemu_sampler_run (three launches per half-step): same chain, bit
    for bit, over several calls and an odd ensemble size."""
    import ctypes as C
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler
    g, model, dm, _ = _setup()
    L = _lib.lib()
    for W in (24, 33, 130, 1025):
        X0 = synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"])
        a = DeviceSampler([dm], W, seed=7)
        a.set_state(X0)
        a.run(11)
        b = DeviceSampler([dm], W, seed=7)
        b.set_state(X0)
        h = (C.c_char * 64)()
        _lib.check(L.gpemu_sampler_peer_export(b._h, C.cast(h, C.c_void_p)))
        _lib.check(L.gpemu_sampler_peer_import(b._h, 1, 0, C.cast(h, C.c_void_p)))
        for n in (1, 4, 6):                       # 11 steps in three calls, crossing no / one RNG batch edge
            _lib.check(L.gpemu_sampler_run_peer(b._h, n, 1))
        ca, la = a.get_chain()
        cb, lb = b.get_chain()
        np.testing.assert_array_equal(ca, cb)
        np.testing.assert_array_equal(la, lb)
        np.testing.assert_array_equal(a.counts()[0], b.counts()[0])
        np.testing.assert_array_equal(a.get_state()[0], b.get_state()[0])
        a.close(); b.close()
    # 40 steps in one call (randomness generated in batches of 16) against one step per call
    W = 64
    X0 = synthetic.make_walkers(W, seed=5, lo=g["lo"], hi=g["hi"])
    a = DeviceSampler([dm], W, seed=11); a.set_state(X0); a.run(40)
    b = DeviceSampler([dm], W, seed=11); b.set_state(X0)
    for _ in range(40):
        b.run(1)
    np.testing.assert_array_equal(a.get_chain()[0], b.get_chain()[0])
    a.close(); b.close()
    dm.close()


def test_fused_run_world1_multigroup_and_lds_likelihood():
    """The fused two-launch half-step over several emulation groups, one of them with more than 16 PCs (its likelihood
    factorises in LDS): golden G7, the reference's shipped shape (k = 5 / 11 / 25).  Same chain as the three-launch run,
    bit for bit, also across calls and for an odd ensemble."""
    import ctypes as C
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler
    g = GU.load("g7_shipped_config")
    names, mapping, block_start, cols = GU.g7_groups(g)
    models = GU.g7_models(g)
    dms = []
    for n in names:
        dm = GU.device_model(models[n])
        dm.likelihood_setup(g["y_exp"][cols[n]], g["y_err"][cols[n]], g["lo"], g["hi"], 1.0, block_start=block_start[n])
        dms.append(dm)
    L = _lib.lib()
    for W in (24, 201):
        X0 = synthetic.make_walkers(W, seed=4, lo=g["design"].min(0), hi=g["design"].max(0))
        a = DeviceSampler(dms, W, seed=9)
        a.set_state(X0)
        a.run(7)
        b = DeviceSampler(dms, W, seed=9)
        b.set_state(X0)
        h = (C.c_char * 64)()
        _lib.check(L.gpemu_sampler_peer_export(b._h, C.cast(h, C.c_void_p)))
        _lib.check(L.gpemu_sampler_peer_import(b._h, 1, 0, C.cast(h, C.c_void_p)))
        for n in (3, 4):
            _lib.check(L.gpemu_sampler_run_peer(b._h, n, 1))
        np.testing.assert_array_equal(a.get_chain()[0], b.get_chain()[0])
        np.testing.assert_array_equal(a.get_chain()[1], b.get_chain()[1])
        np.testing.assert_array_equal(a.counts()[0], b.counts()[0])
        a.close(); b.close()
    for dm in dms:
        dm.close()


def test_device_sampler_posterior_moments_vs_host_stretch_move():
    """Statistical check of the device sampler beyond step equality: the posterior of the G1 emulator sampled (a) by
    the device sampler (Philox randomness, fused kernels) and (b) by the host stretch move -- an independent
    implementation on numpy's RandomState in emcee's draw order -- fed the device log-posterior as a black box.
    Means agree to a fraction of the posterior width, variances to 25 %, acceptance fractions alike."""
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler, HostEnsemble
    g, model, dm, _ = _setup()
    W, burn, steps = 64, 600, 5000
    X0 = synthetic.make_walkers(W, seed=21, lo=g["lo"], hi=g["hi"])
    ds = DeviceSampler([dm], W, seed=2024)
    ds.set_state(X0)
    ds.run(burn, store=False)
    ds.run(steps)
    cd, _ = ds.get_chain()
    nacc, it, _ = ds.counts()
    he = HostEnsemble(W, dm.d, lambda q: dm.logpost(np.ascontiguousarray(q)), seed=77)
    he.set_state(X0)
    he.run(burn, store=False)
    he.run(steps)
    ch = np.stack(he.chain)
    fd, fh = cd.reshape(-1, dm.d), ch.reshape(-1, dm.d)
    sd = fh.std(axis=0)
    assert np.all(np.abs(fd.mean(axis=0) - fh.mean(axis=0)) < 0.1 * sd), (fd.mean(0), fh.mean(0), sd)
    assert np.all(np.abs(fd.std(axis=0) / sd - 1.0) < 0.25)
    # two-sample comparison of the marginal quantiles (10 %, 50 %, 90 %)
    qd, qh = np.quantile(fd, [0.1, 0.5, 0.9], axis=0), np.quantile(fh, [0.1, 0.5, 0.9], axis=0)
    assert np.all(np.abs(qd - qh) < 0.15 * sd)
    af_d = (nacc / it).mean()
    af_h = (he.naccepted / he.iterations).mean()
    assert abs(af_d - af_h) < 0.05 and 0.1 < af_d < 0.9
    ds.close()
    dm.close()


@pytest.mark.parametrize("W", [24, 33, 300])
def test_stacked_chains_equal_separate_chains(W):
    """gpemu_sampler_create_chains: C independent chains stacked in one sampler -- every chain on its own data vector
    (closure pseudo-data) and its own seed -- are, bit for bit, the chains C separate samplers produce (per-chain kernel
    variants, chunked halves, chain-aware likelihood constants)."""
    from gpemu import synthetic
    from gpemu.sampler import DeviceSampler
    g = GU.load("g1_rbf_noise")
    model = GU.group_model(g)
    C, steps = 5, 7
    rng = np.random.default_rng(17)
    ys = g["y_exp"][None, :] + 0.05 * rng.normal(size=(C, g["y_exp"].size))
    seeds = [101 + 13 * c for c in range(C)]
    X0 = np.concatenate([synthetic.make_walkers(W, seed=40 + c, lo=g["lo"], hi=g["hi"]) for c in range(C)])
    dm = GU.device_model(model)
    dm.likelihood_setup(ys, g["y_err"], g["lo"], g["hi"], 1.0)
    ms = DeviceSampler([dm], W, seeds=seeds)
    assert ms.n_chains == C and ms.W == C * W
    ms.set_state(X0)
    lp0 = ms.get_state()[1]
    ms.run(3)
    ms.run(steps - 3)
    chain, lps = ms.get_chain()
    nacc = ms.counts()[0]
    ms.close()
    for c in range(C):
        dm.likelihood_setup(ys[c], g["y_err"], g["lo"], g["hi"], 1.0)
        one = DeviceSampler([dm], W, seed=seeds[c])
        one.set_state(X0[c * W:(c + 1) * W])
        np.testing.assert_array_equal(one.get_state()[1], lp0[c * W:(c + 1) * W])
        one.run(steps)
        c1, l1 = one.get_chain()
        np.testing.assert_array_equal(chain[:, c * W:(c + 1) * W], c1)
        np.testing.assert_array_equal(lps[:, c * W:(c + 1) * W], l1)
        np.testing.assert_array_equal(nacc[c * W:(c + 1) * W], one.counts()[0])
        one.close()
    dm.close()


def test_bench_two_rank_rehearsal(tmp_path):
    """Plain `python bench.py --gpus 2` with NO launcher environment (the form the driver used for --gpus 1): bench.py
    starts its two ranks itself as child processes.  Rehearsed with both ranks on the one GPU of the box over gloo
    (RCCL refuses two ranks on a device): rank-agreed warm-up passes, the collective run and the peer-store run timed in
    one invocation, max-over-ranks timing, one JSON line from rank 0 that says which transport ran and how many ranks
    it saw."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # 64 walkers: two processes time-share the one GPU, and full-size kernels of one rank would spin on stores of a rank
    # that is not scheduled; small grids of both ranks are resident together (as in the other 2-rank tests)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPEMU_DIST_BACKEND="gloo", GPEMU_BENCH_REHEARSAL_WALKERS="64")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5",
           "--no-cpu-baseline", "--no-fit"]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 30 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["rehearsal"] is True and out["config"]["n_walkers"] == 64
    assert out["config"]["workload"].startswith("REHEARSAL") and "64-walker" in out["config"]["workload"]
    assert 0.1 < out["acceptance_fraction_mean"] < 0.9
    # both transports were timed; over gloo the collective one is torch.distributed's all-gather
    tr = out["transports"]
    assert set(tr) == {"rccl", "peer"} and all("error" not in r for r in tr.values()), tr
    assert tr["peer"]["transport_taken"] == "peer" and tr["rccl"]["transport_taken"] == "torch"
    assert out["transport"] in ("peer", "torch")
    assert out["value"] == max(r["value"] for r in tr.values())
    assert out["ranks_seen"]["torch_distributed"] == 2 and out["ranks_seen"]["peer_selftest"] == 2
    assert out["peer_selftest_per_rank"] == [1, 1]
    assert out["fallback_vote"]["peer_to_collective"] is False


def _absent_peer(out_dir):
    """A rank that maps its exchange buffer and then never runs."""
    import ctypes as C
    import os
    import time
    from gpemu import _lib
    from gpemu.sampler import DeviceSampler
    g, model, dm, _ = _setup()
    ds = DeviceSampler([dm], 24, seed=7)
    h = (C.c_char * 64)()
    _lib.check(_lib.lib().gpemu_sampler_peer_export(ds._h, C.cast(h, C.c_void_p)))
    with open(os.path.join(out_dir, "handle.tmp"), "wb") as f:
        f.write(h.raw)
    os.rename(os.path.join(out_dir, "handle.tmp"), os.path.join(out_dir, "handle.bin"))
    t0 = time.time()
    while not os.path.exists(os.path.join(out_dir, "done")) and time.time() - t0 < 120:
        time.sleep(0.05)
    ds.close(); dm.close()


def test_lost_peer_exchange_ends_in_bounded_time(tmp_path, monkeypatch):
    """The bound on the in-kernel wait for a peer's log-probabilities, with a deliberately absent peer: rank 0 of a
    two-rank "world" whose rank 1 exported its buffer and never runs.  The run must come back with GPEMU_ERR_STATE
    (not a model NaN, not a hang) within the configured time-out, the lost-exchange flag must not outlive the run
    (it used to make every later wait give up), and the sampler must still work afterwards."""
    import ctypes as C
    import os
    import time
    import torch.multiprocessing as mp
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setenv("GPEMU_PEER_TIMEOUT_MS", "300")
    ctx = mp.get_context("spawn")
    child = ctx.Process(target=_absent_peer, args=(str(tmp_path),))
    child.start()
    try:
        t0 = time.time()
        while not (tmp_path / "handle.bin").exists():
            assert time.time() - t0 < 110 and child.is_alive(), "the absent peer never exported its buffer"
            time.sleep(0.05)
        theirs = (tmp_path / "handle.bin").read_bytes()
        g, model, dm, _ = _setup()
        L = _lib.lib()
        W = 24
        X0 = synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"])
        ds = DeviceSampler([dm], W, seed=7)
        ds.set_state(X0)
        mine = (C.c_char * 64)()
        _lib.check(L.gpemu_sampler_peer_export(ds._h, C.cast(mine, C.c_void_p)))
        both = C.create_string_buffer(mine.raw + theirs, 128)
        _lib.check(L.gpemu_sampler_peer_import(ds._h, 2, 0, C.cast(both, C.c_void_p)))
        for attempt in range(2):                      # the second run waits the full bound again: the flag was cleared
            t0 = time.time()
            rc = L.gpemu_sampler_run_peer(ds._h, 3, 1)
            dt = time.time() - t0
            assert rc == -4, (rc, _lib.last_error())          # GPEMU_ERR_STATE
            assert "timed out" in _lib.last_error()
            assert 0.25 < dt < 10.0, dt
            ds.set_state(X0)
        # alone again (world 1), the same sampler produces the single-GPU chain
        ds.reset()
        _lib.check(L.gpemu_sampler_peer_import(ds._h, 1, 0, C.cast(mine, C.c_void_p)))
        _lib.check(L.gpemu_sampler_run_peer(ds._h, 5, 1))
        ref = DeviceSampler([dm], W, seed=7)
        ref.set_state(X0)
        # the failed runs advanced the random stream by 6 steps
        ref.run(6, store=False)
        ref.set_state(X0)
        ref.run(5)
        np.testing.assert_array_equal(ds.get_chain()[0][-5:], ref.get_chain()[0])
        ds.close(); ref.close(); dm.close()
    finally:
        (tmp_path / "done").write_text("x")
        child.join(60)
        if child.is_alive():
            child.kill()


def test_lost_peer_exchange_block_is_rerun_from_the_snapshot(tmp_path, monkeypatch):
    """a lost exchange no longer ends the run.  Rank 0 of a two-rank "world" whose rank 1 exported its
    buffer and never runs: the fused block times out, the ranks' vote (here: this rank's own outcome) says so, the
    chain state comes back from the device snapshot (gpemu_sampler_snapshot / _restore) and the block is rerun over
    another transport (here: the single-GPU run).  The random stream is counter based, so the chain -- positions,
    log-probabilities, acceptance counts -- is that of an unbroken run, also for the block after."""
    import ctypes as C
    import time
    import torch.multiprocessing as mp
    from gpemu import _lib, synthetic
    from gpemu.sampler import DeviceSampler
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setenv("GPEMU_PEER_TIMEOUT_MS", "300")
    ctx = mp.get_context("spawn")
    child = ctx.Process(target=_absent_peer, args=(str(tmp_path),))
    child.start()
    try:
        t0 = time.time()
        while not (tmp_path / "handle.bin").exists():
            assert time.time() - t0 < 110 and child.is_alive(), "the absent peer never exported its buffer"
            time.sleep(0.05)
        theirs = (tmp_path / "handle.bin").read_bytes()
        g, model, dm, _ = _setup()
        L = _lib.lib()
        W = 24
        X0 = synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"])
        ds = DeviceSampler([dm], W, seed=7)
        ds.set_state(X0)
        ds.run(4)                                        # a block before: the snapshot is not the initial state
        mine = (C.c_char * 64)()
        _lib.check(L.gpemu_sampler_peer_export(ds._h, C.cast(mine, C.c_void_p)))
        both = C.create_string_buffer(mine.raw + theirs, 128)
        _lib.check(L.gpemu_sampler_peer_import(ds._h, 2, 0, C.cast(both, C.c_void_p)))
        votes = []

        def vote(ok):
            votes.append(ok)
            return ok
        assert ds._run_peer_block(6, True, vote) is False          # lost, restored
        assert votes == [False]
        ds.run(6)                                                   # the rerun of the block, then the next block
        ds.run(3)
        ref = DeviceSampler([dm], W, seed=7)
        ref.set_state(X0)
        ref.run(13)
        np.testing.assert_array_equal(ds.get_chain()[0], ref.get_chain()[0])
        np.testing.assert_array_equal(ds.get_chain()[1], ref.get_chain()[1])
        np.testing.assert_array_equal(ds.counts()[0], ref.counts()[0])
        assert ds.counts()[1:] == ref.counts()[1:]
        ds.close(); ref.close(); dm.close()
    finally:
        (tmp_path / "done").write_text("x")
        child.join(60)
        if child.is_alive():
            child.kill()


def _multigroup_models():
    g = GU.load("g5_multigroup")
    models = {grp: GU.group_model(g, prefix=grp + "_") for grp in ("g1", "g2")}
    dms = []
    for grp, cols, bs in (("g1", g["cols_g1"], [0, 10, 22]), ("g2", g["cols_g2"], [0, 8])):
        dm = GU.device_model(models[grp])
        dm.likelihood_setup(g["y_exp"][cols], g["y_err"][cols], g["lo"], g["hi"], 1.0, block_start=bs)
        dms.append(dm)
    return g, dms


def _sharded_multigroup_worker(rank, world, port, out_dir, fused):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if not fused:
        os.environ["GPEMU_NO_FUSED"] = "1"       # read once per process by the library: set before its first use
    # (a model this small would be REPLICATED by default -- sampler.worth_sharding -- instead of sharded: the threshold
    # is taken away so that the default transport choice is what runs)
    os.environ["GPEMU_SHARD_MIN_GFLOP"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpemu.sampler import DeviceSampler
    g, dms = _multigroup_models()
    ds = DeviceSampler(dms, 16, seed=21)
    ds.set_state(g["Xq"])
    ds.run_sharded(4)                          # default transport
    ds.run_sharded(2)
    if fused:
        assert ds.last_transport == "peer" and all(ds._peer_ok.values()), "two groups should take the fused peer run"
    else:
        assert ds.last_transport == "torch" and not any(ds._peer_ok.values()), \
            "the ranks should have agreed NOT to take the peer transport"
        assert ds.transport_info.get("fallback_from_peer") is True
    chain, lps = ds.get_chain()
    np.save(os.path.join(out_dir, f"chain_{rank}.npy"), chain)
    np.save(os.path.join(out_dir, f"lp_{rank}.npy"), lps)
    dist.barrier()
    dist.destroy_process_group()
    ds.close()
    for dm in dms:
        dm.close()


@pytest.mark.parametrize("fused", [True, False])
def test_sharded_multigroup_two_ranks(tmp_path, fused):
    """Two emulation groups (the shipped analysis has three) on two ranks.  fused: the peer transport takes several
    groups (round 3: one front launch + one triangular GEMM per group per half-step).  not fused (GPEMU_NO_FUSED): both
    ranks agree to leave the peer path and exchange through torch.distributed.  Either way the chain is the single-GPU
    chain of the same sampler, bit for bit."""
    import os
    import torch.multiprocessing as mp
    from gpemu.sampler import DeviceSampler
    port = 29300 + (os.getpid() % 200) + (3 if fused else 0)
    mp.spawn(_sharded_multigroup_worker, args=(2, port, str(tmp_path), fused), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "chain_0.npy"), np.load(tmp_path / "chain_1.npy")
    np.testing.assert_array_equal(c0, c1)
    g, dms = _multigroup_models()
    ds = DeviceSampler(dms, 16, seed=21)
    ds.set_state(g["Xq"])
    ds.run(6)
    chain, lps = ds.get_chain()
    np.testing.assert_array_equal(chain, c0)
    np.testing.assert_array_equal(lps, np.load(tmp_path / "lp_0.npy"))
    ds.close()
    for dm in dms:
        dm.close()


def test_device_autocorrelation_time_equals_host_estimator():
    """emcee's integrated autocorrelation time (emcee/autocorr.py, call site ref: mcmc.py:111-119) estimated from the
    chain as it sits on the device -- direct lag products, a block of lags at a time until Sokal's window closes --
    against the host routine (FFT over the whole chain; itself tested on AR(1) series in test_sampler_host.py): same
    windows, same tau to 1e-10; also for a sub-range of steps, one chain of a stacked sampler, and the short-chain
    error path."""
    from gpemu import synthetic
    from gpemu import sampler as S
    g, model, dm, _ = _setup()
    W, steps = 64, 3000
    ds = S.DeviceSampler([dm], W, seed=5)
    ds.set_state(synthetic.make_walkers(W, seed=2, lo=g["lo"], hi=g["hi"]))
    ds.run(300, store=False)
    ds.run(steps)
    chain, _ = ds.get_chain()
    host = S.integrated_time(chain, quiet=True)
    dev = ds.integrated_time(quiet=True)
    assert np.all(np.isfinite(host)) and np.all(host > 1.0)
    np.testing.assert_allclose(dev, host, rtol=1e-10)
    # the blocks of f themselves against emcee's function_1d, walker by walker
    f = ds.acf_block(0, 48)
    ref = np.stack([np.mean([S.function_1d(chain[:, w, dd])[:48] for w in range(W)], axis=0) for dd in range(dm.d)], axis=1)
    np.testing.assert_allclose(f, ref, rtol=0, atol=1e-12)
    # a sub-range of the steps (discard) and a small lag block size (several blocks until the window closes)
    np.testing.assert_allclose(ds.integrated_time(first=500, quiet=True, block=16),
                               S.integrated_time(chain[500:], quiet=True), rtol=1e-10)
    # too short a chain: emcee's AutocorrError with the estimate attached
    with pytest.raises(S.AutocorrError) as err:
        ds.integrated_time(first=0, n=200)
    np.testing.assert_allclose(err.value.tau, S.integrated_time(chain[:200], quiet=True), rtol=1e-9)
    # a chain so short that the window never closes: emcee's auto_window then answers window 0, tau = 1.  (A walker that
    # never moved in the range makes its series 0 / 0: NaN here, rounding residue in the FFT routine -- not compared.)
    checked = 0
    for n_short in (12, 24, 40):
        moved = np.all(np.any(chain[:n_short] != chain[0], axis=0))
        if not moved:
            assert not np.any(np.isfinite(ds.integrated_time(first=0, n=n_short, quiet=True)))
            continue
        np.testing.assert_allclose(ds.integrated_time(first=0, n=n_short, quiet=True),
                                   S.integrated_time(chain[:n_short], quiet=True), rtol=1e-10)
        with pytest.raises(S.AutocorrError):
            ds.integrated_time(first=0, n=n_short)
        checked += 1
    assert checked >= 1
    # through the emcee facade: the device path is taken while the chain is on the device
    ds.close()
    # one chain of a stacked sampler: its own walkers only
    C = 3
    ys = g["y_exp"][None, :] + 0.05 * np.random.default_rng(1).normal(size=(C, g["y_exp"].size))
    dm.likelihood_setup(ys, g["y_err"], g["lo"], g["hi"], 1.0)
    ms = S.DeviceSampler([dm], 32, seeds=[3, 4, 5])
    ms.set_state(np.concatenate([synthetic.make_walkers(32, seed=30 + c, lo=g["lo"], hi=g["hi"]) for c in range(C)]))
    ms.run(1500)
    mc, _ = ms.get_chain()
    for c in range(C):
        np.testing.assert_allclose(ms.integrated_time(w0=32 * c, nw=32, quiet=True),
                                   S.integrated_time(mc[:, 32 * c:32 * (c + 1)], quiet=True), rtol=1e-10)
    ms.close()
    dm.close()
