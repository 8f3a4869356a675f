"""CPU tests of the sampler logic: Philox restatement, the stretch-move oracle (statistically, since
emcee itself is unavailable -> "parity unpinned"), the host ensemble, and the sharded (N > 1) path
with a world_size-2 gloo group."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import sampler_oracle as SO


def test_philox_known_answers():
    # Random123 known-answer vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        out = SO.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(o[0]) for o in out) == exp


def test_philox_library_matches_oracle():
    from gpemu import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for _ in range(20):
        c = [int(x) for x in rng.integers(0, 2 ** 32, 4)]
        k = [int(x) for x in rng.integers(0, 2 ** 32, 2)]
        out = (C.c_uint32 * 4)()
        assert L.gpemu_philox4x32(*c, *k, out) == 0
        ref = SO.philox4x32_10(*[np.array([x], dtype=np.uint32) for x in c], k[0], k[1])
        assert [int(o) for o in out] == [int(r[0]) for r in ref]


def _gauss_logp(mu, prec):
    def f(X):
        X = np.atleast_2d(X)
        r = X - mu
        return -0.5 * np.einsum("ni,ij,nj->n", r, prec, r)
    return f


@pytest.mark.parametrize("stream_cls", [SO.EmceeStream, SO.PhiloxStream])
def test_oracle_sampler_recovers_gaussian(stream_cls):
    d, W = 3, 64
    rng = np.random.default_rng(5)
    A = rng.normal(size=(d, d))
    cov = A @ A.T + np.eye(d)
    mu = np.array([1.0, -2.0, 0.5])
    f = _gauss_logp(mu, np.linalg.inv(cov))
    X0 = mu + rng.normal(size=(W, d))
    chain, lps, nacc = SO.run(X0, f, stream_cls(123), 1500)
    flat = chain[300:].reshape(-1, d)
    assert np.allclose(flat.mean(0), mu, atol=0.1)
    assert np.allclose(np.cov(flat.T), cov, rtol=0.15, atol=0.15)
    af = nacc / 1500
    assert 0.3 < af.mean() < 0.9
    # bookkeeping: stored log-probs are the log-probs of the stored positions
    np.testing.assert_allclose(lps[-1], f(chain[-1]), rtol=1e-12)


def test_host_ensemble_equals_oracle_with_emcee_stream():
    from gpemu.sampler import HostEnsemble
    d, W = 4, 32
    f = _gauss_logp(np.zeros(d), np.eye(d))
    X0 = np.random.default_rng(1).normal(size=(W, d))
    chain, lps, nacc = SO.run(X0, f, SO.EmceeStream(77), 50)
    he = HostEnsemble(W, d, f, seed=77)
    he.set_state(X0)
    he.run(50)
    np.testing.assert_array_equal(np.stack(he.chain), chain)
    np.testing.assert_array_equal(np.stack(he.lps), lps)
    np.testing.assert_array_equal(he.naccepted, nacc)


def test_logging_blocks_neither_fetch_the_chain_nor_the_ensemble():
    """LoggingEnsembleSampler at the shipped interval (a log line every 10 steps, ref: mcmc.py:187-204): the blocks between
    two log lines return no state, the acceptance fraction comes from the counters (the result cache -- a copy of the
    whole chain so far -- stays empty until somebody asks for the chain), and the chain is the one of an unbroken run."""
    from bayesian_inference.mcmc import LoggingEnsembleSampler
    from gpemu.sampler import EnsembleSampler
    d, W = 3, 16
    f = _gauss_logp(np.zeros(d), np.eye(d))
    X0 = np.random.default_rng(5).normal(size=(W, d))
    a = LoggingEnsembleSampler(W, d, f, seed=9, vectorize=True, sharded=False)
    state = a.run_mcmc(X0, 47, n_logging_steps=10)
    assert a._cache is None                                 # five log lines later: nothing was copied
    assert a.advance(None, 3, want_state=False) is None
    af = a.acceptance_fraction
    assert a._cache is None and af.shape == (W,) and a.iteration == 50
    b = EnsembleSampler(W, d, f, seed=9, vectorize=True, sharded=False)
    b.advance(X0, 50)
    np.testing.assert_array_equal(a.get_chain(), b.get_chain())
    np.testing.assert_array_equal(state.coords, b.get_chain()[46])
    np.testing.assert_array_equal(af, b.acceptance_fraction)


def test_host_ensemble_requires_enough_walkers():
    from gpemu.sampler import HostEnsemble
    with pytest.raises(RuntimeError):
        HostEnsemble(6, 4, lambda X: np.zeros(len(X)))


def test_nan_logprob_raises():
    from gpemu.sampler import HostEnsemble
    calls = {"n": 0}

    def f(X):
        calls["n"] += 1
        out = np.zeros(len(X))
        if calls["n"] > 2:
            out[0] = np.nan
        return out
    he = HostEnsemble(8, 2, f, seed=0)
    he.set_state(np.random.default_rng(0).normal(size=(8, 2)))
    with pytest.raises(ValueError):
        he.run(5)


def test_shard_bounds_cover_everything():
    from gpemu.sampler import shard_bounds
    for n in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi, per = shard_bounds(n, world, r)
                assert hi - lo <= per and lo == min(r * per, n)
                got += list(range(lo, hi))
            assert got == list(range(n))


def test_integrated_time_ar1():
    from gpemu.sampler import AutocorrError, integrated_time
    rng = np.random.default_rng(3)
    rho, n, W = 0.9, 20000, 8
    x = np.zeros((n, W, 1))
    e = rng.normal(size=(n, W, 1))
    for t in range(1, n):
        x[t] = rho * x[t - 1] + e[t]
    tau = integrated_time(x)
    assert abs(tau[0] - (1 + rho) / (1 - rho)) < 3.0       # 19 for rho = 0.9
    with pytest.raises(AutocorrError):
        integrated_time(x[:200])


# ---- world_size 2 over gloo: sharded evaluation == unsharded chain ----------------------------
def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpemu.sampler import HostEnsemble
    d, W = 3, 26        # half sizes 13: not divisible by 2 -> ragged shards
    f = _gauss_logp(np.array([0.5, -0.5, 0.0]), np.diag([1.0, 2.0, 0.5]))
    X0 = np.random.default_rng(9).normal(size=(W, d))
    he = HostEnsemble(W, d, f, seed=4, sharded=True)
    he.set_state(X0)
    he.run(30)
    np.save(os.path.join(out_dir, f"chain_{rank}.npy"), np.stack(he.chain))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_host_ensemble_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    from gpemu.sampler import HostEnsemble
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c0 = np.load(tmp_path / "chain_0.npy")
    c1 = np.load(tmp_path / "chain_1.npy")
    np.testing.assert_array_equal(c0, c1)            # every rank holds the same ensemble
    d, W = 3, 26
    f = _gauss_logp(np.array([0.5, -0.5, 0.0]), np.diag([1.0, 2.0, 0.5]))
    he = HostEnsemble(W, d, f, seed=4)
    he.set_state(np.random.default_rng(9).normal(size=(W, d)))
    he.run(30)
    np.testing.assert_array_equal(np.stack(he.chain), c0)   # and it equals the single-process chain


# ---- closure tests as independent replicas (one chain per rank) ---------------------------------
def _closure_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesian_inference import mcmc
    from gpemu.sampler import EnsembleSampler

    class Cfg:      # only what run_mcmc reads before it decides who owns the closure index
        parameterization = "p"
        analysis_config = {"parameterization": {"p": {"names": ["a", "b"], "min": [0.0, 0.0], "max": [1.0, 1.0]}}}

    skipped = []
    for idx in range(5):
        if mcmc.closure_owner(idx, world) != rank:
            assert mcmc.run_mcmc(Cfg(), closure_index=idx) is None      # returns before touching any file or GPU
            skipped.append(idx)
    # a replica sampler is unsharded even though torch.distributed has two ranks: different seeds per rank
    # give different chains and no collective is entered (a sharded run with unequal step counts would hang)
    f = _gauss_logp(np.array([0.2, -0.1]), np.eye(2))
    es = EnsembleSampler(12, 2, f, seed=100 + rank, vectorize=True, sharded=False)
    assert es.world_size == 1
    es.run_mcmc(np.random.default_rng(rank).normal(size=(12, 2)), 5 + 3 * rank)
    np.save(os.path.join(out_dir, f"skipped_{rank}.npy"), np.array(skipped))
    np.save(os.path.join(out_dir, f"chain_{rank}.npy"), es.get_chain())
    dist.barrier()
    dist.destroy_process_group()


def test_closure_tests_run_as_replicas_gloo_world2(tmp_path, monkeypatch):
    import torch.multiprocessing as mp
    from bayesian_inference import mcmc
    assert mcmc.closure_owner(-1, 8) is None and mcmc.closure_owner(3, 1) is None
    assert [mcmc.closure_owner(i, 4) for i in range(6)] == [0, 1, 2, 3, 0, 1]
    monkeypatch.setenv("GPEMU_CLOSURE_REPLICAS", "0")
    assert mcmc.closure_owner(3, 4) is None
    monkeypatch.delenv("GPEMU_CLOSURE_REPLICAS")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_closure_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "skipped_0.npy"), np.load(tmp_path / "skipped_1.npy")
    assert sorted(np.r_[s0, s1]) == [0, 1, 2, 3, 4] and list(s0) == [1, 3] and list(s1) == [0, 2, 4]
    assert np.load(tmp_path / "chain_0.npy").shape == (5, 12, 2)
    assert np.load(tmp_path / "chain_1.npy").shape == (8, 12, 2)


# ---- seed=None on a sharded chain: rank 0's draw is used everywhere; the group is joined from the env --------
def _unseeded_worker(rank, world, port, out_dir):
    # what torch.distributed.run exports; nothing here calls init_process_group: the drop-in joins by itself
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), GPEMU_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from bayesian_inference import mcmc
    from gpemu.sampler import EnsembleSampler
    assert not dist.is_initialized()
    assert mcmc._rank_world() == (rank, world)            # joins the launcher's group on first use
    assert dist.is_initialized() and dist.get_backend() == "gloo"
    np.random.seed(1000 + rank)                           # the ranks' global numpy states differ (as under torchrun)
    f = _gauss_logp(np.array([0.2, -0.1, 0.4]), np.diag([1.0, 0.5, 2.0]))
    es = EnsembleSampler(14, 3, f, vectorize=True)        # seed=None
    assert es.world_size == world
    es.run_mcmc(np.random.default_rng(5).normal(size=(14, 3)), 12)
    np.save(os.path.join(out_dir, f"chain_{rank}.npy"), es.get_chain())
    np.save(os.path.join(out_dir, f"seed_{rank}.npy"), np.array([es._seed]))
    dist.barrier()
    dist.destroy_process_group()


def test_unseeded_sharded_sampler_uses_rank0_seed_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    from gpemu.sampler import EnsembleSampler
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_unseeded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "chain_0.npy"), np.load(tmp_path / "chain_1.npy")
    s0, s1 = int(np.load(tmp_path / "seed_0.npy")[0]), int(np.load(tmp_path / "seed_1.npy")[0])
    assert s0 == s1
    np.testing.assert_array_equal(c0, c1)
    f = _gauss_logp(np.array([0.2, -0.1, 0.4]), np.diag([1.0, 0.5, 2.0]))
    es = EnsembleSampler(14, 3, f, vectorize=True, seed=s0, sharded=False)
    es.run_mcmc(np.random.default_rng(5).normal(size=(14, 3)), 12)
    np.testing.assert_array_equal(es.get_chain(), c0)     # the single-process chain for that seed


def test_small_models_are_replicated_not_sharded(monkeypatch):
    """DeviceSampler.worth_sharding: the C3 shape (5.4 GFLOP per half-step) shards, the reference's
    shipped three-group shape (0.27 GFLOP: a step is ~10 us launches, which the fused half-step of a sharded run adds to)
    does not -- every rank then runs the chain itself; GPEMU_SHARD_MIN_GFLOP moves the threshold."""
    import types
    from gpemu.sampler import DeviceSampler
    monkeypatch.delenv("GPEMU_SHARD_MIN_GFLOP", raising=False)
    mk = lambda N, k: types.SimpleNamespace(N=N, k=k)
    c3 = types.SimpleNamespace(models=[mk(1000, 10)], ns=(512, 512), n_chains=1)
    shipped = types.SimpleNamespace(models=[mk(150, 5), mk(150, 11), mk(150, 25)], ns=(100, 100), n_chains=1)
    assert DeviceSampler.worth_sharding(c3) is True
    assert DeviceSampler.worth_sharding(shipped) is False
    monkeypatch.setenv("GPEMU_SHARD_MIN_GFLOP", "0")
    assert DeviceSampler.worth_sharding(shipped) is True
