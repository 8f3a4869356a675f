"""CPU tests of the host-side mirror of the reference interface (no GPU compute)."""
import os
import pickle

import numpy as np
import pytest

import dropin_util as DU
import golden_util as GU
from oracle import gp_oracle as O


def test_modules_import_and_expose_reference_names():
    from bayesian_inference import emulation, log_posterior, mcmc
    for name in ("fit_emulators", "fit_emulator_group", "read_emulators", "write_emulators",
                 "compute_emulator_cov_unexplained", "compute_emulator_group_cov_unexplained", "nd_block_diag",
                 "SortEmulationGroupObservables", "predict", "predict_emulation_group", "EmulationGroupConfig",
                 "EmulationConfig"):
        assert hasattr(emulation, name), name
    for name in ("initialize_pool_variables", "log_posterior"):
        assert hasattr(log_posterior, name), name
    for name in ("run_mcmc", "credible_interval", "map_parameters", "LoggingEnsembleSampler", "MCMCConfig"):
        assert hasattr(mcmc, name), name


def test_config_classes(tmp_path):
    from bayesian_inference import emulation, mcmc
    path, analysis = DU.write_config(tmp_path, kernels_active=("matern", "noise"), n_pc=7, n_restarts=3)
    ec = emulation.EmulationConfig.from_config_file(analysis_name="test_analysis", parameterization="exponential",
                                                    analysis_config=analysis, config_file=path)
    assert list(ec.emulation_groups_config) == ["main"]
    g = ec.emulation_groups_config["main"]
    assert g.n_pc == 7 and g.n_restarts == 3 and g.alpha == 1e-10 and g.force_retrain is False
    assert list(g.active_kernels) == ["matern", "noise"] and g.max_n_components_to_calculate is None
    assert g.emulation_outputfile.endswith(os.path.join("test_analysis_exponential", "emulation_group_main.pkl"))
    assert ec.output_dir == g.output_dir and ec.observables_filename == "observables.h5"
    k = emulation.build_kernel(g)
    assert k.kind == 1 and k.nu == 1.5 and k.has_noise and not k.has_const
    np.testing.assert_allclose(k.length_scale, np.array([0.4, 9, 9.9933, 9.9933, 1.5, 99.95]))
    np.testing.assert_allclose(k.bounds[:6], np.log(np.outer(k.length_scale, [0.01, 100])))
    np.testing.assert_allclose(k.bounds[6], np.log([1e-3, 10.0]))
    assert repr(k).startswith("Matern(length_scale=[0.4, 9, 9.99, 9.99, 1.5, 100], nu=1.5) + WhiteKernel(")
    mc = mcmc.MCMCConfig(analysis_name="test_analysis", parameterization="exponential", analysis_config=analysis,
                         config_file=path, closure_index=3)
    assert (mc.n_walkers, mc.n_burn_steps, mc.n_sampling_steps, mc.n_logging_steps) == (24, 8, 12, 5)
    assert mc.mcmc_outputfile.endswith(os.path.join("closure/results/3", "mcmc.h5"))
    assert mc.sampler_outputfile.endswith("mcmc_sampler.pkl")
    with pytest.raises(AssertionError):
        DU_path, an = DU.write_config(tmp_path, kernels_active=("rbf", "matern"))
        emulation.EmulationGroupConfig("test_analysis", "exponential", an, DU_path, "main")


def test_sort_observables_convert_equals_oracle_merge():
    from bayesian_inference import emulation
    rng = np.random.default_rng(0)
    mapping = {"A": ("g1", slice(0, 10), slice(0, 10)), "B": ("g2", slice(10, 18), slice(0, 8)),
               "C": ("g1", slice(18, 30), slice(10, 22))}
    groups = {"g1": {"central_value": rng.normal(size=(3, 22)), "cov": rng.normal(size=(3, 22, 22))},
              "g2": {"central_value": rng.normal(size=(3, 8)), "cov": rng.normal(size=(3, 8, 8))}}
    s = emulation.SortEmulationGroupObservables(mapping, (5, 30))
    out = s.convert(groups)
    ref = O.merge_groups(groups, mapping, 30)
    np.testing.assert_array_equal(out["central_value"], ref["central_value"])
    np.testing.assert_array_equal(out["cov"], ref["cov"])
    cols, starts = s.group_layout("g1")
    np.testing.assert_array_equal(cols, np.r_[0:10, 18:30])
    np.testing.assert_array_equal(starts, [0, 10, 22])
    np.testing.assert_array_equal(s.group_layout("g2")[1], [0, 8])
    blk = emulation.nd_block_diag([np.ones((2, 3, 3)), 2 * np.ones((2, 1, 1))])
    assert blk.shape == (2, 4, 4) and blk[1, 3, 3] == 2 and blk[0, 0, 3] == 0


def test_credible_interval_and_map():
    from bayesian_inference import mcmc
    rng = np.random.default_rng(1)
    x = rng.normal(size=20001)
    lo, hi = mcmc.credible_interval(x, 0.9, "quantile")
    np.testing.assert_allclose([lo, hi], np.quantile(x, [0.05, 0.95]))
    l2, h2 = mcmc.credible_interval(x, 0.9, "hpd")
    assert h2 - l2 <= hi - lo + 1e-12 and abs((h2 - l2) - 3.29) < 0.1
    post = rng.normal(loc=[1.0, -2.0], scale=[0.5, 2.0], size=(50000, 2))
    mp = mcmc.map_parameters(post)
    assert abs(mp[0] - 1.0) < 0.02 and abs(mp[1] + 2.0) < 0.1


def test_estimators_pickle_and_transforms():
    from gpemu import estimators as E
    sc = E.StandardScaler()
    sc.mean_, sc.scale_, sc.var_ = np.array([1.0, 2.0]), np.array([2.0, 4.0]), np.array([4.0, 16.0])
    X = np.array([[3.0, 6.0]])
    np.testing.assert_allclose(sc.inverse_transform(sc.transform(X)), X)
    k = E.ARDKernel(0, [1.0, 2.0], [[0.01, 100], [0.02, 200]], constant_value=4.0, constant_value_bounds=(1e-3, 1e3),
                    noise_level=0.1, noise_level_bounds=(1e-3, 10))
    assert k.n_dims == 4 and repr(k) == "RBF(length_scale=[1, 2]) + 2**2 + WhiteKernel(noise_level=0.1)"
    th = k.theta.copy()
    k.theta = th + 0.5
    np.testing.assert_allclose(k.theta, th + 0.5)
    g = E.GaussianProcessRegressor(k, alpha=1e-10, n_restarts_optimizer=2)
    g.kernel_, g.alpha_, g.L_, g.X_train_ = k.clone(), np.zeros(3), np.eye(3), np.zeros((3, 2))
    g2 = pickle.loads(pickle.dumps({"emulators": [g], "PCA": {"scaler": sc}}))
    assert g2["emulators"][0]._dev is None and g2["emulators"][0].kernel_.diag_value() == k.diag_value()


_CALLS = {"n": 0}


def _gauss_logp_row(x):
    _CALLS["n"] += 1
    assert x.shape == (2,)
    return np.array([-0.5 * np.sum(x ** 2)])        # shape (1,) like the reference's log_posterior


def test_ensemble_sampler_facade_host_backend_emcee_conventions():
    """Generic callable: one call per walker like emcee, state unpacking, flat chains, reset, pickling."""
    from gpemu.sampler import AutocorrError, EnsembleSampler
    d, W = 2, 12
    calls, logp = _CALLS, _gauss_logp_row
    calls["n"] = 0
    s = EnsembleSampler(W, d, logp, seed=3)
    X0 = np.random.default_rng(0).normal(size=(W, d))
    st = s.run_mcmc(X0, 7)
    assert st[0].shape == (W, d) and st.log_prob.shape == (W,)
    assert calls["n"] == W + 7 * W
    assert s.get_chain().shape == (7, W, d) and s.flatchain.shape == (7 * W, d)
    np.testing.assert_array_equal(s.flatchain[W:2 * W], s.get_chain()[1])
    assert s.flatlnprobability.shape == (7 * W,) and s.iteration == 7
    assert np.all((s.acceptance_fraction >= 0) & (s.acceptance_fraction <= 1))
    with pytest.raises(AutocorrError):
        s.get_autocorr_time()
    n = sum(1 for _ in s.sample(None, iterations=3))
    assert n == 3 and s.iteration == 10
    s2 = pickle.loads(pickle.dumps(s))
    np.testing.assert_array_equal(s2.get_chain(), s.get_chain())
    s.reset()
    assert s.get_chain().shape[0] == 0 and s.iteration == 0
    with pytest.raises(RuntimeError):
        EnsembleSampler(3, 2, logp).run_mcmc(np.zeros((3, 2)), 1)
    with pytest.raises(ValueError):
        EnsembleSampler(8, 2, logp).run_mcmc(np.ones((8, 2)), 1)     # degenerate walker cloud


def test_product_fails_loudly_without_gpu():
    from gpemu import _lib
    from gpemu.model import DeviceModel
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.GpemuError):
        DeviceModel(np.zeros((4, 2)), np.ones((1, 2)), np.zeros((1, 4)), np.eye(4)[None], np.ones((1, 3)),
                    np.zeros(3), np.ones(3))
    from gpemu.fit import pca_fit
    with pytest.raises(_lib.GpemuError):
        pca_fit(np.zeros((4, 3)))


def test_library_exports_every_declared_symbol():
    import re
    from gpemu import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "gpemu.h")).read()
    declared = set(re.findall(r"\b(gpemu_[a-z0-9_]+)\s*\(", hdr))
    L = _lib.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())


# ---- learn_mapping against the reference's own run on its fixture (golden G6) ------------------------------
@pytest.mark.skipif(not os.path.exists("/root/reference/src/bayesian_inference/data_IO.py"),
                    reason="needs the reference's data_IO (build container only)")
def test_learn_mapping_matches_the_reference_on_its_fixture(monkeypatch):
    """SortEmulationGroupObservables.learn_mapping (ref: emulation.py:289-344) through the UNTOUCHED reference
    data_IO -- imported without silx thanks to gpemu.h5io's stand-in, reading the reference's observables.h5 with
    the built-in HDF5 reader -- for a two-group split, against the reference's own result (tests/golden/
    make_learn_mapping_golden.py)."""
    import importlib.util
    import sys
    import bayesian_inference
    from bayesian_inference import emulation
    HERE = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("bayesian_inference.data_IO",
                                                  "/root/reference/src/bayesian_inference/data_IO.py")
    ref_io = importlib.util.module_from_spec(spec)
    monkeypatch.setitem(sys.modules, "bayesian_inference.data_IO", ref_io)
    monkeypatch.setattr(bayesian_inference, "data_IO", ref_io, raising=False)
    spec.loader.exec_module(ref_io)
    g = dict(np.load(os.path.join(HERE, "golden", "g6_learn_mapping.npz")))

    class GroupCfg:
        def __init__(self, include):
            self.observable_filter = ref_io.ObservableFilter(include_list=include.split(";"), exclude_list=[])

    class EmuCfg:
        output_dir = "/root/reference/tests/test_data"
        emulation_groups_config = {str(n): GroupCfg(str(i)) for n, i in zip(g["group_names"], g["include"])}

    m = emulation.SortEmulationGroupObservables.learn_mapping(EmuCfg())
    assert list(m.emulation_group_to_observable_matrix) == [str(k) for k in g["observables"]]      # sorted order
    assert tuple(m.shape) == tuple(int(v) for v in g["shape"])
    for i, key in enumerate(g["observables"]):
        grp, so, sg = m.emulation_group_to_observable_matrix[str(key)]
        assert grp == str(g["group"][i])
        assert (so.start, so.stop) == (int(g["out_start"][i]), int(g["out_stop"][i]))
        assert (sg.start, sg.stop) == (int(g["grp_start"][i]), int(g["grp_stop"][i]))
    # the layout handed to the device likelihood: this group's columns in the merged order, observable block starts
    for name in m_groups(g):
        cols, starts = m.group_layout(name)
        sel = [i for i in range(len(g["group"])) if str(g["group"][i]) == name]
        want = np.concatenate([np.arange(g["out_start"][i], g["out_stop"][i]) for i in sel])
        np.testing.assert_array_equal(np.sort(cols), np.sort(want))
        assert starts[0] == 0 and starts[-1] == len(want) and len(starts) == len(sel) + 1


def m_groups(g):
    return [str(n) for n in g["group_names"]]


def test_lbfgsb_lockstep_driver_equals_scipy_minimize():
    """The thread-free lock-step driver (gpemu.estimators._LbfgsbRun: explicit state around scipy's L-BFGS-B routine, cut
    where it asks for a value) asks for the same points in the same order and returns the same result as
    scipy.optimize.minimize(method="L-BFGS-B", jac=True, bounds=...), which is what sklearn's GPR calls
    (skl _gpr.py:655-668) -- also when the objective answers +inf (a kernel matrix that is not positive definite)."""
    import scipy.optimize
    from gpemu import estimators as E
    if not E._setulb_driver_ok():
        pytest.skip("scipy's setulb has another signature here: fit_gps uses one minimize() per thread")
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6))
    A = A @ A.T + np.eye(6)
    b = rng.normal(size=6)
    for trial in range(6):
        calls = []

        def fun(x):
            calls.append(x.copy())
            if trial >= 4 and x[0] > 0.25:                       # part of the box is "not positive definite"
                return np.inf, np.zeros_like(x)
            return 0.5 * x @ A @ x - b @ x + 0.1 * np.sum(x ** 4), A @ x - b + 0.4 * x ** 3
        bounds = np.array([[-1.0, 0.3 + 0.2 * trial]] * 6)
        x0 = rng.uniform(-2, 2, 6)
        ref = scipy.optimize.minimize(fun, x0, method="L-BFGS-B", jac=True, bounds=bounds)
        ref_calls = list(calls)
        calls.clear()
        run = E._LbfgsbRun(x0, bounds)
        while True:
            x = run.advance()
            if x is None:
                break
            f, g = fun(x)
            run.supply(x, f, g)
        assert len(calls) == len(ref_calls) and all(np.array_equal(a, c) for a, c in zip(calls, ref_calls))
        np.testing.assert_array_equal(run.x, ref.x)
        assert run.f == ref.fun and run.nit == ref.nit and run.nfev == ref.nfev and run.status == ref.status


def test_lockstep_groups_alternating_on_the_device_give_the_sequential_results():
    """_lockstep_minimise with two groups of runs alternating on the "device" (a host stand-in evaluating an analytic
    objective per target, called from the driver's worker thread): every run ends where scipy.optimize.minimize ends
    on its own, the results come back in the order given, and every batch respects max_batch."""
    import threading
    import scipy.optimize
    from gpemu import estimators as E
    if not E._setulb_driver_ok():
        pytest.skip("scipy's setulb has another signature here")
    rng = np.random.default_rng(4)
    n, d = 23, 5
    mats = [(lambda M: M @ M.T + np.eye(d))(rng.normal(size=(d, d))) for _ in range(n)]
    vecs = [rng.normal(size=d) for _ in range(n)]
    bounds = np.array([[-1.5, 1.0]] * d)
    starts = [rng.uniform(-1.5, 1.0, d) for _ in range(n)]

    def objective(t, x):        # minimised; the driver is handed (lml, grad) = (-f, -g)
        A, b = mats[t], vecs[t]
        return 0.5 * x @ A @ x - b @ x + 0.05 * np.sum(x ** 4), A @ x - b + 0.2 * x ** 3

    class Device:
        def __init__(self):
            self.sizes, self.threads = [], set()

        def lml_batch(self, ys, thetas):
            self.sizes.append(len(ys))
            self.threads.add(threading.get_ident())
            out = [objective(int(y[0]), x) for y, x in zip(ys, thetas)]
            return (np.array([-f for f, _ in out]), np.array([-g for _, g in out]), np.zeros(len(ys), dtype=np.int32))

    problems = [(np.array([float(t)]), starts[t]) for t in range(n)]
    ref = [scipy.optimize.minimize(lambda x, t=t: objective(t, x), starts[t], method="L-BFGS-B", jac=True, bounds=bounds)
           for t in range(n)]
    for groups in (1, 2):
        dev = Device()
        got, _busy = E._lockstep_minimise(dev, problems, bounds, max_batch=4, groups=groups)
        assert max(dev.sizes) <= 4 and len(got) == n
        assert threading.get_ident() not in dev.threads            # evaluated on the driver's device thread
        for t in range(n):
            np.testing.assert_array_equal(got[t][0], ref[t].x)
            assert got[t][1] == ref[t].fun
    # two handles: group 1's evaluations go through the second one, on a second worker thread; same results, and the
    # driver reports the time with at least one evaluation in flight
    dev, dev2 = Device(), Device()
    got, busy = E._lockstep_minimise(dev, problems, bounds, max_batch=4, groups=2, second=dev2)
    assert dev.sizes and dev2.sizes and max(dev.sizes + dev2.sizes) <= 4
    assert threading.get_ident() not in (dev.threads | dev2.threads)
    for t in range(n):
        np.testing.assert_array_equal(got[t][0], ref[t].x)
        assert got[t][1] == ref[t].fun
    assert busy > 0.0                      # returned, not kept in a function attribute
    # three groups, three handles
    devs = [Device(), Device(), Device()]
    got, _busy = E._lockstep_minimise(devs[0], problems, bounds, max_batch=4, groups=3, second=devs[1:])
    assert all(d.sizes for d in devs) and max(sum((d.sizes for d in devs), [])) <= 4
    for t in range(n):
        np.testing.assert_array_equal(got[t][0], ref[t].x)
        assert got[t][1] == ref[t].fun
    # fewer problems than one batch: one group, nothing to alternate with (the second handle stays unused)
    dev, dev2 = Device(), Device()
    got, _busy = E._lockstep_minimise(dev, problems[:3], bounds, max_batch=4, second=dev2)
    assert all(np.array_equal(got[t][0], ref[t].x) for t in range(3)) and not dev2.sizes


def test_rank_query_of_a_single_process_does_not_import_torch():
    """`gpemu.dist.rank_world()` in a process that no launcher started and that has not imported torch: (0, 1), and
    torch stays unimported (it costs ~1 s, a third of a C3 `fit_emulators`); under a launcher's environment the group is
    looked for as before."""
    import subprocess
    import sys
    code = ("import sys, os; sys.path.insert(0, %r); "
            "[os.environ.pop(k, None) for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')]; "
            "from gpemu import dist as gd; print(gd.rank_world(), 'torch' in sys.modules); "
            "os.environ.update(RANK='0', WORLD_SIZE='2'); print(gd._dist() is not None)") % os.path.join(
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesian-inference_amd")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "(0, 1) False"
    assert lines[1] == "True"          # a launcher's environment: torch.distributed is consulted


def test_malformed_world_size_counts_as_one(monkeypatch):
    """An empty or non-numeric WORLD_SIZE (some schedulers export it so) is a single process, not a ValueError from the
    first sampler call (gpemu/dist.py)."""
    from gpemu import dist
    for bad in ("", "abc", "0", "-3"):
        monkeypatch.setenv("WORLD_SIZE", bad)
        assert dist._env_world() == 1
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert dist._env_world() == 4
    monkeypatch.delenv("WORLD_SIZE")
    assert dist._env_world() == 1
