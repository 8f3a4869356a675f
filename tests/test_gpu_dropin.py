"""-m gpu tests of the drop-in modules through the reference's call signatures
(emulation.fit_emulator_group / predict / predict_emulation_group, log_posterior.log_posterior,
mcmc.run_mcmc) against the goldens produced by running the reference."""
import pickle

import numpy as np
import pytest

import dropin_util as DU
import golden_util as GU
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-8


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


_results_at_golden_theta = DU.results_at_golden_theta


class _GroupCfg:
    def __init__(self, n_pc):
        self.n_pc = n_pc


class _EmuCfg:
    def __init__(self, groups, sorter):
        self.emulation_groups_config = groups
        self.sort_observables_in_matrix = sorter


@pytest.mark.parametrize("name", ["g1_rbf_noise", "g1_matern25_const_noise", "g2_rbf_noise"])
def test_predict_and_log_posterior_module_api(name):
    from bayesian_inference import emulation, log_posterior
    g = GU.load(name)
    res = _results_at_golden_theta(g, None)
    cfg = _GroupCfg(int(g["n_pc"]))
    Xq = g["Xq"]
    # single estimator API (ref: emulation.py:497)
    m0, s0 = res["emulators"][0].predict(Xq, return_std=True)
    assert relerr(m0, g["gp_mean"][:, 0]) < TOL and np.max(np.abs(s0 ** 2 - g["gp_var"][:, 0])) < TOL
    # predict_emulation_group: batch and single-row semantics (cov_unexplained / n_samples)
    cu = emulation.compute_emulator_group_cov_unexplained(cfg, res)
    assert relerr(cu, g["cov_unexplained"]) < 1e-12
    pb = emulation.predict_emulation_group(Xq, res, cfg, emulator_group_cov_unexplained=cu)
    assert relerr(pb["central_value"], g["batch_central_value"]) < TOL
    assert relerr(pb["cov"][:g["batch_cov_head"].shape[0]], g["batch_cov_head"]) < TOL
    p1 = emulation.predict_emulation_group(Xq[0], res, cfg)
    assert relerr(p1["cov"][0], g["single_cov_head"][0]) < TOL
    # predict() over groups + log_posterior through the module globals
    emu_cfg = _EmuCfg({"g": cfg}, DU.TrivialSort("g"))
    merged = emulation.predict(Xq[:4], emu_cfg, emulation_group_results={"g": res})
    assert merged["cov"].shape == (4, g["Y"].shape[1], g["Y"].shape[1])
    log_posterior.initialize_pool_variables(g["lo"], g["hi"], emu_cfg, {"g": res},
                                            {"y": g["y_exp"], "y_err": g["y_err"]}, None)
    nper = g["logpost_per_walker"].shape[0]
    per = np.array([log_posterior.log_posterior(Xq[i])[0] for i in range(min(nper, 12))])
    np.testing.assert_allclose(per, g["logpost_per_walker"][:per.size], rtol=TOL)
    np.testing.assert_allclose(log_posterior.log_posterior(Xq), g["logpost_batched"], rtol=TOL)
    mixed = log_posterior.log_posterior(g["X_mixed"])
    assert np.array_equal(np.isneginf(mixed), np.isneginf(g["logpost_mixed"]))
    fin = np.isfinite(mixed)
    np.testing.assert_allclose(mixed[fin], g["logpost_mixed"][fin], rtol=TOL)
    assert log_posterior.log_posterior(Xq[3]).shape == (1,)


def test_multigroup_log_posterior_module_api():
    from bayesian_inference import emulation, log_posterior
    g = GU.load("g5_multigroup")
    mapping = {"A": ("g1", slice(0, 10), slice(0, 10)), "B": ("g2", slice(10, 18), slice(0, 8)),
               "C": ("g1", slice(18, 30), slice(10, 22))}
    sorter = emulation.SortEmulationGroupObservables(mapping, (60, 30))
    res, cfgs = {}, {}
    for grp in ("g1", "g2"):
        sub = {k[len(grp) + 1:]: v for k, v in g.items() if k.startswith(grp + "_")}
        sub.update(design=g["design"], gpr_alpha=g["gpr_alpha"])
        res[grp] = _results_at_golden_theta(sub, None)
        cfgs[grp] = _GroupCfg(int(sub["n_pc"]))
    emu_cfg = _EmuCfg(cfgs, sorter)
    Xq = g["Xq"]
    merged = emulation.predict(Xq, emu_cfg, emulation_group_results=res)
    assert relerr(merged["central_value"], g["merged_central_value"]) < TOL
    assert relerr(merged["cov"][:2], g["merged_cov_head"]) < TOL
    log_posterior.initialize_pool_variables(g["lo"], g["hi"], emu_cfg, res, {"y": g["y_exp"], "y_err": g["y_err"]}, None)
    np.testing.assert_allclose(log_posterior.log_posterior(Xq), g["logpost_batched"], rtol=TOL)
    per = np.array([log_posterior.log_posterior(Xq[i])[0] for i in range(6)])
    np.testing.assert_allclose(per, g["logpost_per_walker"][:6], rtol=TOL)


def test_fit_emulator_group_end_to_end(tmp_path):
    """The whole fit (device PCA + device LML/gradient under host L-BFGS-B with restarts) lands on the
    reference's optimum: same seeds for the restarts, so the same basins."""
    from bayesian_inference import emulation
    g = GU.load("g1_rbf_noise")
    DU.install_fake_data_IO(g["Y"], g["design"], g["y_exp"], g["y_err"], {})
    path, analysis = DU.write_config(tmp_path, kernels_active=("rbf", "noise"), n_pc=5, n_restarts=2)
    ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", path, analysis)
    np.random.seed(12345)
    emulation.fit_emulators(ec)
    res = emulation.read_emulators(ec.emulation_groups_config["main"])
    assert set(res["PCA"]) == {"Y", "Y_pca", "Y_pca_truncated", "Y_reconstructed_truncated",
                               "Y_reconstructed_truncated_unscaled", "pca", "scaler"}
    assert relerr(res["PCA"]["Y_pca_truncated"], g["Y_pca_truncated"]) < 1e-9
    assert relerr(res["PCA"]["Y_reconstructed_truncated_unscaled"], g["Y_reconstructed_truncated_unscaled"]) < 1e-9
    jitter = float(g["gpr_alpha"])
    agree, dth = DU.certify_fit_against_reference(res["emulators"], g["theta"], g["lml_value"], "G1 fit", g["design"],
                                                  g["Y_pca_truncated"], jitter)
    cfg = ec.emulation_groups_config["main"]
    for i, e in enumerate(res["emulators"]):
        if agree[i]:
            m, sd = e.predict(g["Xq"], return_std=True)
            assert np.max(np.abs(m - g["gp_mean"][:, i])) < 1e-6 * max(1.0, np.max(np.abs(g["gp_mean"][:, i])))
            assert np.max(np.abs(sd ** 2 - g["gp_var"][:, i])) < 1e-6 * max(1.0, np.max(np.abs(g["gp_var"][:, i])))
    p = emulation.predict_emulation_group(g["Xq"], res, cfg)
    nh = g["batch_cov_head"].shape[0]
    # every GP, wherever its optimiser stopped: the reference's arithmetic AT THE DEVICE'S THETA, 1e-6
    om = DU.oracle_group_at(res["emulators"], g["design"], g["Y_pca_truncated"], g["pca_components"],
                            g["pca_explained_variance"], g["scaler_mean"], g["scaler_scale"], jitter)
    po = O.predict_group(g["Xq"], om)
    assert relerr(p["central_value"], po["central_value"]) < 1e-6
    assert relerr(p["cov"], po["cov"]) < 1e-6
    if agree.all():                                  # ... and the reference's own outputs where the optima coincide
        assert relerr(p["central_value"], g["batch_central_value"]) < 1e-6
        assert relerr(p["cov"][:nh], g["batch_cov_head"]) < 1e-6
    # a second call returns {} and does not overwrite (checkpoint behaviour, ref: emulation.py:64-70)
    assert emulation.fit_emulator_group(cfg) == {}


def test_run_mcmc_end_to_end(tmp_path, monkeypatch):
    from bayesian_inference import emulation, mcmc
    g = GU.load("g1_rbf_noise")
    written = {}
    DU.install_fake_data_IO(g["Y"], g["design"], g["y_exp"], g["y_err"], written)
    path, analysis = DU.write_config(tmp_path, n_pc=5, n_restarts=0)
    ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", path, analysis)
    ec._sort_observables_in_matrix = None
    np.random.seed(1)
    emulation.fit_emulators(ec)
    # single group: the merged matrix is the group matrix
    monkeypatch.setattr(emulation.EmulationConfig, "sort_observables_in_matrix",
                        property(lambda self: DU.TrivialSort("main")))
    monkeypatch.setattr(emulation.EmulationConfig, "observable_filter", property(lambda self: None))
    cfg = mcmc.MCMCConfig("test_analysis", "exponential", analysis, path)
    mcmc.run_mcmc(cfg)
    out = written[cfg.mcmc_outputfile]
    W, steps, d = cfg.n_walkers, cfg.n_sampling_steps, 6
    assert out["chain"].shape == (steps, W, d) and out["log_prob"].shape == (steps, W)
    assert out["acceptance_fraction"].shape == (W,)
    # 12 steps: far too short a chain (emcee's AutocorrError -> None, ref: mcmc.py:115-119); a walker that never moved
    # makes the estimate 0 / 0 = NaN instead, which emcee would hand through as well
    tau = out["autocorrelation_time"]
    assert tau is None or not np.any(np.isfinite(tau))
    lo, hi = np.array(g["lo"]), np.array(g["hi"])
    assert np.all(out["chain"] > lo) and np.all(out["chain"] < hi) and np.all(np.isfinite(out["log_prob"]))
    # stored log-probs are the (single-walker semantics) log-posterior of the stored positions
    from bayesian_inference import log_posterior
    lp = np.array([log_posterior.log_posterior(x)[0] for x in out["chain"][-1][:5]])
    np.testing.assert_allclose(lp, out["log_prob"][-1][:5], rtol=1e-10)
    sampler = pickle.load(open(cfg.sampler_outputfile, "rb"))
    np.testing.assert_array_equal(sampler.get_chain(), out["chain"])
    # mcmc.h5 on disk (silx layout, written without silx): what plot_mcmc.py:44-58 reads back
    from gpemu import h5io
    back = h5io.read_dict_from_h5(cfg.mcmc_output_dir, cfg.mcmc_outputfilename)
    assert set(back) == {"chain", "acceptance_fraction", "log_prob", "autocorrelation_time"}
    np.testing.assert_array_equal(back["chain"], out["chain"])
    np.testing.assert_array_equal(back["log_prob"], out["log_prob"])
    np.testing.assert_array_equal(back["acceptance_fraction"], out["acceptance_fraction"])
    if tau is None:
        assert back["autocorrelation_time"] == {}         # None -> empty group, as silx writes it


def test_closure_tests_run_stacked(tmp_path, monkeypatch):
    """The reference's closure loop (ref: steer_analysis.py:168-183) calls run_mcmc once per validation point: the
    first call runs all chains stacked in one multi-chain sampler and writes every chain's files, the later calls
    return at once.  Every chain's stored log-probabilities are the log-posterior of its stored positions under ITS
    pseudo-data."""
    from bayesian_inference import emulation, log_posterior, mcmc
    from gpemu import h5io
    g = GU.load("g1_rbf_noise")
    written = {}
    DU.install_fake_data_IO(g["Y"], g["design"], g["y_exp"], g["y_err"], written)
    path, analysis = DU.write_config(tmp_path, n_pc=5, n_restarts=0)
    analysis["validation_indices"] = [0, 3]
    ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", path, analysis)
    np.random.seed(3)
    emulation.fit_emulators(ec)
    monkeypatch.setattr(emulation.EmulationConfig, "sort_observables_in_matrix",
                        property(lambda self: DU.TrivialSort("main")))
    monkeypatch.setattr(emulation.EmulationConfig, "observable_filter", property(lambda self: None))
    mcmc._closure_done.clear()
    cfgs = [mcmc.MCMCConfig("test_analysis", "exponential", analysis, path, closure_index=j) for j in range(3)]
    np.random.seed(11)
    mcmc.run_mcmc(cfgs[0], closure_index=0)                      # runs chains 0, 1, 2
    assert all(c.mcmc_outputfile in written for c in cfgs)
    n_before = len(written)
    mcmc.run_mcmc(cfgs[1], closure_index=1)                      # nothing left to do
    mcmc.run_mcmc(cfgs[2], closure_index=2)
    assert len(written) == n_before
    W, steps, d = cfgs[0].n_walkers, cfgs[0].n_sampling_steps, 6
    lo, hi = np.array(g["lo"]), np.array(g["hi"])
    ec2 = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", path, analysis)
    res = ec2.read_all_emulator_groups()
    pseudo = []
    for j, c in enumerate(cfgs):
        out = written[c.mcmc_outputfile]
        assert out["chain"].shape == (steps, W, d) and out["log_prob"].shape == (steps, W)
        assert np.all(out["chain"] > lo) and np.all(out["chain"] < hi)
        assert out["design_point"].shape == (d,) and set(out["experimental_pseudodata"]) == {"y", "y_err"}
        back = h5io.read_dict_from_h5(c.mcmc_output_dir, "mcmc.h5")
        np.testing.assert_array_equal(back["chain"], out["chain"])
        np.testing.assert_array_equal(back["experimental_pseudodata"]["y"], out["experimental_pseudodata"]["y"])
        # the chain's log-probabilities under its own pseudo-data (single-walker semantics)
        log_posterior.initialize_pool_variables(g["lo"], g["hi"], ec2, res, out["experimental_pseudodata"], None)
        lp = np.array([log_posterior.log_posterior(x)[0] for x in out["chain"][-1][:4]])
        np.testing.assert_allclose(lp, out["log_prob"][-1][:4], rtol=1e-10)
        pseudo.append(out["experimental_pseudodata"]["y"])
        sampler = pickle.load(open(c.sampler_outputfile, "rb"))
        np.testing.assert_array_equal(sampler.get_chain(), out["chain"])
    assert not np.array_equal(pseudo[0], pseudo[1])              # every chain has its own draw
    # a second pass over the loop reruns everything, like the reference -- here in sub-batches of one chain
    # (chain-memory budget) -- and a repeated stand-alone request for a chain that was handed out reruns it by itself
    first_pass = {c.mcmc_outputfile: written[c.mcmc_outputfile]["chain"].copy() for c in cfgs}
    monkeypatch.setenv("GPEMU_CLOSURE_CHAIN_GIB", "1e-9")
    assert [len(b) for b in mcmc._closure_sub_batches(cfgs[0], [0, 1, 2], d)] == [1, 1, 1]
    mcmc.run_mcmc(cfgs[0], closure_index=0)
    assert all(not np.array_equal(first_pass[c.mcmc_outputfile], written[c.mcmc_outputfile]["chain"]) for c in cfgs)
    mcmc.run_mcmc(cfgs[1], closure_index=1)
    mcmc.run_mcmc(cfgs[2], closure_index=2)                      # handed out ...
    snap = written[cfgs[2].mcmc_outputfile]["chain"].copy()
    mcmc.run_mcmc(cfgs[2], closure_index=2)                      # ... asked again: runs alone
    assert not np.array_equal(snap, written[cfgs[2].mcmc_outputfile]["chain"])
    assert written[cfgs[2].mcmc_outputfile]["chain"].shape == (steps, W, d)
    mcmc._closure_done.clear()


def test_two_process_launch_one_writer_per_file(tmp_path):
    """``python -m torch.distributed.run --nproc-per-node 2`` over a script that, like the reference's steering
    script, never initialises a process group: the drop-in modules join the launcher's group themselves, the
    emulator groups and the closure chains are dealt to the ranks, the production chain shards its walkers, and every
    output file has exactly one writer.  The production chain equals the one a single process makes from rank 0's
    seeds."""
    import os
    import subprocess
    import sys
    from gpemu import h5io
    here = os.path.dirname(os.path.abspath(__file__))
    port = 29800 + (os.getpid() % 1500)
    path, analysis = DU.write_config(tmp_path, n_pc=5, n_restarts=0)
    # two ranks share the one GPU of the box, which RCCL refuses ("duplicate GPU"): the group the drop-in joins is gloo
    # here (GPEMU_DIST_BACKEND), the sampler's exchange the peer stores / torch.distributed path as in a real launch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPEMU_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "dist_dropin_worker.py"), str(tmp_path)]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    writes = [open(tmp_path / f"writes_rank{r}.txt").read().split() for r in (0, 1)]
    everything = writes[0] + writes[1]
    assert len(everything) == len(set(everything)), f"a file was written twice: {writes}"
    names = sorted(os.path.relpath(p, tmp_path / "out") for p in everything)
    # one emulator pickle; mcmc.h5 + sampler pickle of the production chain and of both closure chains
    assert sum(n.endswith(".pkl") and "emulation" in n for n in names) == 1
    assert sum(n.endswith("mcmc.h5") for n in names) == 3 and sum(n.endswith("mcmc_sampler.pkl") for n in names) == 3
    production = [p for p in writes[0] if p.endswith("mcmc.h5") and "closure" not in p]
    assert len(production) == 1                                   # rank 0 writes the sharded production chain
    assert any("closure" in p for p in writes[1])                 # rank 1 owns closure chain 1
    back = h5io.read_dict_from_h5(os.path.dirname(production[0]), "mcmc.h5")
    assert np.all(np.isfinite(back["log_prob"])) and back["chain"].ndim == 3
    # the stored log-probabilities are the log-posterior of the stored positions (the ranks held identical ensembles)
    from bayesian_inference import emulation, log_posterior
    g = GU.load("g1_rbf_noise")
    DU.install_fake_data_IO(g["Y"], g["design"], g["y_exp"], g["y_err"], {})
    ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", path, analysis)
    res = {"main": emulation.read_emulators(ec.emulation_groups_config["main"])}
    emu_cfg = _EmuCfg(ec.emulation_groups_config, DU.TrivialSort("main"))
    log_posterior.initialize_pool_variables(g["lo"], g["hi"], emu_cfg, res, {"y": g["y_exp"], "y_err": g["y_err"]}, None)
    lp = np.array([log_posterior.log_posterior(x)[0] for x in back["chain"][-1][:6]])
    np.testing.assert_allclose(lp, back["log_prob"][-1][:6], rtol=1e-10)
