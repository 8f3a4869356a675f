"""Worker of tests/test_integration_rehearsal.py (build container only: needs /root/reference).

    python tests/integration_rehearsal_worker.py dropin    <workdir>     # PYTHONPATH = <repo pkg>:<reference src>
    python tests/integration_rehearsal_worker.py reference <workdir>     # PYTHONPATH = <reference src>

``dropin``: the reference's REAL ``steer_analysis.SteerAnalysis(config).run_analysis()`` -- unchanged, imported from
/root/reference/src -- with this repository's package in front of it, so that ``emulation`` / ``log_posterior`` / ``mcmc``
resolve to the drop-in modules and ``data_IO``, ``helpers``, ``common_base``, ``steer_analysis`` to the untouched
reference.  There is no GPU here, so ``libgpemu.so`` is replaced by the oracle-backed TEST DOUBLE of
tests/fake_gpemu_lib.py: every line of host glue above the C ABI runs as it does in production.  Plot stages are off;
seaborn / statsmodels / pymc / emcee (imported by plot modules at import time, absent here) are never-called
placeholders.

``reference``: the reference's own ``fit_emulators`` + ``predict`` + ``log_posterior`` on the same config file and the
same numpy seed (what produced golden G7), for the comparison.

Both write ``<workdir>/<mode>.npz``.
"""
import os
import shutil
import sys
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
ANALYSIS, PARAM = "rehearsal", "exponential"
GROUPS = {"pi0_group": (5, ["hadron__pt_pi0_"]), "pion_group": (11, ["hadron__pt_pi_"]),
          "charged_group": (25, ["hadron__pt_ch_"])}
SEED = 20260308


def write_config(workdir, lo, hi):
    emu = {
        "force_retrain": True,
        "kernels": {"active": ["matern", "noise"],
                    "matern": {"nu": 1.5, "length_scale_bounds_factor": [0.01, 100]},
                    "noise": {"type": "white", "args": {"noise_level": 0.25, "noise_level_bounds": [0.0001, 1]}}},
        "GPR": {"n_restarts": 1, "alpha": 1.0e-10},
    }
    analysis = {
        "parameterizations": [PARAM],
        "parameterization": {PARAM: {"names": ["alpha_s", "Q_0", "c_1", "c_2", "tau_0", "c_3"],
                                     "min": [float(v) for v in lo], "max": [float(v) for v in hi]}},
        "validation_indices": [0, 2],
        "parameters": {
            "emulators": {g: dict(emu, n_pc=n, observable_list=list(inc), observable_exclude_list=[])
                          for g, (n, inc) in GROUPS.items()},
            "mcmc": {"n_walkers": 24, "n_burn_steps": 8, "n_sampling_steps": 12, "n_logging_steps": 5},
        },
    }
    cfg = {
        "output_dir": os.path.join(workdir, "out"),
        "initialize_observables": False, "preprocess_input_data": False,
        "fit_emulators": True, "run_mcmc": True, "run_closure_tests": True,
        "plot": {k: False for k in ("input_data", "emulators", "mcmc", "qhat", "closure_tests", "across_analyses")},
        "observable_table_dir": "tables", "observable_config_dir": "configs", "observables_filename": "observables.h5",
        "global_observable_exclude_list": [],
        "analyses": {ANALYSIS: analysis},
    }
    path = os.path.join(workdir, "steer.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f, sort_keys=False)
    with open(path) as f:                # what every consumer sees: the file (group order included)
        cfg = yaml.safe_load(f)
    return path, cfg


def placeholder(name):
    """A module that imports but must never be used (plot-only dependencies)."""
    mod = types.ModuleType(name)

    def __getattr__(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)

        class Never:                                  # can be subclassed at import time, never instantiated or called
            def __init__(self, *a, **k):
                raise RuntimeError(f"{name}.{attr} is a placeholder of the integration rehearsal: plotting is off")
        Never.__name__ = attr
        return Never
    mod.__getattr__ = __getattr__
    sys.modules[name] = mod
    return mod


def box(design):
    sys.path.insert(0, os.path.join(REPO, "bayesian-inference_amd"))
    from gpemu import synthetic
    return (np.minimum(synthetic.BOX_LO, design.min(0) - 1e-6), np.maximum(synthetic.BOX_HI, design.max(0) + 1e-6))


def prepare(workdir):
    run_dir = os.path.join(workdir, "out", f"{ANALYSIS}_{PARAM}")
    os.makedirs(run_dir, exist_ok=True)
    shutil.copy(os.path.join(REF, "tests", "test_data", "observables.h5"), os.path.join(run_dir, "observables.h5"))
    return run_dir


def query_points(design, n=10):
    rng = np.random.default_rng(4)
    return rng.uniform(design.min(0), design.max(0), (n, design.shape[1]))


def collect(emulation, log_posterior, data_IO, config_file, cfg, run_dir, lo, hi):
    """What both sides are compared on, through the modules' public functions."""
    analysis_config = cfg["analyses"][ANALYSIS]
    ec = emulation.EmulationConfig.from_config_file(analysis_name=ANALYSIS, parameterization=PARAM,
                                                    analysis_config=analysis_config, config_file=config_file)
    results = ec.read_all_emulator_groups()
    design = data_IO.design_array_from_h5(run_dir, "observables.h5")
    Xq = query_points(design)
    merged = emulation.predict(Xq, ec, emulation_group_results=results)
    data = data_IO.data_array_from_h5(run_dir, "observables.h5", pseudodata_index=-1, observable_filter=ec.observable_filter)
    log_posterior.initialize_pool_variables(lo, hi, ec, results, data, None)
    per_walker = np.array([log_posterior.log_posterior(Xq[i])[0] for i in range(Xq.shape[0])])
    batched = log_posterior.log_posterior(Xq)
    out = dict(Xq=Xq, central_value=merged["central_value"], cov=merged["cov"], logpost_per_walker=per_walker,
               logpost_batched=batched)
    for g in GROUPS:
        res = results[g]
        out[g + "_theta"] = np.stack([e.kernel_.theta for e in res["emulators"]])
        out[g + "_lml"] = np.array([e.log_marginal_likelihood_value_ for e in res["emulators"]])
        out[g + "_explained_variance"] = np.asarray(res["PCA"]["pca"].explained_variance_)
        out[g + "_Y_pca_truncated"] = np.asarray(res["PCA"]["Y_pca_truncated"])
        out[g + "_scaler_mean"] = np.asarray(res["PCA"]["scaler"].mean_)
    return out


def main(mode, workdir):
    import warnings
    warnings.filterwarnings("ignore")
    run_dir = prepare(workdir)
    if mode == "dropin":
        for name in ("seaborn", "statsmodels", "statsmodels.api", "pymc", "emcee"):
            placeholder(name)
        # the plot modules set a seaborn style when they are imported (ref: plot_mcmc.py:20 etc.): a no-op here
        sys.modules["seaborn"].set_context = lambda *a, **k: None
        sys.path.insert(0, HERE)
        sys.path.insert(0, REPO)
        import fake_gpemu_lib
        fake = fake_gpemu_lib.install()
        from bayesian_inference import data_IO, emulation, log_posterior, mcmc, steer_analysis
        assert "bayesian-inference_amd" in emulation.__file__ and "bayesian-inference_amd" in mcmc.__file__
        assert data_IO.__file__.startswith(REF) and steer_analysis.__file__.startswith(REF)
        design = data_IO.design_array_from_h5(run_dir, "observables.h5")
        lo, hi = box(design)
        config_file, cfg = write_config(workdir, lo, hi)
        np.random.seed(SEED)
        # ---- the reference's steering script, unchanged (ref: steer_analysis.py:66-183) ----
        steer_analysis.SteerAnalysis(config_file=config_file).run_analysis()
        out = collect(emulation, log_posterior, data_IO, config_file, cfg, run_dir, lo, hi)
        # what the steering script left on disk, read back through the reference's data_IO
        prod = data_IO.read_dict_from_h5(run_dir, "mcmc.h5")
        out["mcmc_chain"] = prod["chain"]
        out["mcmc_log_prob"] = prod["log_prob"]
        out["mcmc_acceptance_fraction"] = prod["acceptance_fraction"]
        for j in range(2):
            cl = data_IO.read_dict_from_h5(os.path.join(run_dir, "closure", "results", str(j)), "mcmc.h5")
            out[f"closure{j}_chain"] = cl["chain"]
            out[f"closure{j}_design_point"] = cl["design_point"]
            out[f"closure{j}_pseudodata_y"] = cl["experimental_pseudodata"]["y"]
        out["files"] = np.array(sorted(os.path.relpath(os.path.join(r, f), run_dir)
                                       for r, _d, fs in os.walk(run_dir) for f in fs))
        out["lib_calls"] = np.array([f"{k}={v}" for k, v in sorted(fake.calls.items())])
    else:
        sys.path.insert(0, os.path.join(REPO, "bayesian-inference_amd"))
        from gpemu import h5io
        h5io.install_silx_shim()
        sys.path.remove(os.path.join(REPO, "bayesian-inference_amd"))
        for name in [m for m in sys.modules if m.startswith("bayesian_inference")]:
            del sys.modules[name]
        sys.path.insert(0, os.path.join(REF, "src"))
        from bayesian_inference import data_IO, emulation, log_posterior
        assert emulation.__file__.startswith(REF), emulation.__file__
        design = data_IO.design_array_from_h5(run_dir, "observables.h5")
        lo, hi = box(design)
        config_file, cfg = write_config(workdir, lo, hi)
        ec = emulation.EmulationConfig.from_config_file(analysis_name=ANALYSIS, parameterization=PARAM,
                                                        analysis_config=cfg["analyses"][ANALYSIS], config_file=config_file)
        np.random.seed(SEED)
        emulation.fit_emulators(ec)                      # ref: steer_analysis.py:141-147
        out = collect(emulation, log_posterior, data_IO, config_file, cfg, run_dir, lo, hi)
        # The same at the DROP-IN run's hyper-parameters (L-BFGS-B on a multi-modal LML may stop in another optimum for a
        # few of the 41 GPs -- optimiser path, not arithmetic): the reference's own objective evaluated at those theta,
        # and the reference's own predict / log_posterior with its sklearn GPs re-fitted there (optimizer=None).
        theirs = np.load(os.path.join(os.path.dirname(workdir.rstrip("/")), "dropin", "dropin.npz"))
        import pickle
        import sklearn.gaussian_process as skg
        results = ec.read_all_emulator_groups()
        for g, gcfg in ec.emulation_groups_config.items():
            res = results[g]
            at = []
            refit = []
            for i, e in enumerate(res["emulators"]):
                th = theirs[g + "_theta"][i]
                at.append(e.log_marginal_likelihood(th))
                gp = skg.GaussianProcessRegressor(kernel=e.kernel_.clone_with_theta(th), alpha=gcfg.alpha, optimizer=None,
                                                  copy_X_train=False).fit(e.X_train_, e.y_train_)
                refit.append(gp)
            out[g + "_lml_at_dropin_theta"] = np.array(at)
            res["emulators"] = refit
            with open(gcfg.emulation_outputfile, "wb") as f:
                pickle.dump(res, f)
        again = collect(emulation, log_posterior, data_IO, config_file, cfg, run_dir, lo, hi)
        for key in ("central_value", "cov", "logpost_per_walker", "logpost_batched"):
            out["at_dropin_theta_" + key] = again[key]
    np.savez(os.path.join(workdir, mode + ".npz"), **out)
    print(mode, "done")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
