#!/usr/bin/env python3
"""Golden G7: the reference run END TO END on its own fixture in the SHAPE of its shipped analysis.

Run in the build container only (needs /root/reference and scikit-learn):

    python tests/golden/make_g7_shipped.py          # writes tests/golden/g7_shipped_config.npz

The shipped production analysis (ref: config/jet_substructure.yaml:243-278) emulates THREE groups of observables with
5 / 11 / 25 principal components, Matern-1.5 + White kernel, alpha 1e-10, merged by the real
``SortEmulationGroupObservables``.  Its input tables are not in the repository; the only data the reference ships is
``tests/test_data/observables.h5`` (200 design points x 16 hadron-spectrum observables, 215 bins).  So this script
writes a small YAML in the reference's schema that splits THOSE 16 observables into three groups by observable class
(charged hadrons / charged pions / neutral pions -- they interleave in the sorted observable order) with the shipped
n_pc, kernel and alpha, and then calls, UNCHANGED from /root/reference/src:

    emulation.EmulationConfig.from_config_file  -> data_IO.ObservableFilter per group          (emulation.py:551-709)
    emulation.fit_emulators                     -> data_IO.predictions_matrix_from_h5 / design_array_from_h5 on the real
                                                   observables.h5, StandardScaler, PCA, GaussianProcessRegressor.fit,
                                                   pickles                                      (emulation.py:38-211)
    emulation_config.read_all_emulator_groups, compute_emulator_cov_unexplained                 (emulation.py:196-224)
    emulation.predict (merge_predictions_over_groups=True): learn_mapping + convert             (emulation.py:289-462)
    data_IO.data_array_from_h5 (experimental data of all three groups' observables)             (data_IO.py:345-388)
    log_posterior.initialize_pool_variables / log_posterior (per walker, batched, mixed)        (log_posterior.py:26-146)

i.e. the call sequence of ref: steer_analysis.py:141-162 + mcmc.py:47-67 up to the sampler (emcee is not installed).
silx is not installed: the two names data_IO imports from it are served by gpemu.h5io, whose reader is pinned against
h5py on this very file (tests/test_h5io.py).  n_restarts is 2 instead of the shipped 50 (a restart count changes which
optimum is kept, not the arithmetic; the goldens hold the fitted theta, and parity is checked at identical theta).

Only arrays are stored (inputs + expected outputs); no reference code, no pickled reference objects.
"""
from __future__ import annotations

import os
import shutil
import sys
import tempfile
import warnings

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bayesian-inference_amd"))
from gpemu import h5io, synthetic  # noqa: E402

h5io.install_silx_shim()
sys.path.insert(0, "/root/reference/src")
for name in [m for m in sys.modules if m.startswith("bayesian_inference")]:
    del sys.modules[name]
from bayesian_inference import data_IO, emulation, log_posterior  # noqa: E402  (the reference)

assert emulation.__file__.startswith("/root/reference/"), emulation.__file__
warnings.filterwarnings("ignore")

ANALYSIS, PARAM = "shipped_shape", "exponential"
# group name -> (n_pc, include_list): the shipped n_pc values on the fixture's three observable classes
GROUPS = {
    "pi0_group": (5, ["hadron__pt_pi0_"]),          # 2 observables, 22 bins
    "pion_group": (11, ["hadron__pt_pi_"]),         # 4 observables, 60 bins
    "charged_group": (25, ["hadron__pt_ch_"]),      # 10 observables, 133 bins
}


def analysis_yaml(output_dir):
    """The reference's config schema (ref: config/jet_substructure.yaml:52-81, 243-278), shipped emulator parameters."""
    emu_defaults = {
        "force_retrain": True,
        "kernels": {
            "active": ["matern", "noise"],
            "matern": {"nu": 1.5, "length_scale_bounds_factor": [0.01, 100]},
            "rbf": {"length_scale_bounds_factor": [0.01, 100]},
            "constant": {"constant_value": 1.0, "constant_value_bounds": [0.001, 10]},
            "noise": {"type": "white", "args": {"noise_level": 0.25, "noise_level_bounds": [0.0001, 1]}},
        },
        "GPR": {"n_restarts": 2, "alpha": 1.0e-10},
    }
    emulators = {}
    for g, (n_pc, include) in GROUPS.items():
        emulators[g] = dict(emu_defaults, n_pc=n_pc, observable_list=list(include), observable_exclude_list=[])
    return {
        "observable_table_dir": "tables", "observable_config_dir": "configs",
        "observables_filename": "observables.h5", "output_dir": output_dir,
        "global_observable_exclude_list": [],
        ANALYSIS: {
            "parameterizations": [PARAM],
            "parameterization": {PARAM: {
                "names": ["alpha_s", "Q_0", "c_1", "c_2", "tau_0", "c_3"],
                "min": [float(v) for v in synthetic.BOX_LO], "max": [float(v) for v in synthetic.BOX_HI]}},
            "parameters": {
                "emulators": emulators,
                "mcmc": {"n_walkers": 100, "n_burn_steps": 1000, "n_sampling_steps": 50000, "n_logging_steps": 10},
            },
        },
    }


def main():
    tmp = tempfile.mkdtemp(prefix="g7_")
    try:
        cfg = analysis_yaml(os.path.join(tmp, "out"))
        # the box must hold the fixture's design (its c_1..c_3 columns are in log space, outside the shipped box)
        run_dir = os.path.join(tmp, "out", f"{ANALYSIS}_{PARAM}")
        os.makedirs(run_dir)
        shutil.copy("/root/reference/tests/test_data/observables.h5", os.path.join(run_dir, "observables.h5"))
        design = data_IO.design_array_from_h5(run_dir, "observables.h5")
        lo = np.minimum(synthetic.BOX_LO, design.min(0) - 1e-6)
        hi = np.maximum(synthetic.BOX_HI, design.max(0) + 1e-6)
        cfg[ANALYSIS]["parameterization"][PARAM]["min"] = [float(v) for v in lo]
        cfg[ANALYSIS]["parameterization"][PARAM]["max"] = [float(v) for v in hi]
        config_file = os.path.join(tmp, "analysis.yaml")
        with open(config_file, "w") as f:
            yaml.safe_dump(cfg, f)
        analysis_config = cfg[ANALYSIS]

        # ---- ref: steer_analysis.py:141-147 ----
        emulation_config = emulation.EmulationConfig.from_config_file(
            analysis_name=ANALYSIS, parameterization=PARAM, analysis_config=analysis_config, config_file=config_file)
        np.random.seed(20260307)            # the GPR restarts draw from numpy's global state (skl _gpr.py:327)
        emulation.fit_emulators(emulation_config)

        # ---- ref: mcmc.py:47-67 ----
        emulation_config = emulation.EmulationConfig.from_config_file(
            analysis_name=ANALYSIS, parameterization=PARAM, analysis_config=analysis_config, config_file=config_file)
        results = emulation_config.read_all_emulator_groups()
        cov_unexplained = emulation.compute_emulator_cov_unexplained(emulation_config, results)
        assert cov_unexplained is None      # the reference's wrapper has no return statement (emulation.py:214-224)
        experimental = data_IO.data_array_from_h5(run_dir, "observables.h5", pseudodata_index=-1,
                                                  observable_filter=emulation_config.observable_filter)

        out = dict(lo=lo, hi=hi, gpr_alpha=np.float64(1e-10), design=design,
                   group_names=np.array(list(GROUPS)), y_exp=experimental["y"], y_err=experimental["y_err"])
        # the mapping the real sorter learned from the real file
        sorter = emulation_config.sort_observables_in_matrix
        keys = list(sorter.emulation_group_to_observable_matrix)
        rows = [sorter.emulation_group_to_observable_matrix[k] for k in keys]
        out.update(map_observables=np.array(keys), map_group=np.array([r[0] for r in rows]),
                   map_out_start=np.array([r[1].start for r in rows], dtype=np.int64),
                   map_out_stop=np.array([r[1].stop for r in rows], dtype=np.int64),
                   map_grp_start=np.array([r[2].start for r in rows], dtype=np.int64),
                   map_grp_stop=np.array([r[2].stop for r in rows], dtype=np.int64),
                   map_shape=np.array(sorter.shape, dtype=np.int64))
        for g, gcfg in emulation_config.emulation_groups_config.items():
            res = results[g]
            pca, scaler, emus = res["PCA"]["pca"], res["PCA"]["scaler"], res["emulators"]
            assert len(emus) == GROUPS[g][0]
            Y = res["PCA"]["Y"]
            pre = g + "_"
            out.update({
                pre + "Y": Y, pre + "n_pc": np.int64(gcfg.n_pc),
                pre + "kernel_kind": np.int64(1), pre + "nu": np.float64(1.5),
                pre + "has_const": np.int64(0), pre + "has_noise": np.int64(1),
                pre + "scaler_mean": scaler.mean_, pre + "scaler_scale": scaler.scale_, pre + "scaler_var": scaler.var_,
                pre + "pca_components": pca.components_, pre + "pca_explained_variance": pca.explained_variance_,
                pre + "pca_explained_variance_ratio": pca.explained_variance_ratio_,
                pre + "flip_argmax": np.argmax(np.abs(pca.components_), axis=1).astype(np.int64),
                pre + "Y_pca_truncated": np.ascontiguousarray(res["PCA"]["Y_pca_truncated"]),
                pre + "theta": np.stack([e.kernel_.theta for e in emus]),
                pre + "alpha": np.stack([e.alpha_ for e in emus]),
                pre + "lml_value": np.array([e.log_marginal_likelihood_value_ for e in emus]),
                pre + "L_index": np.array([0], dtype=np.int64), pre + "L": np.stack([emus[0].L_]),
                pre + "L_checksum": np.array([[e.L_.sum(), (e.L_ ** 2).sum(), np.abs(e.L_).max()] for e in emus]),
                pre + "cov_unexplained": emulation.compute_emulator_group_cov_unexplained(gcfg, res),
            })
            lml, grad = zip(*[e.log_marginal_likelihood(e.kernel_.theta, eval_gradient=True) for e in emus])
            out.update({pre + "lml_at_theta": np.array(lml), pre + "grad_at_theta": np.stack(grad)})

        # ---- merged predict (ref: emulation.py:410-462) and the log-posterior (ref: log_posterior.py:42-101) ----
        Xq = synthetic.make_walkers(24, seed=1, lo=design.min(0), hi=design.max(0))
        merged = emulation.predict(Xq, emulation_config, emulation_group_results=results)
        merged1 = emulation.predict(Xq[:1], emulation_config, emulation_group_results=results)
        out.update(Xq=Xq, merged_central_value=merged["central_value"],
                   merged_cov_first=merged["cov"][0].copy(),
                   merged_cov_diag=np.stack([np.diag(c) for c in merged["cov"]]),
                   merged1_central_value=merged1["central_value"], merged1_cov=merged1["cov"][0])
        log_posterior.initialize_pool_variables(lo, hi, emulation_config, results, experimental, cov_unexplained)
        per_walker = np.array([log_posterior.log_posterior(Xq[i])[0] for i in range(Xq.shape[0])])
        batched = log_posterior.log_posterior(Xq)
        Xo = Xq[:8].copy()
        Xo[1, 0] = lo[0]
        Xo[3, 2] = hi[2] + 1.0
        Xo[6, 5] = lo[5] - 1e-9
        mixed = log_posterior.log_posterior(Xo)
        out.update(logpost_per_walker=per_walker, logpost_batched=batched, X_mixed=Xo, logpost_mixed=mixed)
        path = os.path.join(HERE, "g7_shipped_config.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB);",
              {g: (int(out[g + '_n_pc']), out[g + '_Y'].shape) for g in GROUPS},
              "merged F =", merged["central_value"].shape[1])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
