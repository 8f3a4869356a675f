#!/usr/bin/env python3
"""The whole analysis at C3 size THROUGH THE DROP-IN MODULES (not the bench's direct sampler): synthetic 1000 x 500
observables, emulation.fit_emulators (10 PCs, n_restarts as given), mcmc.run_mcmc with 1024 walkers, then the outputs a
reference user reads back (mcmc.h5, sampler pickle).  Prints the wall time of each stage.
    python tools/run_dropin_c3.py [n_restarts] [n_burn] [n_steps]"""
import os
import pickle
import sys
import tempfile
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import yaml  # noqa: E402

import dropin_util as DU  # noqa: E402
from gpemu import h5io, synthetic  # noqa: E402

n_restarts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_burn = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
prob = synthetic.make_problem(1000, 500, seed=0)
tmp = Path(tempfile.mkdtemp(prefix="gpemu_c3_"))
DU.install_fake_data_IO(prob["Y"], prob["design"], prob["y_exp"], prob["y_err"], {})
cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "fixtures", "analysis.yaml")))
cfg["output_dir"] = str(tmp / "out")
ana = cfg["test_analysis"]
ana["parameterization"]["exponential"]["min"] = [float(v) for v in prob["lo"]]
ana["parameterization"]["exponential"]["max"] = [float(v) for v in prob["hi"]]
ana["parameters"]["emulators"]["main"]["n_pc"] = 10
ana["parameters"]["emulators"]["main"]["GPR"]["n_restarts"] = n_restarts
ana["parameters"]["mcmc"].update(n_walkers=1024, n_burn_steps=n_burn, n_sampling_steps=n_steps, n_logging_steps=500)
path = tmp / "analysis.yaml"
yaml.safe_dump(cfg, open(path, "w"))

from bayesian_inference import emulation, mcmc  # noqa: E402

ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", str(path), ana)
np.random.seed(7)
t0 = time.perf_counter()
emulation.fit_emulators(ec)
t_fit = time.perf_counter() - t0
emulation.EmulationConfig.sort_observables_in_matrix = property(lambda self: DU.TrivialSort("main"))
emulation.EmulationConfig.observable_filter = property(lambda self: None)
mc = mcmc.MCMCConfig("test_analysis", "exponential", ana, str(path))
t0 = time.perf_counter()
mcmc.run_mcmc(mc)
t_mcmc = time.perf_counter() - t0
t0 = time.perf_counter()
back = h5io.read_dict_from_h5(mc.mcmc_output_dir, mc.mcmc_outputfilename)
sampler = pickle.load(open(mc.sampler_outputfile, "rb"))
t_read = time.perf_counter() - t0
evals = 1024 * (n_burn + n_steps)
print(f"fit_emulators (10 GPs, {n_restarts} restarts): {t_fit:.2f} s")
print(f"run_mcmc (1024 walkers, {n_burn} burn-in + {n_steps} steps = {evals} evaluations, incl. autocorrelation time, "
      f"mcmc.h5 and sampler pickle): {t_mcmc:.2f} s  ->  {evals / t_mcmc / 1e6:.2f} M evaluations/s end to end")
print(f"read back: chain {back['chain'].shape}, log_prob finite: {bool(np.all(np.isfinite(back['log_prob'])))}, "
      f"acceptance {float(np.mean(back['acceptance_fraction'])):.3f}, autocorrelation_time "
      f"{'None' if isinstance(back['autocorrelation_time'], dict) else np.round(back['autocorrelation_time'], 1)}; "
      f"pickle chain equal: {bool(np.array_equal(sampler.get_chain(), back['chain']))}  ({t_read:.2f} s)")
