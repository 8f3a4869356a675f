#!/usr/bin/env python3
"""The analysis at the SHIPPED size and MCMC settings (ref: config/jet_substructure.yaml:98-101: 100 walkers, 1000 burn-in +
5000 sampling steps, acceptance fraction logged every 10 steps) through the drop-in modules: synthetic 150 x 215 observables, one
emulation group of 11 PCs, emulation.fit_emulators, mcmc.run_mcmc, read-back.  Prints the wall time of each stage.
    python tools/run_dropin_shipped.py [n_restarts] [n_burn] [n_steps] [n_walkers] [n_logging_steps]"""
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

n_restarts = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n_burn = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
n_walk = int(sys.argv[4]) if len(sys.argv) > 4 else 100
n_log = int(sys.argv[5]) if len(sys.argv) > 5 else 10
prob = synthetic.make_problem(150, 215, seed=0)
tmp = Path(tempfile.mkdtemp(prefix="gpemu_c3_"))
DU.install_fake_data_IO(prob["Y"], prob["design"], prob["y_exp"], prob["y_err"], {})
cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "fixtures", "analysis.yaml")))
cfg["output_dir"] = str(tmp / "out")
ana = cfg["test_analysis"]
ana["parameterization"]["exponential"]["min"] = [float(v) for v in prob["lo"]]
ana["parameterization"]["exponential"]["max"] = [float(v) for v in prob["hi"]]
ana["parameters"]["emulators"]["main"]["n_pc"] = 11
ana["parameters"]["emulators"]["main"]["GPR"]["n_restarts"] = n_restarts
ana["parameters"]["mcmc"].update(n_walkers=n_walk, n_burn_steps=n_burn, n_sampling_steps=n_steps, n_logging_steps=n_log)
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
if os.environ.get("GPEMU_PROFILE"):          # host-side profile of run_mcmc (by internal time)
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    mcmc.run_mcmc(mc)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
else:
    mcmc.run_mcmc(mc)
t_mcmc = time.perf_counter() - t0
t0 = time.perf_counter()
back = h5io.read_dict_from_h5(mc.mcmc_output_dir, mc.mcmc_outputfilename)
sampler = pickle.load(open(mc.sampler_outputfile, "rb"))
t_read = time.perf_counter() - t0
evals = n_walk * (n_burn + n_steps)
print(f"fit_emulators (11 GPs, {n_restarts} restarts): {t_fit:.2f} s")
print(f"run_mcmc ({n_walk} walkers, logging every {n_log} steps, {n_burn} burn-in + {n_steps} steps = {evals} evaluations, incl. autocorrelation time, "
      f"mcmc.h5 and sampler pickle): {t_mcmc:.2f} s  ->  {evals / t_mcmc / 1e6:.2f} M evaluations/s end to end")
print(f"read back: chain {back['chain'].shape}, log_prob finite: {bool(np.all(np.isfinite(back['log_prob'])))}, "
      f"acceptance {float(np.mean(back['acceptance_fraction'])):.3f}, autocorrelation_time "
      f"{'None' if isinstance(back['autocorrelation_time'], dict) else np.round(back['autocorrelation_time'], 1)}; "
      f"pickle chain equal: {bool(np.array_equal(sampler.get_chain(), back['chain']))}  ({t_read:.2f} s)")
