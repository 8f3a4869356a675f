#!/usr/bin/env python3
"""cProfile of mcmc.run_mcmc at C3 size through the drop-in modules (second stage of tools/run_dropin_c3.py).
Original description: The whole analysis at C3 size THROUGH THE DROP-IN MODULES (not the bench's direct sampler): synthetic 1000 x 500
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
import cProfile
import pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
mcmc.run_mcmc(mc)
pr.disable()
print(f"run_mcmc: {time.perf_counter() - t0:.2f} s for {1024 * (n_burn + n_steps)} evaluations")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
