#!/usr/bin/env python3
"""Latency of the drop-in `log_posterior.log_posterior(X)` called from the host (what a host-side sampler pays per
half-step) against the bare device call, at C3.   python tools/time_dropin_logpost.py [B ...]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import yaml  # noqa: E402

import dropin_util as DU  # noqa: E402
from gpemu import synthetic  # noqa: E402

prob = synthetic.make_problem(1000, 500, seed=0)
tmp = Path(tempfile.mkdtemp(prefix="gpemu_c3_"))
DU.install_fake_data_IO(prob["Y"], prob["design"], prob["y_exp"], prob["y_err"], {})
cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "fixtures", "analysis.yaml")))
cfg["output_dir"] = str(tmp / "out")
ana = cfg["test_analysis"]
ana["parameterization"]["exponential"]["min"] = [float(v) for v in prob["lo"]]
ana["parameterization"]["exponential"]["max"] = [float(v) for v in prob["hi"]]
ana["parameters"]["emulators"]["main"]["n_pc"] = 10
ana["parameters"]["emulators"]["main"]["GPR"]["n_restarts"] = 0
path = tmp / "analysis.yaml"
yaml.safe_dump(cfg, open(path, "w"))
from bayesian_inference import emulation, log_posterior  # noqa: E402

ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", str(path), ana)
np.random.seed(7)
emulation.fit_emulators(ec)
emulation.EmulationConfig.sort_observables_in_matrix = property(lambda self: DU.TrivialSort("main"))
emulation.EmulationConfig.observable_filter = property(lambda self: None)
results = ec.read_all_emulator_groups()
cov = emulation.compute_emulator_cov_unexplained(ec, results)
data = {"y": np.asarray(prob["y_exp"], float), "y_err": np.asarray(prob["y_err"], float)}
lo, hi = np.asarray(prob["lo"], float), np.asarray(prob["hi"], float)
log_posterior.initialize_pool_variables(lo, hi, ec, results, data, cov)
for B in [int(a) for a in (sys.argv[1:] or ["1", "64", "512"])]:
    X = synthetic.make_walkers(B, seed=1, lo=lo, hi=hi)
    log_posterior.log_posterior(X)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        lp = log_posterior.log_posterior(X)
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:5d}: {dt * 1e6:8.1f} us per log_posterior(X) call  ({B / dt / 1e6:.3f} M evaluations/s)  finite: {bool(np.all(np.isfinite(lp)))}")
X = synthetic.make_walkers(512, seed=1, lo=lo, hi=hi)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    log_posterior.log_posterior(X)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
