#!/usr/bin/env python3
"""cProfile of emulation.fit_emulators at C3 size through the drop-in modules (the first stage of tools/run_dropin_c3.py):
where the ~1 s beside the GP fit itself goes.   python tools/profile_dropin_fit.py [n_restarts]"""
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

n_restarts = int(sys.argv[1]) if len(sys.argv) > 1 else 50
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
path = tmp / "analysis.yaml"
yaml.safe_dump(cfg, open(path, "w"))
from bayesian_inference import emulation  # noqa: E402

ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", str(path), ana)
np.random.seed(7)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
emulation.fit_emulators(ec)
pr.disable()
print(f"fit_emulators: {time.perf_counter() - t0:.2f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
