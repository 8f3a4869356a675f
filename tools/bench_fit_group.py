#!/usr/bin/env python3
"""Wall time of the GP-fitting stage of fit_emulator_group (ref: emulation.py:164-177) on synthetic data:
k GPs x (1 + n_restarts) L-BFGS-B optimisations of the log-marginal likelihood, sequential (1 stream) vs
concurrent (8 host threads / HIP streams).  usage: bench_fit_group.py [N] [F] [k] [n_restarts]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayesian-inference_amd"))
import numpy as np  # noqa: E402

from gpemu import estimators as E  # noqa: E402
from gpemu import synthetic  # noqa: E402

N, F, k, nr = [int(a) for a in (sys.argv[1:5] + ["1000", "500", "10", "2"][len(sys.argv) - 1:])]
prob = synthetic.make_problem(N, F, seed=0)
t0 = time.perf_counter()
scaler, pca, Y_pca = E.scale_and_pca(prob["Y"])
t_pca = time.perf_counter() - t0
ls0 = prob["hi"] - prob["lo"]
kern = E.ARDKernel(kind=0, length_scale=ls0, length_scale_bounds=np.outer(ls0, (0.01, 100.0)), noise_level=0.1,
                   noise_level_bounds=(1e-3, 1e1))
print(f"N={N} F={F} k={k} n_restarts={nr}: scaler+PCA {t_pca * 1e3:.1f} ms", flush=True)
for streams in (1, 4, 8, 16):
    np.random.seed(7)
    t0 = time.perf_counter()
    gps = E.fit_gps(prob["design"], Y_pca[:, :k], kern, alpha=1e-10, n_restarts_optimizer=nr, n_streams=streams)
    dt = time.perf_counter() - t0
    print(f"  streams={streams:2d}: {dt:7.3f} s   sum lml = {sum(g.log_marginal_likelihood_value_ for g in gps):.6f}",
          flush=True)
