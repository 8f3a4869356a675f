#!/usr/bin/env python3
"""The emulator fit of the shipped three-group configuration (tests/golden g7_shipped_config: ~150 design points, groups of
5 / 11 / 25 PCs, Matern-1.5 + White) with the shipped n_restarts = 50: seconds per group and in all.
   python tools/time_shipped_fit.py [n_restarts]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import golden_util as GU  # noqa: E402
from gpemu import estimators as E  # noqa: E402

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = GU.load("g7_shipped_config")
names = [str(n) for n in g["group_names"]]
lo, hi = g["lo"], g["hi"]
for rep in ("cold", "warm"):
    np.random.seed(20260307)
    total = 0.0
    parts = []
    for n in names:
        k = int(g[n + "_n_pc"])
        t0 = time.perf_counter()
        scaler, pca, scores = E.scale_and_pca(g[n + "_Y"])
        t1 = time.perf_counter()
        ls = hi - lo
        kern = E.ARDKernel(E.MATERN_KIND, length_scale=ls, length_scale_bounds=np.outer(ls, (0.01, 100)), nu=1.5,
                           noise_level=0.25, noise_level_bounds=(0.0001, 1))
        emus = E.fit_gps(g["design"], scores[:, :k], kern, alpha=float(g["gpr_alpha"]), n_restarts_optimizer=nr)
        t2 = time.perf_counter()
        parts.append(f"{n}: N = {g['design'].shape[0]}, {k} GPs, PCA {1e3 * (t1 - t0):.1f} ms, fit {t2 - t1:.2f} s "
                     f"({emus[0].n_lml_evaluations_} evaluations, {1e6 * (t2 - t1) / max(emus[0].n_lml_evaluations_, 1):.1f} us each)")
        total += t2 - t0
    print(rep, f"{total:.2f} s in all;", " | ".join(parts), flush=True)
