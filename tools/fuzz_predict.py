#!/usr/bin/env python3
"""Randomised parity sweep of the predict / log-posterior path against the oracle (larger than the pytest sweep):
random N, d, F, k, B and kernel family; prints the worst relative errors.  usage: fuzz_predict.py [n_cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayesian-inference_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import golden_util as GU  # noqa: E402
import test_gpu_shapes as TS  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(mean=0.0, var=0.0, lp=0.0)
for case in range(n_cases):
    N = int(rng.integers(40, 1400))
    d = int(rng.integers(1, 9))
    k = int(rng.integers(1, 13))
    F = int(rng.integers(max(k, 8), 120))
    B = int(rng.integers(129, 1500))
    kind, nu = [(O.RBF, np.inf), (O.MATERN, 0.5), (O.MATERN, 1.5), (O.MATERN, 2.5)][int(rng.integers(0, 4))]
    model, lo, hi, y_exp, y_err, r2 = TS._problem(N, d, F, k, kind, nu, bool(rng.integers(0, 2)), True, seed=1000 + case)
    dm = GU.device_model(model)
    Xq = r2.uniform(lo, hi, (B, d))
    m, v = dm.gp_predict(Xq)
    sub = np.r_[0:3, B // 2, B - 3:B]
    mo, vo = O.gp_predict_all(Xq[sub], model)
    e_m = np.max(np.abs(m[sub] - mo)) / max(np.max(np.abs(mo)), 1e-300)
    e_v = np.max(np.abs(v[sub] - vo)) / max(1.0, np.max(vo))
    dm.likelihood_setup(y_exp, y_err, lo, hi, n_div=1.0)
    lp = dm.logpost(Xq)
    ref = np.array([O.log_posterior(Xq[i], {"g": model}, lo, hi, y_exp, y_err)[0] for i in sub[:4]])
    e_l = np.max(np.abs(lp[sub[:4]] - ref) / np.maximum(np.abs(ref), 1e-300))
    worst = dict(mean=max(worst["mean"], e_m), var=max(worst["var"], e_v), lp=max(worst["lp"], e_l))
    print(f"case {case:3d}: N={N:5d} d={d} F={F:4d} k={model.n_pc:2d} B={B:5d} kind={kind} nu={nu}: "
          f"mean {e_m:.1e} var {e_v:.1e} logpost {e_l:.1e}", flush=True)
    dm.close()
print("worst:", worst)
assert max(worst.values()) < 1e-8, worst
