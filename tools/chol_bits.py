#!/usr/bin/env python3
"""SHA-256 of the device Cholesky factor / LML + gradient of fixed problems: two library builds that print the same
digests compute the same bits (used to check that a rescheduling of the diagonal-block sweep changed no arithmetic)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayesian-inference_amd"))
import numpy as np  # noqa: E402

from gpemu import synthetic  # noqa: E402
from gpemu.fit import DeviceFit, cholesky  # noqa: E402

rng = np.random.default_rng(7)
for n in (64, 200, 1000, 2300):
    A = rng.normal(size=(n, n))
    A = A @ A.T + n * np.eye(n)
    L = cholesky(A)
    print(f"cholesky n={n}: {hashlib.sha256(np.ascontiguousarray(L).tobytes()).hexdigest()[:24]}")
for N in (1000, 5000):
    prob = synthetic.make_problem(N, 40, seed=3)
    theta = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.5, 0.05])
    fit = DeviceFit(prob["design"], kernel_kind=0, has_noise=True, jitter=1e-10)
    y = prob["Y"][:, 0] - prob["Y"][:, 0].mean()
    lml, g = fit.lml(y, theta, eval_gradient=True)
    print(f"lml N={N}: {lml!r} grad digest {hashlib.sha256(np.asarray(g).tobytes()).hexdigest()[:24]}")
    fit.close()
