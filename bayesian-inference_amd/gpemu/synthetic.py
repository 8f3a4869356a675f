"""Synthetic design / observable matrices for parity tests and bench.py.

The generator is the one fixed in SURVEY.md section 8(d): the parameter box is the
one shipped in the reference's ``config/jet_substructure.yaml:130-131`` (d = 6), the
observable matrix is smooth and low rank plus noise so that a handful of principal
components carry >95 % of the variance.  The same function is used by the golden
generator (tests/golden/make_goldens.py), by the tests and by bench.py, so every
side sees identical inputs for a given (N, F, seed).
"""
from __future__ import annotations

import numpy as np

# ref: config/jet_substructure.yaml:130-131 (parameterization "exponential")
BOX_LO = np.array([0.1, 1.0, 0.0067, 0.0067, 0.0, 0.05])
BOX_HI = np.array([0.5, 10.0, 10.0, 10.0, 1.5, 100.0])


def make_problem(n_design: int, n_obs: int, seed: int = 0):
    """Return dict(design (N,d), Y (N,F), lo, hi, y_exp (F,), y_err (F,))."""
    rng = np.random.default_rng(seed)
    lo, hi = BOX_LO.copy(), BOX_HI.copy()
    d = lo.size
    design = rng.uniform(lo, hi, (n_design, d))
    Wm = rng.normal(size=(d, n_obs))
    noise = rng.normal(size=(n_design, n_obs))
    Y = np.tanh(((design - lo) / (hi - lo)) @ Wm) + 0.01 * noise
    y_err = rng.uniform(0.01, 0.2, n_obs)
    y_exp = Y[0] + 0.05
    return dict(design=design, Y=Y, lo=lo, hi=hi, y_exp=y_exp, y_err=y_err)


def make_walkers(n_walkers: int, seed: int = 1, lo=None, hi=None):
    """Uniform walker start positions inside the box (ref: mcmc.py:88)."""
    lo = BOX_LO if lo is None else np.asarray(lo, dtype=np.float64)
    hi = BOX_HI if hi is None else np.asarray(hi, dtype=np.float64)
    rng = np.random.default_rng(seed)
    return rng.uniform(lo, hi, (n_walkers, lo.size))
