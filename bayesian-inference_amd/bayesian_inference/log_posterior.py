"""Drop-in replacement of ``bayesian_inference.log_posterior`` (ref: src/bayesian_inference/log_posterior.py).

``initialize_pool_variables`` keeps the reference's module-global calling convention; ``log_posterior(X)``
returns an ndarray of shape (n_samples,), ``-inf`` outside the open parameter box, evaluated by
libgpemu on the device.  The reference's batch semantics are kept: the truncation covariance is
divided by the number of IN-BOUNDS rows of the call (ref: log_posterior.py:67,80 ->
emulation.py:493,531-532), so one call with B rows differs from B single-row calls exactly as it
does upstream; emcee calls it with one walker at a time.  The ensemble sampler's fused device path
(``gpemu.sampler.DeviceSampler``) uses the single-walker semantics (n = 1).

A covariance that is not positive definite yields NaN (the reference's dpotrf error branch is dead
code, ref: log_posterior.py:125-135, and it would also compute with an invalid factor).
"""
from __future__ import annotations

import logging

import numpy as np

from bayesian_inference import emulation

logger = logging.getLogger(__name__)

min = None
max = None
emulation_config = None
emulation_results = None
experimental_results = None
emulator_cov_unexplained = None

_state = {"models": None, "n_div": None}


def initialize_pool_variables(local_min, local_max, local_emulation_config, local_emulation_results,
                              local_experimental_results, local_emulator_cov_unexplained) -> None:
    """Same signature as the reference (ref: log_posterior.py:26-38); also drops cached device state."""
    global min, max, emulation_config, emulation_results, experimental_results, emulator_cov_unexplained
    min = local_min
    max = local_max
    emulation_config = local_emulation_config
    emulation_results = local_emulation_results
    experimental_results = local_experimental_results
    emulator_cov_unexplained = local_emulator_cov_unexplained
    _state["models"] = None
    _state["n_div"] = None
    # the previous run's device models (k N^2 doubles each) are not needed any more
    emulation.release_device_models()


def _group_layouts():
    """[(group name, group config, columns in the merged observable order, observable block starts)]."""
    groups = list(emulation_config.emulation_groups_config.items())
    sorter = getattr(emulation_config, "sort_observables_in_matrix", None)
    out = []
    for name, cfg in groups:
        if hasattr(sorter, "group_layout"):
            cols, starts = sorter.group_layout(name)
        elif len(groups) == 1:     # single group whose matrix already is the merged matrix
            F = experimental_results['y'].shape[0]
            cols, starts = np.arange(F), np.array([0, F], dtype=np.int64)
        else:
            raise ValueError("multiple emulation groups need emulation_config.sort_observables_in_matrix "
                             "with a group_layout() (bayesian_inference.emulation.SortEmulationGroupObservables)")
        out.append((name, cfg, cols, starts))
    return out


def device_models(n_div: float = 1.0):
    """Device models of all groups with the likelihood set up for ``n_div`` (cached)."""
    if _state["models"] is None:
        results = emulation_results or emulation_config.read_all_emulator_groups()
        models = []
        for name, cfg, cols, starts in _group_layouts():
            cov_un = emulator_cov_unexplained[name] if emulator_cov_unexplained else None
            models.append((emulation.device_model_for(results[name], cfg.n_pc, cov_un), cols, starts))
        _state["models"] = models
        _state["n_div"] = None
    if _state["n_div"] != float(n_div):
        lo = np.asarray(min, dtype=np.float64)
        hi = np.asarray(max, dtype=np.float64)
        y, y_err = experimental_results['y'], experimental_results['y_err']
        for dm, cols, starts in _state["models"]:
            dm.likelihood_setup(y[cols], y_err[cols], lo, hi, n_div=float(n_div), block_start=starts)
        _state["n_div"] = float(n_div)
    return [m for m, _, _ in _state["models"]]


def chains_can_stack(config) -> bool:
    """Closure chains share one multi-chain device sampler for every configuration the device models support (any
    number of emulation groups, up to 64 PCs each: ``tests/test_gpu_shipped.py`` stacks the shipped three-group shape);
    what bounds a stacked run is chain memory, which ``mcmc._closure_sub_batches`` handles.  False only for a config
    without an ``emulators`` block (then there is nothing to run either way)."""
    try:
        groups = config.analysis_config['parameters']['emulators']
    except (KeyError, TypeError):
        return False
    return len(groups) >= 1


def device_models_for_chains(y_chains):
    """Device models (n_div = 1) with ONE DATA VECTOR PER CHAIN: ``y_chains`` (C, F) in the merged observable
    order; the uncertainties are those of ``experimental_results``."""
    y_chains = np.asarray(y_chains, dtype=np.float64)
    models = device_models(n_div=1.0)               # builds / caches the models
    lo = np.asarray(min, dtype=np.float64)
    hi = np.asarray(max, dtype=np.float64)
    y_err = experimental_results['y_err']
    for dm, cols, starts in _state["models"]:
        dm.likelihood_setup(y_chains[:, cols], y_err[cols], lo, hi, n_div=1.0, block_start=starts)
    _state["n_div"] = None                          # the single-vector constants are gone
    return models


def log_posterior(X):
    """log-posterior of each row of X; shape (n_samples,) (ref: log_posterior.py:42-101)."""
    X = np.array(X, ndmin=2, dtype=np.float64)
    log_post = np.zeros(X.shape[0])
    inside = np.all((X > min) & (X < max), axis=1)
    log_post[~inside] = -np.inf
    n_samples = int(np.count_nonzero(inside))
    if n_samples > 0:
        models = device_models(n_div=n_samples)
        total = np.zeros(X.shape[0])
        for dm in models:
            total += dm.logpost(X)          # rows outside the box come back as -inf
        log_post[inside] = total[inside]
    return log_post


# the ensemble sampler recognises this function and takes the fused device path (n_div = 1)
log_posterior._gpemu_device_models = lambda: device_models(n_div=1.0)
