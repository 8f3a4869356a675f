"""Light, picklable stand-ins for the sklearn objects the reference stores in its results dict
(ref: emulation.py:181-192): ``StandardScaler``, ``PCA``, ARD kernels and ``GaussianProcessRegressor``.

They carry the same attribute names downstream code touches (``pca.components_``,
``.explained_variance_``, ``.explained_variance_ratio_``, ``scaler.inverse_transform``, ``.scale_``,
``emulator.predict(X, return_std=True)``, ``.kernel_``) but hold only numpy arrays, so the pickles load
without scikit-learn.  All arithmetic runs on the device through libgpemu; there is no CPU fallback.
"""
from __future__ import annotations

import math
import logging
import os
import time
import warnings
from operator import itemgetter

import numpy as np

logger = logging.getLogger(__name__)

from . import _lib
from . import fit as _fit
from .model import DeviceModel

RBF_KIND, MATERN_KIND = 0, 1


# ------------------------------------------------------------------------------------------------
class StandardScaler:
    """mean_/var_/scale_ as sklearn.preprocessing.StandardScaler (skl _data.py:1015-1051)."""

    def __init__(self):
        self.mean_ = self.var_ = self.scale_ = None
        self.n_samples_seen_ = 0

    def transform(self, X):
        return (np.asarray(X, dtype=np.float64) - self.mean_) / self.scale_

    def inverse_transform(self, X):
        return np.asarray(X, dtype=np.float64) * self.scale_ + self.mean_


class PCA:
    """components_/explained_variance_(_ratio_)/mean_ as sklearn.decomposition.PCA(svd_solver='full')."""

    def __init__(self, n_components=None, svd_solver="full", whiten=False):
        if whiten:
            raise ValueError("whiten=True is not supported (the reference uses whiten=False)")
        self.n_components = n_components
        self.svd_solver = svd_solver
        self.whiten = whiten
        self.components_ = self.explained_variance_ = self.explained_variance_ratio_ = self.mean_ = None

    def transform(self, X):
        return (np.asarray(X, dtype=np.float64) - self.mean_) @ self.components_.T

    def inverse_transform(self, X):
        return np.asarray(X, dtype=np.float64) @ self.components_ + self.mean_


def scale_and_pca(Y, n_components=None, device=None):
    """``pca.fit_transform(scaler.fit_transform(Y))`` on the device (ref: emulation.py:109-117).
    Returns (scaler, pca, Y_pca (N, n_components))."""
    out = _fit.pca_fit(Y, n_components=n_components, device=device)
    scaler = StandardScaler()
    scaler.mean_, scaler.var_, scaler.scale_ = out["scaler_mean"], out["scaler_var"], out["scaler_scale"]
    scaler.n_samples_seen_ = int(np.asarray(Y).shape[0])
    pca = PCA(n_components=n_components)
    pca.components_ = out["components"]
    pca.explained_variance_ = out["explained_variance"]
    pca.explained_variance_ratio_ = out["explained_variance_ratio"]
    pca.mean_ = out["pca_mean"]
    pca.singular_values_ = np.sqrt(out["explained_variance"] * (np.asarray(Y).shape[0] - 1))
    pca.n_components_ = out["components"].shape[0]
    pca.flip_argmax_ = out["flip_argmax"]
    return scaler, pca, out["Y_pca"]


def truncation_covariance(pca, n_pc, device=None):
    """Covariance carried by the discarded components, ``S_{>k} diag(var_{>k}) S_{>k}^T``
    (ref: emulation.py:227-251), as one device GEMM."""
    return _fit.truncation_cov(pca.components_, pca.explained_variance_, n_pc, device=device)


# ------------------------------------------------------------------------------------------------
class ARDKernel:
    """base (+ ConstantKernel) (+ WhiteKernel) in the order the reference builds it
    (ref: emulation.py:132-162).  theta = log([l_1..l_d, (constant), (noise)]) and bounds follow
    sklearn's Sum/hyperparameter ordering (skl kernels.py:733-760, 861-866)."""

    def __init__(self, kind, length_scale, length_scale_bounds, nu=math.inf, constant_value=None,
                 constant_value_bounds=None, noise_level=None, noise_level_bounds=None):
        self.kind = int(kind)
        self.nu = float(nu)
        self.length_scale = np.atleast_1d(np.asarray(length_scale, dtype=np.float64)).copy()
        self.length_scale_bounds = np.atleast_2d(np.asarray(length_scale_bounds, dtype=np.float64)).copy()
        self.constant_value = None if constant_value is None else float(constant_value)
        self.constant_value_bounds = None if constant_value is None else tuple(constant_value_bounds)
        self.noise_level = None if noise_level is None else float(noise_level)
        self.noise_level_bounds = None if noise_level is None else tuple(noise_level_bounds)

    @property
    def has_const(self):
        return self.constant_value is not None

    @property
    def has_noise(self):
        return self.noise_level is not None

    @property
    def n_dims(self):
        return self.length_scale.size + int(self.has_const) + int(self.has_noise)

    @property
    def theta(self):
        t = list(np.log(self.length_scale))
        if self.has_const:
            t.append(math.log(self.constant_value))
        if self.has_noise:
            t.append(math.log(self.noise_level))
        return np.array(t)

    @theta.setter
    def theta(self, value):
        value = np.asarray(value, dtype=np.float64)
        d = self.length_scale.size
        self.length_scale = np.exp(value[:d])
        i = d
        if self.has_const:
            self.constant_value = float(np.exp(value[i]))
            i += 1
        if self.has_noise:
            self.noise_level = float(np.exp(value[i]))

    @property
    def bounds(self):
        b = [np.log(self.length_scale_bounds)]
        if self.has_const:
            b.append(np.log(np.atleast_2d(self.constant_value_bounds)))
        if self.has_noise:
            b.append(np.log(np.atleast_2d(self.noise_level_bounds)))
        return np.vstack(b)

    def clone(self):
        return ARDKernel(self.kind, self.length_scale, self.length_scale_bounds, self.nu, self.constant_value,
                         self.constant_value_bounds, self.noise_level, self.noise_level_bounds)

    def diag_value(self):
        """kernel_.diag: 1 (+ constant) (+ noise) (skl kernels.py:868-885, 1433-1435)."""
        return 1.0 + (self.constant_value or 0.0) + (self.noise_level or 0.0)

    def __repr__(self):
        ls = ", ".join(f"{v:.3g}" for v in self.length_scale)
        if self.kind == RBF_KIND:
            s = f"RBF(length_scale=[{ls}])"
        else:
            s = f"Matern(length_scale=[{ls}], nu={self.nu:.3g})"
        if self.has_const:
            s += f" + {math.sqrt(self.constant_value):.3g}**2"
        if self.has_noise:
            s += f" + WhiteKernel(noise_level={self.noise_level:.3g})"
        return s


class ConvergenceWarning(UserWarning):
    pass


class GaussianProcessRegressor:
    """The subset of sklearn.gaussian_process.GaussianProcessRegressor the reference uses
    (ref: emulation.py:169-172, 497), fitted and evaluated on the device.

    fit: maximise the log-marginal likelihood with L-BFGS-B from the initial theta and
    ``n_restarts_optimizer`` restarts drawn uniformly in the log-bounds from numpy's global RandomState
    (skl _gpr.py:299-337; restarts use ``check_random_state(None)`` = ``np.random.mtrand._rand``).
    """

    def __init__(self, kernel, alpha=1e-10, n_restarts_optimizer=0, copy_X_train=True, optimizer="fmin_l_bfgs_b",
                 device=None):
        self.kernel = kernel
        self.alpha = alpha
        self.n_restarts_optimizer = n_restarts_optimizer
        self.copy_X_train = copy_X_train
        self.optimizer = optimizer
        self.device = device
        self._dev = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_dev"] = None
        return st

    def _optimise(self, dfit, y, theta0, bounds):
        import scipy.optimize

        def obj(theta):
            try:
                lml, grad = dfit.lml(y, theta, eval_gradient=True)
            except _fit.LinAlgError:        # skl _gpr.py:586-590: a kernel matrix that is not positive definite
                return np.inf, np.zeros_like(theta)
            return -lml, -grad
        res = scipy.optimize.minimize(obj, theta0, method="L-BFGS-B", jac=True, bounds=bounds)
        if res.status != 0:
            warnings.warn(f"lbfgs failed to converge (status={res.status}): {res.message}", ConvergenceWarning)
        return res.x, res.fun

    def fit(self, X, y, _device_fit=None):
        X = np.array(X, dtype=np.float64) if self.copy_X_train else np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        self.X_train_, self.y_train_ = X, y
        self.kernel_ = self.kernel.clone()
        k = self.kernel_
        own = _device_fit is None
        dfit = _device_fit or _fit.DeviceFit(X, k.kind, k.nu, k.has_const, k.has_noise, self.alpha, self.device)
        try:
            if self.optimizer is not None and k.n_dims > 0:
                bounds = k.bounds
                optima = [self._optimise(dfit, y, k.theta, bounds)]
                if self.n_restarts_optimizer > 0:
                    if not np.isfinite(bounds).all():
                        raise ValueError("Multiple optimizer restarts (n_restarts_optimizer>0) requires that all "
                                         "bounds are finite.")
                    rng = np.random.mtrand._rand
                    for _ in range(self.n_restarts_optimizer):
                        theta_initial = rng.uniform(bounds[:, 0], bounds[:, 1])
                        optima.append(self._optimise(dfit, y, theta_initial, bounds))
                lml_values = list(map(itemgetter(1), optima))
                k.theta = optima[int(np.argmin(lml_values))][0]
                self._check_bounds(k)
            self.L_, self.alpha_, self.log_marginal_likelihood_value_ = dfit.factor(y, k.theta)
        finally:
            if own:
                dfit.close()
        self._y_train_mean, self._y_train_std = 0.0, 1.0
        return self

    def _adopt(self, dfit, X, y, optima):
        """Take the best of ``optima`` [(theta, -lml)] and factorise there (skl _gpr.py:339-364)."""
        self.X_train_, self.y_train_ = X, y
        if not hasattr(self, "kernel_"):
            self.kernel_ = self.kernel.clone()
        k = self.kernel_
        if optima:
            lml_values = list(map(itemgetter(1), optima))
            k.theta = optima[int(np.argmin(lml_values))][0]
            self._check_bounds(k)
        self.L_, self.alpha_, self.log_marginal_likelihood_value_ = dfit.factor(y, k.theta)
        self._y_train_mean, self._y_train_std = 0.0, 1.0
        return self

    @staticmethod
    def _check_bounds(k):
        th, b = k.theta, k.bounds
        close_lo = np.isclose(b[:, 0], th)
        close_hi = np.isclose(b[:, 1], th)
        if np.any(close_lo) or np.any(close_hi):
            warnings.warn("The optimal value found for some hyper-parameters is close to the specified bound; "
                          "changing the bound and repeating the fit may find a better value.", ConvergenceWarning)

    def log_marginal_likelihood(self, theta=None, eval_gradient=False):
        if theta is None:
            return self.log_marginal_likelihood_value_
        k = self.kernel_
        dfit = _fit.DeviceFit(self.X_train_, k.kind, k.nu, k.has_const, k.has_noise, self.alpha, self.device)
        try:
            return dfit.lml(self.y_train_, theta, eval_gradient=eval_gradient)
        finally:
            dfit.close()

    def _device(self):
        if self._dev is None:
            k = self.kernel_
            self._dev = DeviceModel(
                X_train=self.X_train_, ls=k.length_scale[None, :], alpha=self.alpha_[None, :], L=self.L_[None],
                components=np.ones((1, 1)), scaler_mean=np.zeros(1), scaler_scale=np.ones(1),
                kernel_kind=k.kind, nu=k.nu,
                const=np.array([k.constant_value]) if k.has_const else None,
                noise=np.array([k.noise_level]) if k.has_noise else None, device=self.device)
        return self._dev

    def predict(self, X, return_std=False, return_cov=False):
        if return_cov:
            raise NotImplementedError("return_cov is not on the reference's path")
        X = np.array(X, ndmin=2, dtype=np.float64)
        mean, var = self._device().gp_predict(X)
        if return_std:
            return mean[:, 0], np.sqrt(var[:, 0])
        return mean[:, 0]


class _LockStepEvaluator:
    """Rendezvous of concurrent L-BFGS-B runs on ONE DeviceFit: every optimiser thread hands in its (target, theta)
    and waits; when all running optimisers are waiting (or ``max_batch`` have gathered) the last arriver evaluates
    the whole batch in one launch chain (``gpemu_fit_lml_batch``) and wakes the others.  Each problem's arithmetic
    is what a stand-alone evaluation does, so the optima do not depend on who shares a batch with whom."""

    def __init__(self, dfit, max_batch):
        import threading
        self.dfit, self.max_batch = dfit, int(max_batch)
        self.cv = threading.Condition()
        self.waiting = []
        self.running = 0
        self.error = None

    def enter(self):
        with self.cv:
            self.running += 1

    def leave(self):
        with self.cv:
            self.running -= 1
            if self.waiting and len(self.waiting) >= self.running:
                self._flush()

    def _flush(self):
        batch, self.waiting = self.waiting, []
        try:
            lml, grad, info = self.dfit.lml_batch(np.stack([b["y"] for b in batch]), np.stack([b["theta"] for b in batch]))
            for i, b in enumerate(batch):
                b["out"] = (lml[i], grad[i], int(info[i]))
        except BaseException as exc:        # every waiting optimiser must wake up
            self.error = exc
            for b in batch:
                b["out"] = (np.nan, None, -1)
        self.cv.notify_all()

    def evaluate(self, y, theta):
        slot = {"y": y, "theta": np.array(theta, dtype=np.float64), "out": None}
        with self.cv:
            self.waiting.append(slot)
            if len(self.waiting) >= min(self.running, self.max_batch):
                self._flush()
            while slot["out"] is None:
                self.cv.wait()
            if self.error is not None:
                raise self.error
        lml, grad, info = slot["out"]
        if info != 0:                       # skl _gpr.py:586-590
            return np.inf, np.zeros_like(slot["theta"])
        return -lml, -grad


def _setulb_driver_ok():
    """True when scipy's L-BFGS-B routine has the reverse-communication signature this module drives directly
    (scipy 1.15: ``setulb(m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task)``);
    otherwise ``fit_gps`` runs one ``scipy.optimize.minimize`` per host thread."""
    if os.environ.get("GPEMU_FIT_DRIVER", "") == "threads":
        return False
    try:
        from scipy.optimize import _lbfgsb
        doc = _lbfgsb.setulb.__doc__ or ""
        return "ln_task" in doc and "csave" not in doc
    except Exception:
        return False


class _LbfgsbRun:
    """One L-BFGS-B minimisation held as explicit state around ``_lbfgsb.setulb``: what
    ``scipy.optimize.minimize(fun, x0, method="L-BFGS-B", jac=True, bounds=bounds)`` does with its default options
    (maxcor 10, ftol 2.22e-9, gtol 1e-5, maxls 20, maxiter = maxfun = 15000), cut at the points where it asks for a
    function value, so that many runs can advance together and have their values computed in one batch.  The
    sequence of points, values and the result are those of ``minimize`` (scipy/optimize/_lbfgsb_py.py: the
    objective is first evaluated at the clipped start point, which is also the first point the routine asks for)."""

    _setulb = None                    # scipy.optimize._lbfgsb.setulb, looked up once

    def __init__(self, x0, bounds):
        n = len(x0)
        self.m, self.maxls, self.maxiter, self.maxfun = 10, 20, 15000, 15000
        self.factr, self.pgtol = 2.220446049250313e-09 / np.finfo(float).eps, 1e-5
        lo, hi = np.asarray(bounds, dtype=np.float64)[:, 0], np.asarray(bounds, dtype=np.float64)[:, 1]
        self.x = np.clip(np.asarray(x0, dtype=np.float64).ravel(), lo, hi)
        self.nbd = np.zeros(n, np.int32)
        self.low, self.up = np.zeros(n), np.zeros(n)
        for i in range(n):
            fin_l, fin_u = np.isfinite(lo[i]), np.isfinite(hi[i])
            if fin_l:
                self.low[i] = lo[i]
            if fin_u:
                self.up[i] = hi[i]
            self.nbd[i] = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}[(bool(fin_l), bool(fin_u))]
        m = self.m
        self.f = 0.0
        self.g = np.zeros(n)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29)
        self.nit = self.nfev = 0
        self.done = False
        self.last_x = None            # point of the newest value (the memo of scipy's ScalarFunction)

    def advance(self):
        """Run the routine up to its next request.  Returns the point to evaluate, or None when finished."""
        setulb = _LbfgsbRun._setulb
        if setulb is None:
            from scipy.optimize import _lbfgsb
            setulb = _LbfgsbRun._setulb = _lbfgsb.setulb
        task = self.task
        while True:
            setulb(self.m, self.x, self.low, self.up, self.nbd, self.f, self.g, self.factr, self.pgtol, self.wa,
                   self.iwa, task, self.lsave, self.isave, self.dsave, self.maxls, self.ln_task)
            if task[0] == 3:
                # (np.array_equal's answer for two finite float arrays of one shape, at a third of its cost: with ~150
                # design points the group fit is bound by this host loop, ~13 us per evaluation of which the routine
                # itself is two calls of ~6 us)
                if self.last_x is not None and (self.x == self.last_x).all():
                    continue            # value already known (scipy's memo): hand it straight back
                return self.x.copy()
            if task[0] == 1:
                self.nit += 1
                if self.nit >= self.maxiter:
                    task[0], task[1] = 5, 504
                elif self.nfev > self.maxfun:
                    task[0], task[1] = 5, 502
                continue
            self.done = True
            return None

    def supply(self, x, f, g):
        self.f = float(f)
        self.g = g if (type(g) is np.ndarray and g.dtype == np.float64 and g.flags.c_contiguous) else \
            np.ascontiguousarray(g, dtype=np.float64)
        self.last_x = x
        self.nfev += 1

    @property
    def status(self):
        if self.task[0] == 4:
            return 0
        return 1 if (self.nfev > self.maxfun or self.nit >= self.maxiter) else 2


def fit_batch_size(n_design, requested=None, device=None):
    """Problems (target, theta) evaluated through one launch chain: ``GPEMU_FIT_BATCH`` (default 64), within a memory
    budget -- every problem of a batch owns ~6 N x N f64 work matrices on the device, and a batch may take a quarter of
    the MI355X's 288 GB (``GPEMU_FIT_BATCH_GB``, default 72): 64 problems up to N ~ 4800, 58 at N = 5000.  The serial
    steps of the factorisation cost the same for 8 problems as for 64, so large batches are what the chain wants
    (N = 5000: 3.13 / 2.93 / 2.85 ms per problem at 8 / 16 / 32)."""
    if requested is None:
        requested = int(os.environ.get("GPEMU_FIT_BATCH", "64"))
    budget = float(os.environ.get("GPEMU_FIT_BATCH_GB", "72")) * 1e9
    # never more than half of what the device has FREE right now (a smaller part, or memory already held by chains and
    # models): the reservation would fail with GPEMU_ERR_HIP and abort the fit instead of shrinking the batch
    free = _lib.device_free_bytes(device)
    if free is not None:
        budget = min(budget, 0.5 * free)
    n_pad = -(-int(n_design) // 64) * 64
    return max(1, min(int(requested), int(budget // (6 * 8 * n_pad * n_pad))))


def _lockstep_minimise(dfit, problems, bounds, max_batch, groups=None, second=None):
    """``problems``: list of (target y, start theta).  All minimisations advance together: every round takes the next
    requested point of each run of a group (at most ``max_batch``), evaluates them in ONE launch chain
    (``gpemu_fit_lml_batch``) and hands the values back.  Returns ([(theta, minimum)] in the order given, seconds with at
    least one evaluation inside the library -- the union of the handles' busy intervals).

    Side effect, for the duration of the call only: with more than one group the interpreter's thread switch interval is
    lowered to 20 us (``sys.setswitchinterval``, process wide -- other Python threads of the process switch more often
    meanwhile) and restored in a ``finally``.

    ``groups`` = 2 (default when there is more than one batch of problems; GPEMU_FIT_GROUPS): two groups of runs
    alternate -- while the device evaluates the points of one (a worker thread inside the C call, the GIL released), this
    thread hands the other group's values back and advances its optimisers to their next requests.  At C3 the host side
    of a round (64 ``setulb`` steps) is a quarter of the evaluation time, which was added to it before.  The runs are
    independent and a batched evaluation has the bits of a single one, so every run visits the points it visits alone,
    whatever group it is in.

    ``second``: further device handles (own stream, own work matrices), one per further group.  The evaluations of group
    g go through handle g on a worker thread of its own, so the groups' launch chains are on the device TOGETHER: a chain
    of batched evaluations leaves most CUs idle during its serial stretches (16 diagonal-block launches on one CU per
    problem, launches at the launch floor), which the other groups' chains fill."""
    import concurrent.futures
    if groups is None:
        groups = int(os.environ.get("GPEMU_FIT_GROUPS", "2"))
    extra = [] if second is None else (list(second) if isinstance(second, (list, tuple)) else [second])
    # (more groups than two only where every group has a handle of its own: groups that take turns on ONE handle gain
    # nothing beyond the second)
    groups = max(1, min(int(groups), max(2, 1 + len(extra)))) if len(problems) > max_batch else 1
    # no more groups than batches of problems
    groups = max(1, min(groups, -(-len(problems) // max_batch)))
    handles = [dfit] + [extra[g - 1] if g - 1 < len(extra) else dfit for g in range(1, groups)]
    n_workers = len({id(h) for h in handles})
    runs = [None] * len(problems)
    results = [None] * len(problems)
    nxt = 0
    active = [[] for _ in range(groups)]
    pending = [None] * groups            # (future of the evaluation in flight, the points it was asked for)

    busy = []                            # (start, end) of every evaluation: the device's busy time is their union

    def evaluate(h, ys, xs):
        t0 = time.perf_counter()
        out = h.lml_batch(ys, xs)
        busy.append((t0, time.perf_counter()))
        return out

    def hand_back(g):
        fut, ask = pending[g]
        pending[g] = None
        lml, grad, info = fut.result()
        # (whole-array negation and plain Python scalars: this loop runs once per evaluation of the fit, and with ~150
        # design points the fit is bound by the host side of an evaluation, not by the device)
        neg_lml, neg_grad, bad = (-np.asarray(lml, dtype=np.float64)).tolist(), -np.asarray(grad, dtype=np.float64), \
            np.asarray(info).tolist()
        for j, (idx, x) in enumerate(ask):
            if bad[j] != 0:             # skl _gpr.py:586-590: not positive definite -> +inf, zero gradient
                runs[idx].supply(x, np.inf, np.zeros_like(x))
            else:
                runs[idx].supply(x, neg_lml[j], neg_grad[j])

    def advance(g):
        """refill the group, run its optimisers to their next requests; the points asked for (may be empty)"""
        nonlocal nxt
        while True:
            while len(active[g]) < max_batch and nxt < len(problems):
                runs[nxt] = _LbfgsbRun(problems[nxt][1], bounds)
                active[g].append(nxt)
                nxt += 1
            ask, still = [], []
            for idx in active[g]:
                x = runs[idx].advance()
                if x is None:
                    r = runs[idx]
                    if r.status != 0:
                        warnings.warn(f"lbfgs failed to converge (status={r.status})", ConvergenceWarning)
                    results[idx] = (r.x, r.f)
                else:
                    ask.append((idx, x))
                    still.append(idx)
            active[g] = still
            if ask or nxt >= len(problems):
                return ask
            # every run of the group finished in this round and there are problems left: start them right away

    # The device thread needs the interpreter lock for a few microseconds between two evaluations, while this thread is
    # in the middle of a round of optimiser steps: with the default switch interval (5 ms) it would get it when the round
    # is over, the device idle meanwhile.
    import sys
    switch_interval = sys.getswitchinterval()
    if groups > 1:
        sys.setswitchinterval(2e-5)
    try:
        with concurrent.futures.ThreadPoolExecutor(max_workers=n_workers) as device:     # one evaluation per handle at a time
            g = 0
            while True:
                if pending[g] is not None:
                    hand_back(g)
                ask = advance(g)
                if ask:
                    ys = np.stack([problems[i][0] for i, _ in ask])
                    xs = np.stack([x for _, x in ask])
                    pending[g] = (device.submit(evaluate, handles[g], ys, xs), ask)
                if all(p is None for p in pending) and not any(active) and nxt >= len(problems):
                    break
                g = (g + 1) % groups
    finally:
        sys.setswitchinterval(switch_interval)
    # wall time with at least one evaluation inside the library (with two handles the calls overlap)
    union, end = 0.0, -1.0
    for t0, t1 in sorted(busy):
        if t1 > end:
            union += t1 - max(t0, end)
            end = t1
    return results, union


def fit_gps(design, Y_columns, kernel, alpha=1e-10, n_restarts_optimizer=0, copy_X_train=False, device=None,
            n_streams=None):
    """One GaussianProcessRegressor per column of ``Y_columns`` (N x k), fitted concurrently.

    The k GPs and their restarts are ``k (1 + n_restarts)`` independent L-BFGS-B problems on the same design
    (ref: emulation.py:169-172 fits them one after the other).  They are spread over a pool of host threads,
    advancing in LOCK STEP on one ``DeviceFit``: whenever all running optimisers wait for a function value, the
    whole batch of (target, theta) pairs goes through ONE chain of launches (``gpemu_fit_lml_batch``: kernel matrices,
    blocked Cholesky, triangular inverses, K^-1 and gradient contractions of all problems together).  The
    restart points are drawn first, from numpy's global RandomState in the order the sequential loop draws
    them, so the result does not depend on the schedule.  ``n_streams`` (env GPEMU_FIT_BATCH, default 64, less for
    large N): optimisers in flight = problems per launch chain; 1 reproduces the sequential loop.  Where scipy's
    L-BFGS-B routine has the expected reverse-communication signature, ONE host thread drives all runs through it
    (``_lockstep_minimise``; same points, values and results as ``scipy.optimize.minimize``), otherwise -- or with
    GPEMU_FIT_DRIVER=threads -- every run is a ``minimize`` call on its own host thread, meeting the others in
    ``_LockStepEvaluator``.
    """
    import concurrent.futures

    X = np.array(design, dtype=np.float64) if copy_X_train else np.asarray(design, dtype=np.float64)
    Yc = np.asarray(Y_columns, dtype=np.float64)
    k_gp = Yc.shape[1]
    gprs = [GaussianProcessRegressor(kernel=kernel, alpha=alpha, n_restarts_optimizer=n_restarts_optimizer,
                                     copy_X_train=copy_X_train, device=device) for _ in range(k_gp)]
    for g in gprs:
        g.kernel_ = g.kernel.clone()
    kk = gprs[0].kernel_ if gprs else None
    optimise = bool(gprs) and gprs[0].optimizer is not None and kk.n_dims > 0
    starts = [[] for _ in range(k_gp)]
    if optimise:
        bounds = kk.bounds
        if n_restarts_optimizer > 0 and not np.isfinite(bounds).all():
            raise ValueError("Multiple optimizer restarts (n_restarts_optimizer>0) requires that all "
                             "bounds are finite.")
        rng = np.random.mtrand._rand
        for i in range(k_gp):
            starts[i].append(np.array(kk.theta))
            for _ in range(n_restarts_optimizer):
                starts[i].append(rng.uniform(bounds[:, 0], bounds[:, 1]))
    if n_streams is None:
        n_streams = fit_batch_size(X.shape[0], device=device)
    tasks = [(i, j) for i in range(k_gp) for j in range(len(starts[i]))]
    n_threads = max(1, min(int(n_streams), max(len(tasks), 1)))
    shared = _fit.DeviceFit(X, kk.kind, kk.nu, kk.has_const, kk.has_noise, alpha, device)
    handles = [shared]
    # a handle of its own for every group of the lock-step driver (GPEMU_FIT_HANDLES=1: two groups taking turns on one)
    # measured (tools/fit_c3_groups.py, tools/time_lml_batch_handles.py): the C3 fit 1.09 / 0.97 / 0.89 / 0.89-0.95 s with
    # 1 / 2 / 3 / 4 handles, 64 x N = 1000 evaluations 45 / 37 / 34 / 40 us per problem; 58 x N = 5000 2.74 / 2.70 / 2.73 ms
    # per problem -- large problems fill the chip by themselves and keep to one handle (and a third of the memory)
    n_handles = int(os.environ.get("GPEMU_FIT_HANDLES", "3" if X.shape[0] <= 2048 else "1"))
    n_groups = int(os.environ.get("GPEMU_FIT_GROUPS", str(max(2, n_handles))))
    n_handles = max(1, min(n_handles, n_groups, -(-len(tasks) // max(n_threads, 1)))) if n_threads > 1 else 1
    if n_handles > 1:    # all handles' work matrices must fit beside each other
        n_pad = -(-X.shape[0] // 64) * 64
        free = _lib.device_free_bytes(device)
        while n_handles > 1 and free is not None and n_handles * 6.0 * 8 * n_pad * n_pad * n_threads >= 0.8 * free:
            n_handles -= 1
    lockstep = n_threads > 1 and optimise and _setulb_driver_ok()
    if not lockstep:
        n_handles = 1               # the other drivers evaluate through `shared` alone: no further work matrices
    two_handles = n_handles > 1
    evaluator = _LockStepEvaluator(shared, n_threads)
    columns = [np.ascontiguousarray(Yc[:, i]) for i in range(k_gp)]

    class _Shared:                      # what GaussianProcessRegressor._optimise calls on its fit handle
        def lml(self, y, theta, eval_gradient=True):
            val, grad = evaluator.evaluate(y, theta)
            if not np.isfinite(val):
                raise _fit.LinAlgError("kernel matrix is not positive definite")
            return -val, -grad

    def run_start(task):
        i, j = task
        evaluator.enter()
        try:
            if n_threads == 1:          # the sequential loop, stand-alone evaluations
                return gprs[i]._optimise(shared, columns[i], starts[i][j], kk.bounds)
            return gprs[i]._optimise(_Shared(), columns[i], starts[i][j], kk.bounds)
        finally:
            evaluator.leave()

    t_fit = time.perf_counter()
    t_busy = None
    try:
        for _ in range(n_handles - 1):      # inside the try: a failed allocation must not leak the handles made so far
            handles.append(_fit.DeviceFit(X, kk.kind, kk.nu, kk.has_const, kk.has_noise, alpha, device))
        if lockstep:
            # one host thread drives all L-BFGS-B runs through the routine's reverse-communication interface
            driver = ("lockstep (scipy.optimize._lbfgsb.setulb, one host thread, "
                      + (f"{n_handles} groups of runs, each with its own device handle: their evaluations overlap on the device)"
                         if two_handles else "two groups of runs alternating on the device)"))
            optima, t_busy = _lockstep_minimise(shared, [(columns[i], starts[i][j]) for i, j in tasks], kk.bounds, n_threads,
                                                groups=n_groups if two_handles else None,
                                                second=handles[1:] if two_handles else None)
        else:
            driver = ("sequential (scipy.optimize.minimize)" if n_threads == 1 else
                      f"threads ({n_threads} x scipy.optimize.minimize meeting in a lock-step evaluator)")
            with concurrent.futures.ThreadPoolExecutor(max_workers=n_threads) as pool:
                optima = list(pool.map(run_start, tasks))
        per_gp = [[] for _ in range(k_gp)]
        for (i, _j), opt in zip(tasks, optima):
            per_gp[i].append(opt)
        for i in range(k_gp):
            gprs[i]._adopt(shared, X, columns[i], per_gp[i])
    finally:
        n_eval = sum(h.n_evaluations for h in handles)
        t_lib = sum(h.seconds_in_library for h in handles)
        if len(handles) > 1 and t_busy is not None:     # overlapping calls: the union of their busy intervals
            t_lib = t_busy
        for h in handles:
            h.close()
    t_fit = time.perf_counter() - t_fit
    # which L-BFGS-B driver ran (the fast one depends on a private scipy routine's signature) and where the time went
    logger.info(f"fit_gps: {len(tasks)} L-BFGS-B runs for {k_gp} GPs, {n_eval} LML evaluations, driver: {driver}; "
                f"{t_fit:.2f} s of which {t_lib:.2f} s inside libgpemu (device), {t_fit - t_lib:.2f} s in the optimiser "
                f"on the host")
    for g in gprs:
        g.n_lml_evaluations_ = n_eval          # of the whole group fit (all GPs and restarts)
        g.fit_driver_ = driver
        g.fit_seconds_ = (t_fit, t_lib)        # (whole group fit, inside the library)
    return gprs
