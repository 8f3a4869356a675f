"""Stretch-move ensemble samplers.

``DeviceSampler``    the ensemble lives on one MI355X (libgpemu ``gpemu_sampler_*``); with
                     ``torch.distributed`` initialised the proposing half is sharded over the ranks
                     and the new log-probabilities are all-gathered (RCCL over xGMI) per half-step.
``HostEnsemble``     the same move for an arbitrary Python ``log_prob_fn`` (the emcee calling
                     convention the reference uses, ref: mcmc.py:83-85), vectorised over the
                     proposing half; also shardable over ranks (any torch.distributed backend).
``EnsembleSampler``  emcee-compatible facade (the subset the reference touches: ref: mcmc.py:83-116,
                     187-204, plot_mcmc.py) that picks the device path when ``log_prob_fn`` is bound
                     to device models.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


def shard_bounds(n, world, rank):
    """Contiguous block of ``ceil(n/world)`` proposals per rank: (lo, hi, per)."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi, per


class AutocorrError(Exception):
    """Raised when the chain is too short for a reliable estimate (emcee.autocorr.AutocorrError)."""

    def __init__(self, tau, *args, **kwargs):
        self.tau = tau
        super().__init__(*args, **kwargs)


def _next_pow_two(n):
    i = 1
    while i < n:
        i = i << 1
    return i


def function_1d(x):
    """Normalised autocorrelation function of a 1-D series via FFT (emcee.autocorr.function_1d)."""
    x = np.atleast_1d(x)
    n = _next_pow_two(len(x))
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[: len(x)].real
    acf /= acf[0]
    return acf


def integrated_time(x, c=5, tol=50, quiet=False):
    """Integrated autocorrelation time per dimension with Sokal's window
    (emcee.autocorr.integrated_time; x has shape (steps, walkers, ndim))."""
    x = np.atleast_1d(x)
    if x.ndim == 1:
        x = x[:, np.newaxis, np.newaxis]
    if x.ndim == 2:
        x = x[:, :, np.newaxis]
    n_t, n_w, n_d = x.shape
    tau_est = np.empty(n_d)
    windows = np.empty(n_d, dtype=int)
    for d in range(n_d):
        f = np.zeros(n_t)
        for k in range(n_w):
            f += function_1d(x[:, k, d])
        f /= n_w
        taus = 2.0 * np.cumsum(f) - 1.0
        m = np.arange(len(taus)) < c * taus
        windows[d] = np.argmin(m) if np.any(m) else len(taus) - 1
        tau_est[d] = taus[windows[d]]
    flag = tol * tau_est > n_t
    if np.any(flag) and not quiet:
        msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} "
               "parameter(s). Use this estimate with caution and run a longer chain!\n"
               ).format(tol, np.sum(flag))
        msg += "N/{0} = {1:.0f};\ntau: {2}".format(tol, n_t / tol, tau_est)
        raise AutocorrError(tau_est, msg)
    return tau_est


# ------------------------------------------------------------------------------------------------
class DeviceSampler:
    """Ensemble resident on the device; log-posterior = sum over the given DeviceModels."""

    def __init__(self, models, n_walkers, a=2.0, seed=0):
        _lib.require_device()
        self.models = list(models)
        arr = (C.c_void_p * len(self.models))(*[m.handle for m in self.models])
        h = C.c_void_p()
        check(_lib.lib().gpemu_sampler_create(C.byref(h), arr, len(self.models), int(n_walkers), float(a),
                                              C.c_uint64(int(seed) & (2 ** 64 - 1))))
        self._h = h
        self.W, self.d = int(n_walkers), self.models[0].d
        self.ns = ((self.W + 1) // 2, self.W // 2)
        self.device = self.models[0].device

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().gpemu_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, X0, logp0=None):
        X0 = as_f64(X0, (self.W, self.d))
        lp = None if logp0 is None else as_f64(logp0, (self.W,))
        check(_lib.lib().gpemu_sampler_set_state(self._h, ptr(X0), ptr(lp)))

    def get_state(self):
        X = np.empty((self.W, self.d))
        lp = np.empty(self.W)
        check(_lib.lib().gpemu_sampler_get_state(self._h, ptr(X), ptr(lp)))
        return X, lp

    def reset(self):
        check(_lib.lib().gpemu_sampler_reset(self._h))

    def run(self, steps, store=True):
        rc = _lib.lib().gpemu_sampler_run(self._h, int(steps), int(bool(store)))
        if rc == 1:
            raise ValueError("Probability function returned NaN")
        check(rc)

    def step_host_rng(self, inds, zz, rint, logu, store=True):
        inds = np.ascontiguousarray(inds, dtype=np.int32)
        zz = as_f64(np.concatenate(zz), (self.W,))
        logu = as_f64(np.concatenate(logu), (self.W,))
        rint = np.ascontiguousarray(np.concatenate(rint), dtype=np.int64)
        rc = _lib.lib().gpemu_sampler_step_host_rng(self._h, ptr(inds), ptr(zz), ptr(rint), ptr(logu),
                                                    int(bool(store)))
        if rc == 1:
            raise ValueError("Probability function returned NaN")
        check(rc)

    def counts(self):
        nacc = np.zeros(self.W, dtype=np.int64)
        it, cl = C.c_int64(), C.c_int64()
        check(_lib.lib().gpemu_sampler_get_counts(self._h, ptr(nacc), C.byref(it), C.byref(cl)))
        return nacc, int(it.value), int(cl.value)

    def get_chain(self, first=0, n=None):
        _, _, cl = self.counts()
        n = cl - first if n is None else n
        chain = np.empty((n, self.W, self.d))
        lp = np.empty((n, self.W))
        check(_lib.lib().gpemu_sampler_get_chain(self._h, int(first), int(n), ptr(chain), ptr(lp)))
        return chain, lp

    # -- multi-GPU: one process per GPU, the ensemble replicated, proposals sharded -------------
    def run_sharded(self, steps, store=True, group=None):
        """Same chain as ``run`` (every rank draws identical randomness); rank r evaluates its
        block of each half's proposals and the log-probabilities are all-gathered."""
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if world == 1:
            return self.run(steps, store)
        L = _lib.lib()
        dev = torch.device("cuda", self.device)
        stream = torch.cuda.current_stream(dev)
        check(L.gpemu_sampler_set_stream(self._h, C.c_void_p(stream.cuda_stream)))
        bounds = [shard_bounds(self.ns[h], world, rank) for h in (0, 1)]
        mine = [torch.zeros(b[2], dtype=torch.float64, device=dev) for b in bounds]
        full = [torch.zeros(b[2] * world, dtype=torch.float64, device=dev) for b in bounds]
        try:
            for _ in range(int(steps)):
                check(L.gpemu_sampler_begin_step(self._h))
                for h in (0, 1):
                    lo, hi, _per = bounds[h]
                    check(L.gpemu_sampler_half_propose_eval(self._h, h, lo, hi, C.c_void_p(mine[h].data_ptr())))
                    dist.all_gather_into_tensor(full[h], mine[h], group=group)
                    check(L.gpemu_sampler_half_accept(self._h, h, C.c_void_p(full[h].data_ptr())))
                check(L.gpemu_sampler_end_step(self._h, int(bool(store))))
            rc = L.gpemu_sampler_check(self._h)
            if rc == 1:
                raise ValueError("Probability function returned NaN")
            check(rc)
        finally:
            L.gpemu_sampler_set_stream(self._h, None)


# ------------------------------------------------------------------------------------------------
class HostEnsemble:
    """Stretch move for an arbitrary vectorised ``log_prob_fn(X (n,d)) -> (n,)`` on the host.

    Randomness: numpy ``RandomState`` consumed in emcee's order (choice, shuffle, rand, randint,
    rand...).  With torch.distributed initialised (``sharded=True``) rank r evaluates its block of
    the proposals and the values are all-gathered; every rank keeps the whole ensemble.
    """

    def __init__(self, n_walkers, ndim, log_prob_fn, a=2.0, seed=None, sharded=False, group=None):
        if n_walkers < 2 * ndim:
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than twice "
                               "the number of dimensions.")
        self.W, self.d, self.fn, self.a = n_walkers, ndim, log_prob_fn, a
        self.random = np.random.RandomState(seed)
        self.sharded, self.group = sharded, group
        self.X = None
        self.lp = None
        self.reset()

    def reset(self):
        self.chain, self.lps = [], []
        self.naccepted = np.zeros(self.W, dtype=np.int64)
        self.iterations = 0

    def _eval(self, q):
        if not self.sharded:
            return np.asarray(self.fn(q), dtype=np.float64).reshape(-1)
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        lo, hi, per = shard_bounds(q.shape[0], world, rank)
        mine = torch.zeros(per, dtype=torch.float64)
        if hi > lo:
            mine[: hi - lo] = torch.from_numpy(np.asarray(self.fn(q[lo:hi]), dtype=np.float64).reshape(-1))
        full = torch.zeros(per * world, dtype=torch.float64)
        dist.all_gather_into_tensor(full, mine, group=self.group)
        return full.numpy()[: q.shape[0]].copy()

    def set_state(self, X0, logp0=None):
        self.X = np.array(X0, dtype=np.float64)
        if self.X.shape != (self.W, self.d):
            raise ValueError("incompatible input dimensions")
        self.lp = self._eval(self.X) if logp0 is None else np.array(logp0, dtype=np.float64)
        if np.any(np.isnan(self.lp)):
            raise ValueError("The initial log_prob was NaN")

    def step(self, store=True):
        rs, W, a = self.random, self.W, self.a
        rs.choice(1, p=[1.0])
        inds = np.arange(W) % 2
        rs.shuffle(inds)
        for split in range(2):
            S1 = inds == split
            s, c = self.X[S1], self.X[~S1]
            ns, nc = s.shape[0], c.shape[0]
            zz = ((a - 1.0) * rs.rand(ns) + 1) ** 2.0 / a
            factors = (self.d - 1.0) * np.log(zz)
            rint = rs.randint(nc, size=(ns,))
            q = c[rint] - (c[rint] - s) * zz[:, None]
            new_lp = self._eval(q)
            if np.any(np.isnan(new_lp)):
                raise ValueError("Probability function returned NaN")
            with np.errstate(divide="ignore"):
                logu = np.log(np.array([rs.rand() for _ in range(ns)]))
            acc = factors + new_lp - self.lp[S1] > logu
            widx = np.flatnonzero(S1)[acc]
            self.X[widx] = q[acc]
            self.lp[widx] = new_lp[acc]
            self.naccepted[widx] += 1
        self.iterations += 1
        if store:
            self.chain.append(self.X.copy())
            self.lps.append(self.lp.copy())

    def run(self, steps, store=True):
        for _ in range(int(steps)):
            self.step(store)
