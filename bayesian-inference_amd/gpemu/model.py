"""DeviceModel: one emulation group resident on one MI355X (a libgpemu model handle)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr

RBF, MATERN = 0, 1
LOWRANK, EXACT = 0, 1


class DeviceModel:
    """Owns a ``gpemu_model`` handle.

    Arrays follow the results dict of the reference (ref: emulation.py:181-192):
    ``X_train`` = GaussianProcessRegressor.X_train_, per-PC ``ls/const/noise`` = kernel_
    hyper-parameters, ``alpha`` = alpha_, ``L`` = L_, ``components`` = pca.components_[:k],
    ``scaler_mean/scale`` = StandardScaler, ``cov_unexplained`` = ref: emulation.py:246-249.
    """

    def __init__(self, X_train, ls, alpha, L, components, scaler_mean, scaler_scale,
                 kernel_kind=RBF, nu=np.inf, const=None, noise=None, cov_unexplained=None, device=None):
        _lib.require_device()
        device = _lib.resolve_device(device)
        X_train = as_f64(X_train)
        N, d = X_train.shape
        ls = as_f64(ls)
        k = ls.shape[0]
        components = as_f64(components)
        F = components.shape[1]
        ls = as_f64(ls, (k, d))
        alpha = as_f64(alpha, (k, N))
        L = as_f64(L, (k, N, N))
        components = as_f64(components, (k, F))
        scaler_mean = as_f64(scaler_mean, (F,))
        scaler_scale = as_f64(scaler_scale, (F,))
        const_a = None if const is None else as_f64(const, (k,))
        noise_a = None if noise is None else as_f64(noise, (k,))
        cu = None if cov_unexplained is None else as_f64(cov_unexplained, (F, F))
        h = C.c_void_p()
        check(_lib.lib().gpemu_model_create(
            C.byref(h), int(device), N, d, F, k, int(kernel_kind),
            float(nu) if np.isfinite(nu) else 0.0, int(const is not None), int(noise is not None),
            ptr(X_train), ptr(ls), ptr(const_a), ptr(noise_a), ptr(alpha), ptr(L), ptr(components),
            ptr(scaler_mean), ptr(scaler_scale), ptr(cu)))
        self._h = h
        self.N, self.d, self.F, self.k, self.device = N, d, F, k, int(device)
        self._lik_key = None

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().gpemu_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(_lib.lib().gpemu_model_sync(self._h))

    def profile(self, enable=True):
        """Bracket every launch of the two hot kernels with HIP events (bench.py roofline leg)."""
        check(_lib.lib().gpemu_model_profile(self._h, int(bool(enable))))

    def profile_read(self):
        """{'trmm_vsq': (ms_total, launches), 'kstar': (ms_total, launches)}"""
        ms = np.zeros(2)
        n = np.zeros(2, dtype=np.int64)
        check(_lib.lib().gpemu_model_profile_read(self._h, ptr(ms), ptr(n)))
        return {"trmm_vsq": (float(ms[0]), int(n[0])), "kstar": (float(ms[1]), int(n[1]))}

    # -- host-buffer API -----------------------------------------------------------------------
    def _X(self, X):
        X = np.array(X, ndmin=2, dtype=np.float64)
        X = np.ascontiguousarray(X)
        if X.shape[1] != self.d:
            raise ValueError(f"expected {self.d} parameters per row, got {X.shape[1]}")
        return X

    @staticmethod
    def _finite(X):
        """sklearn's GaussianProcessRegressor.predict validates its input (check_array: ValueError on NaN / inf; skl
        utils/validation.py), which is what the reference's predict path goes through (ref: emulation.py:497).  The
        cross-kernel on the matrix cores does not propagate a NaN coordinate into K_* (only into the mean), so the
        host entry points refuse non-finite queries the way the reference does.  log_posterior is not affected: a NaN
        parameter fails the box prior (-inf), as in ref: log_posterior.py:63-64."""
        if not np.isfinite(X).all():
            raise ValueError("Input X contains NaN or infinity.")
        return X

    def gp_predict(self, X):
        """(B,k) predictive means and variances of the k PCs (ref: emulation.py:494-499)."""
        X = self._finite(self._X(X))
        B = X.shape[0]
        mean = np.empty((B, self.k))
        var = np.empty((B, self.k))
        check(_lib.lib().gpemu_gp_predict(self._h, B, ptr(X), ptr(mean), ptr(var)))
        return mean, var

    def predict_full(self, X, n_div=None):
        """central_value (B,F), cov (B,F,F) as ref: emulation.py:466-548 (n_div defaults to B)."""
        X = self._finite(self._X(X))
        B = X.shape[0]
        cv = np.empty((B, self.F))
        cov = np.empty((B, self.F, self.F))
        check(_lib.lib().gpemu_predict_full(self._h, B, ptr(X), float(B if n_div is None else n_div),
                                            ptr(cv), ptr(cov)))
        return cv, cov

    def likelihood_setup(self, y_exp, y_err, lo, hi, n_div=1.0, block_start=None):
        """Data, box prior and the observable block boundaries of this group (``block_start`` =
        first feature of each observable plus F at the end; None = a single block).  ``y_exp`` of shape (C, F):
        one data vector per chain of a multi-chain sampler (closure tests), all against the same ``y_err``."""
        y_exp = np.ascontiguousarray(y_exp, dtype=np.float64)
        if y_exp.ndim == 2:
            return self._likelihood_setup_chains(y_exp, y_err, lo, hi, n_div, block_start)
        y_exp = as_f64(y_exp, (self.F,))
        y_err = as_f64(y_err, (self.F,))
        lo = as_f64(lo, (self.d,))
        hi = as_f64(hi, (self.d,))
        bs = None
        nb = 0
        if block_start is not None:
            bs = np.ascontiguousarray(block_start, dtype=np.int64)
            nb = bs.size - 1
        check(_lib.lib().gpemu_likelihood_setup(self._h, ptr(y_exp), ptr(y_err), ptr(lo), ptr(hi),
                                                float(n_div), nb, ptr(bs)))
        self._lik_key = (float(n_div), None if bs is None else tuple(bs.tolist()))

    def _likelihood_setup_chains(self, y_exp, y_err, lo, hi, n_div, block_start):
        n_chains = y_exp.shape[0]
        y_exp = as_f64(y_exp, (n_chains, self.F))
        y_err = as_f64(y_err, (self.F,))
        lo = as_f64(lo, (self.d,))
        hi = as_f64(hi, (self.d,))
        bs, nb = None, 0
        if block_start is not None:
            bs = np.ascontiguousarray(block_start, dtype=np.int64)
            nb = bs.size - 1
        check(_lib.lib().gpemu_likelihood_setup_chains(self._h, int(n_chains), ptr(y_exp), ptr(y_err), ptr(lo), ptr(hi),
                                                       float(n_div), nb, ptr(bs)))
        self._lik_key = (float(n_div), None if bs is None else tuple(bs.tolist()))

    def logpost(self, X, mode=LOWRANK):
        X = self._X(X)
        B = X.shape[0]
        out = np.empty(B)
        check(_lib.lib().gpemu_logpost(self._h, B, ptr(X), ptr(out), int(mode)))
        return out

    # -- device-pointer API (torch tensors on this device; stream = torch's current stream) ----
    def logpost_dev(self, dX_ptr, B, dout_ptr, mode=LOWRANK, stream=0):
        check(_lib.lib().gpemu_logpost_dev(self._h, int(B), C.c_void_p(dX_ptr), C.c_void_p(dout_ptr),
                                           int(mode), C.c_void_p(stream)))

    def gp_predict_dev(self, dX_ptr, B, dmean_ptr, dvar_ptr, stream=0):
        check(_lib.lib().gpemu_gp_predict_dev(self._h, int(B), C.c_void_p(dX_ptr), C.c_void_p(dmean_ptr),
                                              C.c_void_p(dvar_ptr), C.c_void_p(stream)))

    def predict_full_dev(self, dX_ptr, B, n_div, dcv_ptr, dcov_ptr, stream=0):
        check(_lib.lib().gpemu_predict_full_dev(self._h, int(B), C.c_void_p(dX_ptr), float(n_div),
                                                C.c_void_p(dcv_ptr), C.c_void_p(dcov_ptr), C.c_void_p(stream)))
