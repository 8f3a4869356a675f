"""DeviceFit: the GP log-marginal likelihood / gradient / factorisation on the device (libgpemu
``gpemu_fit_*``), plus thin wrappers of the stand-alone kernel-matrix and Cholesky entry points."""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


class LinAlgError(np.linalg.LinAlgError):
    pass


class DeviceFit:
    """Workspace for fitting GPs on one design matrix ``X`` (N x d) with a fixed kernel structure.

    theta follows sklearn: log([l_1..l_d, (constant_value), (noise_level)]) (skl kernels.py:733-760).
    """

    def __init__(self, X, kernel_kind=0, nu=np.inf, has_const=False, has_noise=False, jitter=1e-10, device=None):
        _lib.require_device()
        device = _lib.resolve_device(device)
        X = as_f64(X)
        self.N, self.d = X.shape
        self.n_theta = self.d + int(has_const) + int(has_noise)
        h = C.c_void_p()
        check(_lib.lib().gpemu_fit_create(C.byref(h), int(device), self.N, self.d, ptr(X), int(kernel_kind),
                                          float(nu) if np.isfinite(nu) else 0.0, int(has_const), int(has_noise),
                                          float(jitter)))
        self._h = h
        self.n_evaluations = 0          # log-marginal-likelihood evaluations run through this handle
        self.seconds_in_library = 0.0   # wall time inside the (synchronous) C calls: upload, launch chain, download

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().gpemu_fit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc > 0:
            raise LinAlgError(_lib.lib().gpemu_last_error().decode())
        check(rc)

    def lml(self, y, theta, eval_gradient=True):
        y = as_f64(y, (self.N,))
        theta = as_f64(theta, (self.n_theta,))
        val = C.c_double()
        grad = np.empty(self.n_theta) if eval_gradient else None
        self.n_evaluations += 1
        t0 = time.perf_counter()
        rc = _lib.lib().gpemu_fit_lml(self._h, ptr(y), ptr(theta), self.n_theta, C.byref(val), ptr(grad))
        self.seconds_in_library += time.perf_counter() - t0
        self._check(rc)
        return (val.value, grad) if eval_gradient else val.value

    def lml_batch(self, ys, thetas, eval_gradient=True):
        """``n`` (target, theta) pairs evaluated together (one launch chain for all of them): ys (n, N),
        thetas (n, n_theta) -> lml (n,), grad (n, n_theta) or None, info (n,) (non-zero: that kernel matrix is not
        positive definite)."""
        ys = as_f64(ys)
        thetas = as_f64(thetas)
        n = ys.shape[0]
        if ys.shape != (n, self.N) or thetas.shape != (n, self.n_theta):
            raise ValueError("ys must be (n, N) and thetas (n, n_theta)")
        lml = np.empty(n)
        grad = np.empty((n, self.n_theta)) if eval_gradient else None
        info = np.zeros(n, dtype=np.int32)
        self.n_evaluations += n
        t0 = time.perf_counter()
        rc = _lib.lib().gpemu_fit_lml_batch(self._h, n, ptr(ys), ptr(thetas), self.n_theta, ptr(lml), ptr(grad), ptr(info))
        self.seconds_in_library += time.perf_counter() - t0
        check(rc)
        return lml, grad, info

    def factor(self, y, theta):
        """(L_ (N,N) lower, alpha_ (N,), lml) at theta (skl _gpr.py:346-364)."""
        y = as_f64(y, (self.N,))
        theta = as_f64(theta, (self.n_theta,))
        L = np.empty((self.N, self.N))
        alpha = np.empty(self.N)
        val = C.c_double()
        self._check(_lib.lib().gpemu_fit_factor(self._h, ptr(y), ptr(theta), self.n_theta, ptr(L), ptr(alpha),
                                                C.byref(val)))
        return L, alpha, val.value


def kernel_matrix(X, theta, kernel_kind=0, nu=np.inf, has_const=False, has_noise=False, jitter=0.0, device=None):
    device = _lib.resolve_device(device)
    X = as_f64(X)
    N, d = X.shape
    theta = as_f64(theta)
    K = np.empty((N, N))
    check(_lib.lib().gpemu_kernel_matrix(int(device), N, d, ptr(X), ptr(theta), theta.size, int(kernel_kind),
                                         float(nu) if np.isfinite(nu) else 0.0, int(has_const), int(has_noise),
                                         float(jitter), ptr(K)))
    return K


def cholesky(A, device=None):
    """Lower Cholesky factor on the device (scipy.linalg.cholesky(A, lower=True))."""
    device = _lib.resolve_device(device)
    A = np.array(A, dtype=np.float64, order="C")
    rc = _lib.lib().gpemu_cholesky(int(device), A.shape[0], ptr(A))
    if rc > 0:
        raise LinAlgError(_lib.lib().gpemu_last_error().decode())
    check(rc)
    return A


def pca_fit(Y, n_components=None, device=None):
    """StandardScaler + full PCA on the device.  Returns a dict with scaler_mean/scale/var, pca_mean,
    components (nc,F), explained_variance(_ratio) (nc,), Y_pca (N,nc), flip_argmax (nc,), n_sweeps."""
    _lib.require_device()
    device = _lib.resolve_device(device)
    Y = as_f64(Y)
    N, F = Y.shape
    nc = min(N, F) if n_components is None else int(n_components)
    out = dict(scaler_mean=np.empty(F), scaler_scale=np.empty(F), scaler_var=np.empty(F), pca_mean=np.empty(F),
               components=np.empty((nc, F)), explained_variance=np.empty(nc), explained_variance_ratio=np.empty(nc),
               Y_pca=np.empty((N, nc)), flip_argmax=np.empty(nc, dtype=np.int64))
    ns = np.zeros(1, dtype=np.int64)
    check(_lib.lib().gpemu_pca_fit(int(device), N, F, ptr(Y), nc, ptr(out["scaler_mean"]), ptr(out["scaler_scale"]),
                                   ptr(out["scaler_var"]), ptr(out["pca_mean"]), ptr(out["components"]),
                                   ptr(out["explained_variance"]), ptr(out["explained_variance_ratio"]),
                                   ptr(out["Y_pca"]), ptr(out["flip_argmax"]), ptr(ns)))
    out["n_sweeps"] = int(ns[0])
    return out


def truncation_cov(components, explained_variance, n_pc, device=None):
    """S_{>k} diag(explained_variance_{>k}) S_{>k}^T (F x F) on the device (ref: emulation.py:227-251)."""
    _lib.require_device()
    device = _lib.resolve_device(device)
    components = as_f64(components)
    nc, F = components.shape
    ev = as_f64(explained_variance, (nc,))
    out = np.empty((F, F))
    check(_lib.lib().gpemu_truncation_cov(int(device), nc, F, int(n_pc), ptr(components), ptr(ev), ptr(out)))
    return out
