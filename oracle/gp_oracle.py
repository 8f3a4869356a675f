"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A numpy/scipy restatement of the reference's GP-emulator + log-posterior hot path in the
reference's own form (per-sample loops, explicit F x F covariance, LAPACK dpotrf/dpotrs).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / reported baseline.  The product (``bayesian-inference_amd/``)
never imports this package and fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by running the reference itself in the build container
(``tests/golden/make_goldens.py``; sklearn 1.7.2 / scipy 1.15.3 / numpy 2.2.6).  The stretch-move
sampler restatement lives in ``oracle/sampler_oracle.py`` and is "parity unpinned" (emcee is not
available offline, see its header).

Citations: ``ref:`` = /root/reference/src/bayesian_inference/, ``skl:`` = scikit-learn 1.7.2
(sklearn/...), the third-party library the reference delegates its arithmetic to.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
from scipy.linalg import cho_solve, cholesky, lapack, solve_triangular, svd
from scipy.spatial.distance import cdist, pdist, squareform

RBF, MATERN = 0, 1


# --------------------------------------------------------------------------------------------
# model containers (plain arrays; nothing from sklearn)
# --------------------------------------------------------------------------------------------
@dataclass
class KernelSpec:
    """Kernel structure built by ref: emulation.py:132-162 (order: base [+const] [+noise])."""
    kind: int = RBF            # RBF (skl kernels.py:1553) or MATERN (skl kernels.py:1708)
    nu: float = math.inf       # Matern nu in {0.5, 1.5, 2.5}; inf == RBF
    has_const: bool = False    # + ConstantKernel (ref: emulation.py:150-156)
    has_noise: bool = False    # + WhiteKernel    (ref: emulation.py:157-162)

    def n_theta(self, d):
        return d + int(self.has_const) + int(self.has_noise)


@dataclass
class GP:
    """One fitted GaussianProcessRegressor (skl _gpr.py:346-364): kernel_ hyper-parameters, L_, alpha_."""
    ls: np.ndarray             # (d,) length scales
    const: float               # ConstantKernel value (0 if absent)
    noise: float               # WhiteKernel noise level (0 if absent)
    alpha: np.ndarray          # (N,)
    L: np.ndarray              # (N, N) lower Cholesky factor of K + jitter*I


@dataclass
class GroupModel:
    """One emulation group = the results dict of ref: emulation.py:181-192 as plain arrays."""
    X_train: np.ndarray        # (N, d) design
    spec: KernelSpec
    gps: list                  # n_pc GP objects
    components: np.ndarray     # (n_comp, F)  pca.components_
    explained_variance: np.ndarray  # (n_comp,)
    scaler_mean: np.ndarray    # (F,)
    scaler_scale: np.ndarray   # (F,)
    n_pc: int
    extra: dict = field(default_factory=dict)


def split_theta(theta, d, spec: KernelSpec):
    """theta = log([l_1..l_d, (const), (noise)]) (skl kernels.py:733-760, Sum ordering :861-866)."""
    t = np.exp(np.asarray(theta, dtype=np.float64))
    ls = t[:d]
    i = d
    const = noise = 0.0
    if spec.has_const:
        const = float(t[i]); i += 1
    if spec.has_noise:
        noise = float(t[i]); i += 1
    return ls, const, noise


# --------------------------------------------------------------------------------------------
# R2: kernels
# --------------------------------------------------------------------------------------------
def _base_from_dists(dists, spec: KernelSpec):
    """RBF takes squared distances, Matern takes distances (skl kernels.py:1564-1565, 1715-1733)."""
    if spec.kind == RBF:
        return np.exp(-0.5 * dists)
    if spec.nu == 0.5:
        return np.exp(-dists)
    if spec.nu == 1.5:
        K = dists * math.sqrt(3)
        return (1.0 + K) * np.exp(-K)
    if spec.nu == 2.5:
        K = dists * math.sqrt(5)
        return (1.0 + K + K ** 2 / 3.0) * np.exp(-K)
    raise ValueError("Matern nu must be 0.5, 1.5 or 2.5")


def kernel_cross(X, Y, ls, spec: KernelSpec, const=0.0):
    """kernel_(X, Y): base + const; WhiteKernel contributes 0 for Y given (skl kernels.py:1413-1414)."""
    metric = "sqeuclidean" if spec.kind == RBF else "euclidean"
    K = _base_from_dists(cdist(X / ls, Y / ls, metric=metric), spec)
    if spec.has_const:
        K = K + const
    return K


def kernel_diag(n, spec: KernelSpec, const=0.0, noise=0.0):
    """kernel_.diag(X): 1 (+const) (+noise) (skl kernels.py:868-885, 1291-1314, 1433-1435)."""
    v = np.ones(n)
    if spec.has_const:
        v = v + const
    if spec.has_noise:
        v = v + noise
    return v


def kernel_train(X, ls, spec: KernelSpec, const=0.0, noise=0.0, eval_gradient=False):
    """kernel_(X) on the training set, optionally with d/dlog(theta) (N,N,n_theta)."""
    N, d = X.shape
    metric = "sqeuclidean" if spec.kind == RBF else "euclidean"
    dists = pdist(X / ls, metric=metric)
    Kb = squareform(_base_from_dists(dists, spec))
    np.fill_diagonal(Kb, 1)
    K = Kb.copy()
    if spec.has_const:
        K += const
    if spec.has_noise:
        K += noise * np.eye(N)
    if not eval_gradient:
        return K
    D = (X[:, None, :] - X[None, :, :]) ** 2 / (ls ** 2)          # (N,N,d)
    if spec.kind == RBF:
        G = D * Kb[..., None]                                      # skl kernels.py:1573-1579
    elif spec.nu == 0.5:
        den = np.sqrt(D.sum(axis=2))[:, :, None]
        div = np.zeros_like(D)
        np.divide(D, den, out=div, where=den != 0)
        G = Kb[..., None] * div                                    # skl kernels.py:1753-1762
    elif spec.nu == 1.5:
        G = 3 * D * np.exp(-np.sqrt(3 * D.sum(-1)))[..., None]     # skl kernels.py:1763-1764
    else:
        tmp = np.sqrt(5 * D.sum(-1))[..., None]
        G = 5.0 / 3.0 * D * (tmp + 1) * np.exp(-tmp)               # skl kernels.py:1765-1767
    grads = [G]
    if spec.has_const:
        grads.append(np.full((N, N, 1), const))                    # skl kernels.py:1279-1288
    if spec.has_noise:
        grads.append((noise * np.eye(N))[:, :, None])              # skl kernels.py:1403-1408
    return K, np.concatenate(grads, axis=2)


# --------------------------------------------------------------------------------------------
# R3: fit at fixed theta, log-marginal likelihood and gradient
# --------------------------------------------------------------------------------------------
def gp_fit_at_theta(X, y, theta, spec: KernelSpec, jitter=1e-10):
    """skl _gpr.py:346-364: K += alpha*I ; L = cholesky(K, lower) ; alpha_ = cho_solve(L, y)."""
    ls, const, noise = split_theta(theta, X.shape[1], spec)
    K = kernel_train(X, ls, spec, const, noise)
    K[np.diag_indices_from(K)] += jitter
    L = cholesky(K, lower=True, check_finite=False)
    alpha = cho_solve((L, True), y, check_finite=False)
    return GP(ls=ls, const=const, noise=noise, alpha=alpha, L=L)


def lml_and_grad(X, y, theta, spec: KernelSpec, jitter=1e-10):
    """skl _gpr.py:537-652 log_marginal_likelihood(theta, eval_gradient=True), single target."""
    N, d = X.shape
    ls, const, noise = split_theta(theta, d, spec)
    K, dK = kernel_train(X, ls, spec, const, noise, eval_gradient=True)
    K[np.diag_indices_from(K)] += jitter
    L = cholesky(K, lower=True, check_finite=False)
    a = cho_solve((L, True), y, check_finite=False)
    lml = -0.5 * y.dot(a) - np.log(np.diag(L)).sum() - N / 2 * np.log(2 * np.pi)
    Kinv = cho_solve((L, True), np.eye(N), check_finite=False)
    inner = np.outer(a, a) - Kinv
    grad = 0.5 * np.einsum("ij,jik->k", inner, dK)
    return lml, grad


# --------------------------------------------------------------------------------------------
# R1: StandardScaler + PCA
# --------------------------------------------------------------------------------------------
def scaler_fit(Y):
    """skl preprocessing/_data.py:1015-1051 via utils/extmath.py:_incremental_mean_and_var (first batch)."""
    n = Y.shape[0]
    s = np.sum(Y, axis=0)
    mean = s / n
    T = s / n
    temp = Y - T
    corr = np.sum(temp, axis=0)
    unnorm = np.sum(temp ** 2, axis=0) - corr ** 2 / n
    var = unnorm / n
    eps = np.finfo(np.float64).eps
    constant = var <= n * eps * var + (n * mean * eps) ** 2       # _data.py:76-89
    scale = np.sqrt(var)
    scale[constant] = 1.0
    scale[scale == 0.0] = 1.0
    return mean, scale, var


def pca_fit(Ys, n_components=None, u_based=False):
    """skl decomposition/_pca.py:544-702 (_fit_full, LAPACK gesdd) + svd_flip v-based (extmath.py:944-952).
    ``u_based``: the decision of the scikit-learn the reference pins (1.3.0, ref: pdm.lock:1998-1999:
    ``svd_flip(U, Vt)`` with u_based_decision=True -- per column of U, skl utils/extmath.py:934-942).

    Returns dict(mean, components, explained_variance, explained_variance_ratio, Y_pca, flip_argmax).
    """
    n, F = Ys.shape
    mean = np.mean(Ys, axis=0)
    Xc = Ys - mean
    U, S, Vt = svd(Xc, full_matrices=False)
    ev = S ** 2 / (n - 1)
    if u_based:
        idx = np.argmax(np.abs(U), axis=0)
        signs = np.sign(U[idx, np.arange(U.shape[1])])
    else:
        idx = np.argmax(np.abs(Vt), axis=1)
        signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    U = U * signs[None, :]
    Vt = Vt * signs[:, None]
    evr = ev / ev.sum()
    nc = min(n, F) if n_components is None else n_components
    Y_pca = (U * S)[:, :nc]                                        # _pca.py:466-477 (U *= S)
    return dict(mean=mean, components=Vt[:nc], explained_variance=ev[:nc],
                explained_variance_ratio=evr[:nc], Y_pca=Y_pca, flip_argmax=idx[:nc])


# --------------------------------------------------------------------------------------------
# R4: GaussianProcessRegressor.predict(X, return_std=True)
# --------------------------------------------------------------------------------------------
def gp_predict(Xq, X_train, gp: GP, spec: KernelSpec):
    """skl _gpr.py:441-494.  Returns (mean, var) with var = std**2 as the reference squares the
    returned standard deviation again (ref: emulation.py:497-499)."""
    Kt = kernel_cross(Xq, X_train, gp.ls, spec, gp.const)
    mean = Kt @ gp.alpha
    V = solve_triangular(gp.L, Kt.T, lower=True, check_finite=False)
    var = kernel_diag(Xq.shape[0], spec, gp.const, gp.noise)
    var -= np.einsum("ij,ji->i", V.T, V)
    var[var < 0] = 0.0
    std = np.sqrt(var)
    return mean, std ** 2


def gp_predict_all(Xq, model: GroupModel):
    """(B,k) means and variances of all PCs (ref: emulation.py:494-499)."""
    B = Xq.shape[0]
    m = np.zeros((B, model.n_pc))
    v = np.zeros((B, model.n_pc))
    for i, gp in enumerate(model.gps):
        m[:, i], v[:, i] = gp_predict(Xq, model.X_train, gp, model.spec)
    return m, v


# --------------------------------------------------------------------------------------------
# R5-R8: predict_emulation_group / predict
# --------------------------------------------------------------------------------------------
def cov_unexplained(model: GroupModel):
    """ref: emulation.py:246-249."""
    S_un = model.components.T[:, model.n_pc:]
    D_un = np.diag(model.explained_variance[model.n_pc:])
    return S_un.dot(D_un.dot(S_un.T))


def predict_group(Xq, model: GroupModel, cov_unexpl=None):
    """ref: emulation.py:466-548, same order of operations incl. the per-sample loops."""
    if cov_unexpl is None:
        cov_unexpl = cov_unexplained(model)
    n = Xq.shape[0]
    k = model.n_pc
    m, v = gp_predict_all(Xq, model)
    cv_scaled = m.dot(model.components[:k, :])
    cv = cv_scaled * model.scaler_scale + model.scaler_mean        # scaler.inverse_transform
    F = model.components.shape[1]
    S = model.components.T[:, :k]
    cov = np.zeros((n, F, F))
    for i in range(n):
        cov[i] = S.dot(np.diagflat(v[i]).dot(S.T))
    for i in range(n):
        cov[i] += cov_unexpl / n
    cov = cov * np.outer(model.scaler_scale, model.scaler_scale)
    return dict(central_value=cv, cov=cov)


def merge_groups(group_out: dict, mapping: dict, F_total: int):
    """ref: emulation.py:346-406 (SortEmulationGroupObservables.convert) + nd_block_diag (:254-270).

    mapping: {observable: (group, slice_in_output, slice_in_group)} in sorted-observable order.
    """
    some = next(iter(group_out.values()))
    n = some["central_value"].shape[0]
    cv = np.zeros((n, F_total))
    cov = np.zeros((n, F_total, F_total))
    for obs, (g, so, sg) in mapping.items():
        cv[:, so] = group_out[g]["central_value"][:, sg]
        cov[:, so, so] = group_out[g]["cov"][:, sg, sg]
    return dict(central_value=cv, cov=cov)


# --------------------------------------------------------------------------------------------
# R9-R10: log_posterior / _loglikelihood
# --------------------------------------------------------------------------------------------
def loglik_exact(y, cov):
    """ref: log_posterior.py:104-146 (dpotrf upper factor, dpotrs, no 2*pi term)."""
    L, info = lapack.dpotrf(cov, clean=False)
    alpha, info2 = lapack.dpotrs(L, y)
    return -.5 * np.dot(y, alpha) - np.log(L.diagonal()).sum()


def log_posterior(X, models: dict, lo, hi, y_exp, y_err, mapping=None, cov_unexpl=None):
    """ref: log_posterior.py:42-101.  ``models``: {group: GroupModel}; mapping as in merge_groups
    (None for a single group).  n_samples = number of in-bounds rows (the /n_samples quirk).
    ``cov_unexpl``: {group: F_g x F_g} computed once by the caller; None recomputes the truncation covariance in
    every call, which is what the reference does (its wrapper ref: emulation.py:214-224 returns None)."""
    X = np.array(X, ndmin=2, dtype=np.float64)
    out = np.zeros(X.shape[0])
    inside = np.all((X > lo) & (X < hi), axis=1)
    out[~inside] = -np.inf
    n = np.count_nonzero(inside)
    F = y_exp.shape[0]
    if n > 0:
        go = {g: predict_group(X[inside], mdl, None if cov_unexpl is None else cov_unexpl[g])
              for g, mdl in models.items()}
        pred = next(iter(go.values())) if mapping is None else merge_groups(go, mapping, F)
        dY = pred["central_value"] - y_exp
        cov = np.zeros((n, F, F))
        cov += pred["cov"]
        cov += np.diag(y_err ** 2)
        out[inside] += [loglik_exact(a, b) for a, b in zip(dY, cov)]
    return out


# --------------------------------------------------------------------------------------------
# Low-rank form of the same likelihood -- used to check the device kernel's
# algebra independently of the exact form above.
# --------------------------------------------------------------------------------------------
def lowrank_setup(model: GroupModel, y_exp, y_err, n_div=1, cov_unexpl=None):
    """Sigma(theta) = A + U diag(var) U^T ; r = U m + r0.  Returns dict(G,g0,q0,logdetA,U,r0,A)."""
    if cov_unexpl is None:
        cov_unexpl = cov_unexplained(model)
    s = model.scaler_scale
    k = model.n_pc
    A = (cov_unexpl / n_div) * np.outer(s, s) + np.diag(y_err ** 2)
    U = s[:, None] * model.components[:k].T                       # (F,k)
    r0 = model.scaler_mean - y_exp
    cA = cholesky(A, lower=True, check_finite=False)
    AiU = cho_solve((cA, True), U, check_finite=False)
    Air0 = cho_solve((cA, True), r0, check_finite=False)
    return dict(G=U.T @ AiU, g0=U.T @ Air0, q0=float(r0 @ Air0),
                logdetA=float(2 * np.log(np.diag(cA)).sum()), U=U, r0=r0, A=A)


def loglik_lowrank(m, var, st):
    """log p = -1/2 r^T Sigma^-1 r - 1/2 log det Sigma via Woodbury / matrix-determinant lemma."""
    G, g0, q0 = st["G"], st["g0"], st["q0"]
    k = m.shape[0]
    sd = np.sqrt(var)
    M = np.eye(k) + sd[:, None] * G * sd[None, :]
    LM = cholesky(M, lower=True, check_finite=False)
    h = G @ m + g0
    w = solve_triangular(LM, sd * h, lower=True, check_finite=False)
    quad = m @ G @ m + 2 * m @ g0 + q0 - w @ w
    return -0.5 * quad - 0.5 * (st["logdetA"] + 2 * np.log(np.diag(LM)).sum())


def lowrank_setup_blocks(model: GroupModel, y_exp, y_err, block_start, n_div=1, cov_unexpl=None):
    """Per-observable low-rank setups: the reference's merge keeps only the within-observable
    covariance blocks (ref: emulation.py:370-388), so Sigma is block diagonal over observables."""
    full = lowrank_setup(model, y_exp, y_err, n_div, cov_unexpl)
    out = []
    for o in range(len(block_start) - 1):
        sl = slice(int(block_start[o]), int(block_start[o + 1]))
        A, U, r0 = full["A"][sl, sl], full["U"][sl], full["r0"][sl]
        cA = cholesky(A, lower=True, check_finite=False)
        AiU = cho_solve((cA, True), U, check_finite=False)
        Air0 = cho_solve((cA, True), r0, check_finite=False)
        out.append(dict(G=U.T @ AiU, g0=U.T @ Air0, q0=float(r0 @ Air0),
                        logdetA=float(2 * np.log(np.diag(cA)).sum())))
    return out


def loglik_lowrank_blocks(m, var, setups):
    return sum(loglik_lowrank(m, var, st) for st in setups)
