"""-m gpu parity tests of the fit side: kernel matrix, blocked Cholesky, log-marginal likelihood and
gradient at identical theta against the goldens (reference run) and the CPU oracle."""
import numpy as np
import pytest

import golden_util as GU
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
SYN = ["g1_rbf_noise", "g1_matern15_noise", "g1_matern25_const_noise", "g1_rbf_only", "g2_rbf_noise"]


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _fit_for(g, design=None):
    from gpemu.fit import DeviceFit
    spec = GU.spec_of(g)
    X = g["design"] if design is None else design
    return DeviceFit(X, kernel_kind=spec.kind, nu=spec.nu, has_const=spec.has_const, has_noise=spec.has_noise,
                     jitter=float(g["gpr_alpha"])), spec, X


@pytest.mark.parametrize("name", SYN)
def test_kernel_matrix_and_cholesky(name):
    from gpemu.fit import cholesky, kernel_matrix
    g = GU.load(name)
    spec = GU.spec_of(g)
    X = g["design"]
    th = g["theta"][0]
    ls, c, nz = O.split_theta(th, X.shape[1], spec)
    Kref = O.kernel_train(X, ls, spec, c, nz)
    K = kernel_matrix(X, th, spec.kind, spec.nu, spec.has_const, spec.has_noise, jitter=0.0)
    assert relerr(K, Kref) < 1e-13
    Kj = Kref + float(g["gpr_alpha"]) * np.eye(X.shape[0])
    L = cholesky(Kj)
    assert np.allclose(np.triu(L, 1), 0.0)
    j = list(g["L_index"]).index(0)
    assert relerr(L, g["L"][j]) < 1e-8          # sklearn's L_ for PC 0 (same theta)


@pytest.mark.parametrize("name", SYN + ["g3_realdata_matern15"])
def test_lml_grad_factor_at_golden_theta(name):
    g = GU.load(name)
    design = GU.load("observables_fixture")["design"] if name.startswith("g3") else None
    fit, spec, X = _fit_for(g, design)
    ytr = g["Y_pca_truncated"]
    cond_loose = not spec.has_noise          # cond(K) ~ 1e9 without a noise term
    for i in range(int(g["n_pc"])):
        for th, lk, gk in ((g["theta"][i], "lml_at_theta", "grad_at_theta"),
                           (g["theta2"][i], "lml_at_theta2", "grad_at_theta2")):
            lml, grad = fit.lml(ytr[:, i], th)
            assert abs(lml - g[lk][i]) <= (1e-6 if cond_loose else 1e-8) * max(1.0, abs(g[lk][i]))
            scale = max(1.0, np.max(np.abs(g[gk][i])))
            assert np.max(np.abs(grad - g[gk][i])) <= (1e-4 if cond_loose else 1e-6) * scale
    for j, i in enumerate(g["L_index"]):
        L, alpha, lml = fit.factor(ytr[:, i], g["theta"][i])
        assert relerr(L, g["L"][j]) < 1e-8
        assert relerr(alpha, g["alpha"][i]) < (1e-4 if cond_loose else 1e-7)
        assert abs(lml - g["lml_value"][i]) <= 1e-6 * max(1.0, abs(g["lml_value"][i]))
    fit.close()


def test_not_positive_definite_raises():
    from gpemu.fit import LinAlgError, cholesky
    A = np.eye(70)
    A[5, 5] = -1.0
    with pytest.raises(LinAlgError):
        cholesky(A)


@pytest.mark.parametrize("bad", [-1.0, 0.0])
def test_first_non_positive_pivot_is_reported_like_lapack(bad):
    """dpotrf's info = order of the first leading minor that is not positive definite.  The device finds it from the
    1 / sqrt(pivot) values a sweep leaves behind (a NaN from the first bad pivot on): first, last and inner pivots of the
    16-column panels, of the first and of later 64 x 64 blocks; later bad pivots do not overwrite the first."""
    from gpemu import _lib
    from gpemu.fit import LinAlgError, cholesky
    n = 200
    for p in (0, 1, 15, 16, 31, 47, 48, 63, 64, 79, 127, 128, 199):
        A = np.eye(n) * 4.0 + 0.01
        A[p, p] = bad
        if p + 3 < n:
            A[p + 3, p + 3] = -5.0
        with pytest.raises(LinAlgError):
            cholesky(A)
        assert f"(pivot {p + 1})" in _lib.lib().gpemu_last_error().decode(), (p, _lib.lib().gpemu_last_error().decode())
    L = cholesky(np.eye(n) * 4.0 + 0.01)        # and the handle still factors a good matrix
    assert relerr(np.tril(L), np.linalg.cholesky(np.eye(n) * 4.0 + 0.01)) < 1e-13


def test_cholesky_random_sizes_same_bits_twice_and_lapack():
    """The column sweep of a diagonal block hands its columns from wave to wave through LDS flags: random SPD matrices of
    many sizes (one block, several blocks, several panels) and conditionings, each factored twice -- same bits -- and
    compared with LAPACK (tools/soak_cholesky.py is the long form)."""
    from gpemu.fit import cholesky
    rng = np.random.default_rng(2024)
    for n in list(rng.integers(1, 700, size=60)) + [64, 65, 256, 257, 1100]:
        n = int(n)
        M = rng.normal(size=(n, n))
        cond = 10.0 ** rng.uniform(0, 7)
        A = M @ M.T / n + np.eye(n) / cond
        L1, L2 = np.tril(cholesky(A)), np.tril(cholesky(A))
        assert L1.tobytes() == L2.tobytes(), n
        ref = np.linalg.cholesky(A)
        assert np.max(np.abs(L1 - ref)) / np.max(np.abs(ref)) < 1e-9 * max(1.0, cond * 1e-4), (n, cond)


def test_c3_size_fit_matches_oracle():
    """N = 1000 (BASELINE config 3): device lml/grad/factor vs the oracle at the fixed theta."""
    from gpemu.fit import DeviceFit
    model, prob, pca = GU.fixed_theta_model(1000, 500, 2, seed=0)
    X = prob["design"]
    theta = np.log(np.r_[model.gps[0].ls, model.gps[0].noise])
    fit = DeviceFit(X, kernel_kind=0, has_noise=True, jitter=1e-10)
    y = pca["Y_pca"][:, 0]
    lml, grad = fit.lml(y, theta)
    lo, go = O.lml_and_grad(X, y, theta, model.spec)
    assert abs(lml - lo) <= 1e-8 * abs(lo)
    assert np.max(np.abs(grad - go)) <= 1e-6 * max(1.0, np.max(np.abs(go)))
    L, alpha, _ = fit.factor(y, theta)
    assert relerr(L, model.gps[0].L) < 1e-9
    assert relerr(alpha, model.gps[0].alpha) < 1e-7
    fit.close()


@pytest.mark.parametrize("name", SYN + ["g3_realdata_matern15"])
def test_scaler_pca_vs_reference(name):
    """StandardScaler + PCA (device Jacobi SVD) against the reference's sklearn objects: the sign-decision
    indices bit-exact, scaler statistics to the last bits, leading components / scores / variances to 1e-9,
    and the full-rank physical-space quantity (covariance of the truncated components) to 1e-9."""
    from gpemu.fit import pca_fit
    g = GU.load(name)
    Y = GU.load("observables_fixture")["Y"] if name.startswith("g3") else g["Y"]
    k = int(g["n_pc"])
    out = pca_fit(Y)
    np.testing.assert_allclose(out["scaler_mean"], g["scaler_mean"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(out["scaler_var"], g["scaler_var"], rtol=1e-13, atol=0)
    np.testing.assert_allclose(out["scaler_scale"], g["scaler_scale"], rtol=1e-14, atol=0)
    assert np.array_equal(out["flip_argmax"][:k], g["flip_argmax"][:k])          # integer decisions: exact
    nall = g["pca_explained_variance"].shape[0]
    ev_scale = g["pca_explained_variance"][0]
    assert np.max(np.abs(out["explained_variance"][:nall] - g["pca_explained_variance"])) < 1e-12 * ev_scale
    assert relerr(out["explained_variance_ratio"][:k], g["pca_explained_variance_ratio"][:k]) < 1e-11
    assert relerr(out["components"][:k], g["pca_components"][:k]) < 1e-9
    assert relerr(out["Y_pca"][:, :k], g["Y_pca_truncated"]) < 1e-9
    # truncation covariance S_{>k} diag(ev_{>k}) S_{>k}^T (ref: emulation.py:246-249)
    cu = (out["components"][k:].T * out["explained_variance"][k:]) @ out["components"][k:]
    assert relerr(cu, g["cov_unexplained"]) < 1e-9
    # n_components truncation returns the leading block
    out5 = pca_fit(Y, n_components=k)
    np.testing.assert_array_equal(out5["components"], out["components"][:k])
    assert 2 <= out["n_sweeps"] <= 30


@pytest.mark.parametrize("name", ["g1", "g2", "g3"])
def test_scaler_pca_u_based_sign_rule_of_the_pinned_sklearn(name, monkeypatch):
    """GPEMU_SVD_FLIP=u: the sign rule of the scikit-learn the reference PINS (ref: pdm.lock:1998-1999 -> 1.3.0, whose
    PCA._fit_full decides per column of U; call site ref: emulation.py:115-117) against
    ``sklearn.utils.extmath.svd_flip(u, v, u_based_decision=True)`` on the inputs of G1-G3
    (tests/golden/make_goldens.py: golden_svd_flip_u).  The deciding ROW INDICES of U exact; components and scores (what
    the pickles hold and plot_emulation.py:74,103 reads) to 1e-9; and component by component the sign against the
    default (v-based) rule as the two sklearn rules relate."""
    from gpemu.fit import pca_fit
    g = GU.load("g_svd_flip_u")
    Y = g[name + "_Y"]
    k = g[name + "_components_u"].shape[0]
    monkeypatch.delenv("GPEMU_SVD_FLIP", raising=False)
    out_v = pca_fit(Y)
    monkeypatch.setenv("GPEMU_SVD_FLIP", "u")
    out_u = pca_fit(Y)
    assert np.array_equal(out_u["flip_argmax"][:k], g[name + "_flip_u_argmax"][:k])       # integer decisions: exact
    assert np.array_equal(out_v["flip_argmax"][:k], g[name + "_flip_v_argmax"][:k])
    assert relerr(out_u["components"][:k], g[name + "_components_u"]) < 1e-9
    assert relerr(out_u["Y_pca"][:, :k], g[name + "_Y_pca_u"]) < 1e-9
    ratio = np.sign(np.sum(out_u["components"][:k] * out_v["components"][:k], axis=1)).astype(np.int64)
    assert np.array_equal(ratio, g[name + "_u_over_v_sign"][:k])
    assert (ratio == -1).any(), "the two rules were meant to differ on these inputs"
    # physical space is invariant: the reconstruction from k components does not see the rule
    rec_u = out_u["Y_pca"][:, :k] @ out_u["components"][:k]
    rec_v = out_v["Y_pca"][:, :k] @ out_v["components"][:k]
    assert relerr(rec_u, rec_v) < 1e-12
    np.testing.assert_array_equal(out_u["explained_variance"], out_v["explained_variance"])


def test_concurrent_gp_fits_equal_the_sequential_loop():
    """fit_gps (k GPs x (1 + restarts) optimisations advancing in lock step, their log-marginal-likelihood
    evaluations batched into one launch chain: gpemu_fit_lml_batch) gives exactly what the sequential per-PC loop
    gives for the same numpy seed."""
    from gpemu import estimators as E
    g = GU.load("g1_matern15_noise")
    spec = GU.spec_of(g)
    X, Yc = g["design"], g["Y_pca_truncated"]
    ls0 = g["hi"] - g["lo"]
    kern = E.ARDKernel(kind=spec.kind, nu=spec.nu, length_scale=ls0,
                       length_scale_bounds=np.outer(ls0, (0.01, 100.0)), noise_level=0.1,
                       noise_level_bounds=(1e-3, 1e1))
    np.random.seed(99)
    seq = [E.GaussianProcessRegressor(kernel=kern, alpha=1e-10, n_restarts_optimizer=2).fit(X, y) for y in Yc.T]
    np.random.seed(99)
    par = E.fit_gps(X, Yc, kern, alpha=1e-10, n_restarts_optimizer=2, n_streams=6)
    np.random.seed(99)
    one = E.fit_gps(X, Yc, kern, alpha=1e-10, n_restarts_optimizer=2, n_streams=1)
    for a, b, c in zip(seq, par, one):
        np.testing.assert_array_equal(a.kernel_.theta, b.kernel_.theta)
        np.testing.assert_array_equal(a.kernel_.theta, c.kernel_.theta)
        np.testing.assert_array_equal(a.L_, b.L_)
        np.testing.assert_array_equal(a.alpha_, b.alpha_)
        assert a.log_marginal_likelihood_value_ == b.log_marginal_likelihood_value_


@pytest.mark.parametrize("handles", ["1", "2", "3"])
def test_group_fit_is_the_same_through_one_two_or_three_device_handles(handles, monkeypatch):
    """GPEMU_FIT_HANDLES: the groups of the lock-step driver evaluate through a device handle each, their launch chains on
    the device together (three by default up to N = 2048) -- or take turns on one.  Same optima bit for bit as the
    sequential loop, whatever the number."""
    from gpemu import estimators as E
    g = GU.load("g1_matern15_noise")
    spec = GU.spec_of(g)
    X, Yc = g["design"], g["Y_pca_truncated"]
    ls0 = g["hi"] - g["lo"]
    kern = E.ARDKernel(kind=spec.kind, nu=spec.nu, length_scale=ls0,
                       length_scale_bounds=np.outer(ls0, (0.01, 100.0)), noise_level=0.1,
                       noise_level_bounds=(1e-3, 1e1))
    np.random.seed(99)
    one = E.fit_gps(X, Yc, kern, alpha=1e-10, n_restarts_optimizer=4, n_streams=1)
    monkeypatch.setenv("GPEMU_FIT_HANDLES", handles)
    np.random.seed(99)
    par = E.fit_gps(X, Yc, kern, alpha=1e-10, n_restarts_optimizer=4, n_streams=3)       # several batches per group
    assert (f"{handles} groups of runs, each with its own device handle" in par[0].fit_driver_) == (handles != "1")
    for a, b in zip(one, par):
        np.testing.assert_array_equal(a.kernel_.theta, b.kernel_.theta)
        np.testing.assert_array_equal(a.L_, b.L_)
        assert a.log_marginal_likelihood_value_ == b.log_marginal_likelihood_value_


def test_batched_lml_equals_single_evaluations():
    """gpemu_fit_lml_batch: several (target, theta) pairs through ONE launch chain give, bit for bit, what the
    stand-alone evaluations give (every kernel carries the problem index, the GEMMs run batched), including a
    problem whose kernel matrix is not positive definite next to valid ones."""
    from gpemu.fit import DeviceFit, LinAlgError
    g = GU.load("g2_rbf_noise")
    fit, spec, X = _fit_for(g)
    ytr = g["Y_pca_truncated"]
    k = int(g["n_pc"])
    rng = np.random.default_rng(8)
    ys = np.stack([ytr[:, i % k] for i in range(7)])
    thetas = np.stack([g["theta"][i % k] + 0.3 * rng.normal(size=g["theta"].shape[1]) for i in range(7)])
    lml, grad, info = fit.lml_batch(ys, thetas)
    assert np.all(info == 0)
    for i in range(7):
        l1, g1 = fit.lml(ys[i], thetas[i])
        assert l1 == lml[i]
        np.testing.assert_array_equal(g1, grad[i])
    lml2, _, info2 = fit.lml_batch(ys[:3], thetas[:3], eval_gradient=False)
    np.testing.assert_array_equal(lml2, lml[:3])
    fit.close()
    # a noise-free kernel with tiny length scales on duplicated rows is singular: that problem alone is flagged
    Xd = np.vstack([X[:40], X[:40]])
    f2 = DeviceFit(Xd, kernel_kind=0, has_noise=False, jitter=0.0)
    th_ok = np.log((g["hi"] - g["lo"]) * 0.5)
    yd = np.r_[ytr[:40, 0], ytr[:40, 0]]
    with pytest.raises(LinAlgError):
        f2.lml(yd, th_ok)
    f3 = DeviceFit(X[:80], kernel_kind=0, has_noise=False, jitter=1e-6)
    l_ok, _ = f3.lml(ytr[:80, 0], th_ok)
    f3.close()
    lb, gb, ib = f2.lml_batch(np.stack([yd, yd]), np.stack([th_ok, th_ok]))
    assert np.all(ib != 0)
    f2.close()
    assert np.isfinite(l_ok)


def test_pca_rank_deficient_and_wide_matrices():
    """Block-Jacobi SVD on matrices the goldens do not cover: duplicated and constant observables (numerically rank
    deficient: columns of rounding residue), more observables than design points (the transposed work matrix), and
    sizes around the block widths.  Checked against the oracle's LAPACK path in what is well defined: singular values
    (absolute accuracy), the reconstruction Y_pca @ components, orthonormal components, sign rule indices of the
    leading components."""
    from gpemu.fit import pca_fit
    rng = np.random.default_rng(42)
    cases = []
    base = rng.normal(size=(120, 12)) @ rng.normal(size=(12, 70)) + 1e-3 * rng.normal(size=(120, 70))
    dup = np.hstack([base, base[:, :20], np.full((120, 3), 2.5)])          # 20 duplicated + 3 constant columns
    cases.append(("rank deficient", dup, 12))
    cases.append(("wide", rng.normal(size=(40, 6)) @ rng.normal(size=(6, 150)) + 1e-2 * rng.normal(size=(40, 150)), 6))
    for n in (31, 33, 65):
        cases.append((f"n={n}", rng.normal(size=(90, 5)) @ rng.normal(size=(5, n)) + 0.05 * rng.normal(size=(90, n)), 5))
    for name, Y, k in cases:
        out = pca_fit(Y)
        mean, scale = O.scaler_fit(Y)[:2]
        Ys = (Y - mean) / scale
        ref = O.pca_fit(Ys)
        nmin = min(Y.shape)
        ev_ref = ref["explained_variance"]
        assert np.max(np.abs(out["explained_variance"][:nmin] - ev_ref[:nmin])) < 1e-11 * ev_ref[0], name
        Xc = Ys - Ys.mean(axis=0)
        recon = out["Y_pca"] @ out["components"]
        assert np.max(np.abs(recon - Xc)) < 1e-10 * np.max(np.abs(Xc)), name
        # rows of components: orthonormal wherever the singular value is not rounding residue
        keep = out["explained_variance"] > 1e-20 * ev_ref[0]
        Cm = out["components"][keep]
        assert np.max(np.abs(Cm @ Cm.T - np.eye(Cm.shape[0]))) < 1e-10, name
        # sign rule: the entry picked is (one of) the largest in magnitude -- duplicated observables tie exactly -- and
        # it is positive (skl utils/extmath.py:944-952)
        for c in range(k):
            row = out["components"][c]
            j = int(out["flip_argmax"][c])
            assert abs(row[j]) >= np.max(np.abs(row)) * (1 - 1e-12) and row[j] > 0, name
        assert relerr(np.abs(out["components"][:k]), np.abs(ref["components"][:k])) < 1e-8, name
        assert 1 <= out["n_sweeps"] <= 40, name


@pytest.mark.parametrize("N", [130, 300, 700, 3300])
def test_panel_factorisation_has_the_bits_of_the_three_launch_steps(N, monkeypatch):
    """The one-launch-per-panel Cholesky (strips in registers, heads publishing through flags) and its look-ahead stream
    keep every MFMA chain's operand order and the ``C - acc`` form of the updates of the three-launch steps: factor,
    alpha, LML and gradient are bit-identical in all three variants (N = 130: one ragged panel of three blocks; 300: a
    full panel and a one-block one; 700: three panels, ragged last one; 3300: 52 tile rows, so the look-ahead stream is in
    use for the first panels).  The small case also against the
    oracle (skl _gpr.py:537-652 restated)."""
    from gpemu import synthetic
    from gpemu.fit import DeviceFit
    prob = synthetic.make_problem(N, 6, seed=3)
    X = prob["design"]
    y = prob["Y"][:, 0] - prob["Y"][:, 0].mean()
    theta = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.4, 0.02])
    results = {}
    for label, env in (("panel+lookahead", {}), ("panel", {"GPEMU_CHOL_LOOKAHEAD": "0"}),
                       ("three launches", {"GPEMU_CHOL_PANEL": "0"})):
        for key in ("GPEMU_CHOL_LOOKAHEAD", "GPEMU_CHOL_PANEL"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        fit = DeviceFit(X, kernel_kind=0, has_noise=True, jitter=1e-10)
        lml, grad = fit.lml(y, theta, eval_gradient=True)
        L, alpha, lml2 = fit.factor(y, theta)
        fit.close()
        assert lml == lml2
        results[label] = (lml, grad, L, alpha)
    ref = results["three launches"]
    for label in ("panel", "panel+lookahead"):
        got = results[label]
        assert got[0] == ref[0], label
        np.testing.assert_array_equal(got[1], ref[1], err_msg=label)
        np.testing.assert_array_equal(got[2], ref[2], err_msg=label)
        np.testing.assert_array_equal(got[3], ref[3], err_msg=label)
    if N <= 1000:
        spec = O.KernelSpec(kind=O.RBF, has_noise=True)
        lml_ref, grad_ref = O.lml_and_grad(X, y, theta, spec, jitter=1e-10)
        gp = O.gp_fit_at_theta(X, y, theta, spec, jitter=1e-10)
        assert abs(ref[0] - lml_ref) <= 1e-8 * abs(lml_ref)
        assert relerr(ref[1], grad_ref) < 1e-6
        assert relerr(ref[2], gp.L) < 1e-8


def test_panel_factorisation_in_a_batch(monkeypatch):
    """Eight problems of N = 700 through one launch chain: 88 strips, so the batch takes the one-launch-per-panel
    kernel too (blockIdx.y = problem, flags per problem).  Same bits as the three-launch steps and as eight single
    evaluations; a problem that is not positive definite is reported for that problem alone."""
    from gpemu import synthetic
    from gpemu.fit import DeviceFit
    N, nb = 700, 8
    prob = synthetic.make_problem(N, 8, seed=5)
    X = prob["design"]
    rng = np.random.default_rng(11)
    ys = np.stack([prob["Y"][:, j] - prob["Y"][:, j].mean() for j in range(nb)])
    base = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.4, 0.02])
    thetas = base + 0.3 * rng.standard_normal((nb, base.size))
    out = {}
    for label, env in (("panel", {}), ("three launches", {"GPEMU_CHOL_PANEL": "0"})):
        monkeypatch.delenv("GPEMU_CHOL_PANEL", raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        fit = DeviceFit(X, kernel_kind=0, has_noise=True, jitter=1e-10)
        lml, grad, info = fit.lml_batch(ys, thetas, eval_gradient=True)
        singles = [fit.lml(ys[z], thetas[z], eval_gradient=True) for z in range(nb)]
        fit.close()
        assert not info.any()
        for z in range(nb):
            assert lml[z] == singles[z][0]
            np.testing.assert_array_equal(grad[z], singles[z][1])
        out[label] = (lml, grad)
    np.testing.assert_array_equal(out["panel"][0], out["three launches"][0])
    np.testing.assert_array_equal(out["panel"][1], out["three launches"][1])
    # one numerically indefinite problem in the batch (length scales of 1e6 box widths, no noise, no jitter: K is the
    # all-ones matrix to 1e-12, rounding turns a pivot non-positive within a few rows): flagged alone
    monkeypatch.delenv("GPEMU_CHOL_PANEL", raising=False)
    fit = DeviceFit(X, kernel_kind=0, has_noise=True, jitter=0.0)
    th = thetas.copy()
    th[3, :-1] = np.log(1e6)
    th[3, -1] = -80.0
    lml, grad, info = fit.lml_batch(ys, th, eval_gradient=True)
    fit.close()
    assert info[3] != 0 and not np.delete(info, 3).any()
    assert np.all(np.isfinite(np.delete(lml, 3)))


def test_panel_head_placement_rule_and_bits(monkeypatch):
    """The heads of a panel sit on ONE XCD (workgroups 0, 8, 16, 24) only where the device holds many more workgroups of
    the panel kernel than can be blocked on undispatched heads (k_fit.hip: chol_heads_placement); otherwise --
    forced off, or a pretended small partition (GPEMU_CHOL_CAPACITY) with three handles alive, each with a panel launch
    in flight from its own host thread -- they stay at workgroups 0 .. 3.  Same bits either way, single and batched, and
    every evaluation completes (no expired wait)."""
    import concurrent.futures
    from gpemu import synthetic
    from gpemu.fit import DeviceFit
    N = 2300                                            # 36 tile rows: the placement applies to the first panels
    prob = synthetic.make_problem(N, 6, seed=3)
    X = prob["design"]
    y = prob["Y"][:, 0] - prob["Y"][:, 0].mean()
    theta = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.4, 0.02])
    ref = None
    for env in ({}, {"GPEMU_CHOL_HEADS_ONE_XCD": "1"}, {"GPEMU_CHOL_HEADS_ONE_XCD": "0"}, {"GPEMU_CHOL_CAPACITY": "64"}):
        for key in ("GPEMU_CHOL_HEADS_ONE_XCD", "GPEMU_CHOL_CAPACITY"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        fits = [DeviceFit(X, kernel_kind=0, has_noise=True, jitter=1e-10) for _ in range(3)]
        with concurrent.futures.ThreadPoolExecutor(3) as pool:        # three launch chains on the device together
            outs = list(pool.map(lambda f: [f.lml(y, theta, eval_gradient=True) for _ in range(3)][-1], fits))
        for f in fits:
            f.close()
        ref = ref or outs[0]
        for lml, grad in outs:
            assert lml == ref[0], env
            np.testing.assert_array_equal(grad, ref[1], err_msg=str(env))


def test_panel_wait_is_bounded(monkeypatch):
    """Every wait inside the one-launch-per-panel kernel is bounded: with a head that never publishes its inverse
    (fault injection, GPEMU_CHOL_FAULT) the strips' waits expire, every later wait gives up at once, and the evaluation
    ends with GPEMU_ERR_STATE in seconds instead of hanging the GPU; the handle stays usable afterwards."""
    import time
    from gpemu import _lib, synthetic
    from gpemu.fit import DeviceFit
    N = 700
    prob = synthetic.make_problem(N, 6, seed=3)
    X = prob["design"]
    y = prob["Y"][:, 0] - prob["Y"][:, 0].mean()
    theta = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.4, 0.02])
    fit = DeviceFit(X, kernel_kind=0, has_noise=True, jitter=1e-10)
    good = fit.lml(y, theta, eval_gradient=False)
    monkeypatch.setenv("GPEMU_CHOL_FAULT", "1")
    t0 = time.perf_counter()
    with pytest.raises(_lib.GpemuError) as err:
        fit.lml(y, theta, eval_gradient=False)
    assert time.perf_counter() - t0 < 30.0
    assert err.value.code == -4 and "poll bound" in str(err.value)          # GPEMU_ERR_STATE
    monkeypatch.delenv("GPEMU_CHOL_FAULT")
    assert fit.lml(y, theta, eval_gradient=False) == good
    fit.close()
