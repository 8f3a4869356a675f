"""-m gpu: ragged / extreme shapes through the C ABI against the oracle.

The goldens fix (N, d, F, k); this sweep draws shapes that sit on the kernels' tile edges
(N around 64/128 multiples, N < 16, d = 1 and the maximum d = 8, k = 1 and k = 16/17, F = k,
B = 1 / odd / across the 256-proposal small-batch switch) for every kernel family, with and
without the constant and noise terms.  Tolerance as in test_gpu_parity (north_star 1e-6; 1e-8 used).
"""
import numpy as np
import pytest

import golden_util as GU
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-8

# (N, d, F, k, B, kind, nu, has_const, has_noise)
SHAPES = [
    (1,   1,  1,  1,   1, O.RBF,    np.inf, False, True),
    (2,   1,  3,  2,   5, O.MATERN, 0.5,    False, True),
    (7,   2,  9,  3,   3, O.MATERN, 1.5,    True,  True),
    (15,  8, 12, 12,  17, O.RBF,    np.inf, True,  False),
    (16,  3, 20,  1,  64, O.MATERN, 2.5,    False, True),
    (63,  6, 40, 16, 255, O.RBF,    np.inf, False, True),
    (64,  6, 40, 17, 256, O.MATERN, 1.5,    False, True),
    (65,  4, 33,  5, 257, O.RBF,    np.inf, True,  True),
    (127, 5, 64,  8,   1, O.MATERN, 2.5,    True,  True),
    (128, 6, 70, 10, 129, O.RBF,    np.inf, False, True),
    (129, 7, 70, 10, 513, O.MATERN, 0.5,    False, True),
    (257, 6, 90,  6,  31, O.RBF,    np.inf, False, False),
    (300, 1, 50,  4, 300, O.MATERN, 1.5,    False, True),
    (385, 8, 21, 20,  77, O.RBF,    np.inf, True,  True),
]


def _problem(N, d, F, k, kind, nu, has_const, has_noise, seed):
    rng = np.random.default_rng(seed)
    lo = rng.uniform(-2.0, 0.0, d)
    hi = lo + rng.uniform(0.5, 3.0, d)
    design = rng.uniform(lo, hi, (N, d))
    Wm = rng.normal(size=(d, F))
    Y = np.tanh(((design - lo) / (hi - lo)) @ Wm) + 0.02 * rng.normal(size=(N, F))
    if N == 1:
        Y = Y + 0.0       # a single design point: zero variance columns -> scale 1 (skl rule)
    mean, scale, _ = O.scaler_fit(Y)
    pca = O.pca_fit((Y - mean) / scale)
    kk = min(k, pca["components"].shape[0])
    spec = O.KernelSpec(kind=kind, nu=nu, has_const=has_const, has_noise=has_noise)
    gps = []
    for i in range(kk):
        th = [np.log((hi - lo) * rng.uniform(0.3, 1.5, d))]
        if has_const:
            th.append(np.log(rng.uniform(0.1, 2.0, 1)))
        if has_noise:
            th.append(np.log(rng.uniform(1e-3, 0.1, 1)))
        gps.append(O.gp_fit_at_theta(design, pca["Y_pca"][:, i], np.concatenate(th), spec,
                                     1e-10 if has_noise else 1e-6))
    model = O.GroupModel(X_train=design, spec=spec, gps=gps, components=pca["components"],
                         explained_variance=pca["explained_variance"], scaler_mean=mean,
                         scaler_scale=scale, n_pc=kk)
    y_exp = Y[0] + 0.05
    y_err = rng.uniform(0.02, 0.2, F)
    return model, lo, hi, y_exp, y_err, rng


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "N%d_d%d_F%d_k%d_B%d_kind%d" % s[:6])
def test_shape_sweep(shape):
    N, d, F, k, B, kind, nu, has_const, has_noise = shape
    model, lo, hi, y_exp, y_err, rng = _problem(N, d, F, k, kind, nu, has_const, has_noise, seed=N * 131 + B)
    dm = GU.device_model(model)
    Xq = rng.uniform(lo, hi, (B, d))
    if B >= 3:                      # one query exactly on a training point, one far outside the design hull
        Xq[1] = model.X_train[0]
        Xq[2] = hi + 5.0 * (hi - lo)
    m, v = dm.gp_predict(Xq)
    mo, vo = O.gp_predict_all(Xq, model)
    mscale = max(np.max(np.abs(mo)), 1e-300)
    vscale = max(1.0, np.max(vo))
    assert m.shape == mo.shape and v.shape == vo.shape
    assert np.max(np.abs(m - mo)) < TOL * mscale
    assert np.max(np.abs(v - vo)) < TOL * vscale
    assert np.all(v >= 0.0)
    # full covariance API on a few rows (reference batch semantics n_div = rows passed)
    nb = min(B, 4)
    cv, cov = dm.predict_full(Xq[:nb])
    po = O.predict_group(Xq[:nb], model)
    assert np.max(np.abs(cv - po["central_value"])) < TOL * max(np.max(np.abs(po["central_value"])), 1e-300)
    assert np.max(np.abs(cov - po["cov"])) < TOL * max(np.max(np.abs(po["cov"])), 1e-300)
    # log-posterior, both forms, MCMC semantics (n_div = 1); out-of-box rows give -inf
    dm.likelihood_setup(y_exp, y_err, lo, hi, n_div=1.0)
    n_ref = min(B, 6)
    ref = np.array([O.log_posterior(Xq[i], {"g": model}, lo, hi, y_exp, y_err)[0] for i in range(n_ref)])
    for mode in (0, 1):
        lp = dm.logpost(Xq, mode=mode)
        assert lp.shape == (B,)
        assert np.array_equal(np.isneginf(lp[:n_ref]), np.isneginf(ref))
        fin = np.isfinite(ref)
        np.testing.assert_allclose(lp[:n_ref][fin], ref[fin], rtol=TOL)
        if B >= 3:
            assert np.isneginf(lp[2])
    # the two likelihood forms agree over the whole batch
    l0, l1 = dm.logpost(Xq, mode=0), dm.logpost(Xq, mode=1)
    fin = np.isfinite(l0)
    assert np.array_equal(fin, np.isfinite(l1))
    np.testing.assert_allclose(l0[fin], l1[fin], rtol=TOL)
    dm.close()


def test_empty_batch_and_all_out_of_bounds():
    model, lo, hi, y_exp, y_err, rng = _problem(40, 3, 10, 4, O.RBF, np.inf, False, True, seed=5)
    dm = GU.device_model(model)
    dm.likelihood_setup(y_exp, y_err, lo, hi, n_div=1.0)
    # an empty batch gives an empty log-posterior (ref: log_posterior.py:59,68); the predict entry
    # points refuse it like sklearn's check_array does (ValueError: 0 samples)
    assert dm.logpost(np.empty((0, 3))).shape == (0,)
    assert dm.logpost(np.empty((0, 3)), mode=1).shape == (0,)
    from gpemu._lib import GpemuError
    with pytest.raises(GpemuError):
        dm.gp_predict(np.empty((0, 3)))
    with pytest.raises(GpemuError):
        dm.predict_full(np.empty((0, 3)))
    X = np.tile(hi + 1.0, (9, 1))
    assert np.all(np.isneginf(dm.logpost(X)))
    assert np.all(np.isneginf(dm.logpost(X, mode=1)))
    # a row exactly on the box edge is outside (strict inequalities, ref: log_posterior.py:63-64)
    Xe = rng.uniform(lo, hi, (4, 3))
    Xe[0, 1] = lo[1]
    Xe[3, 2] = hi[2]
    lp = dm.logpost(Xe)
    assert np.isneginf(lp[0]) and np.isneginf(lp[3]) and np.all(np.isfinite(lp[1:3]))
    dm.close()


FIT_SHAPES = [(1, 1), (2, 3), (5, 8), (63, 2), (64, 6), (65, 6), (127, 4), (128, 6), (129, 1), (191, 6), (257, 5),
              (449, 6)]


@pytest.mark.parametrize("N,d", FIT_SHAPES)
@pytest.mark.parametrize("kind,nu,has_const", [(O.RBF, np.inf, False), (O.MATERN, 0.5, True), (O.MATERN, 2.5, False)])
def test_fit_side_shape_sweep(N, d, kind, nu, has_const):
    """Kernel matrix, blocked Cholesky, LML + gradient and the (L_, alpha_) factorisation at sizes around the
    64-row panel / 128-row padding edges against the oracle."""
    from gpemu.fit import DeviceFit, cholesky, kernel_matrix
    rng = np.random.default_rng(N * 17 + d)
    X = rng.uniform(-1.0, 1.0, (N, d))
    y = np.sin(X.sum(axis=1)) + 0.05 * rng.normal(size=N)
    spec = O.KernelSpec(kind=kind, nu=nu, has_const=has_const, has_noise=True)
    th = [np.log(rng.uniform(0.5, 2.0, d))]
    if has_const:
        th.append(np.log([0.7]))
    th.append(np.log([0.02]))
    th = np.concatenate(th)
    ls, c, nz = O.split_theta(th, d, spec)
    Kref = O.kernel_train(X, ls, spec, c, nz)
    K = kernel_matrix(X, th, spec.kind, spec.nu, spec.has_const, spec.has_noise, jitter=0.0)
    assert np.max(np.abs(K - Kref)) < 1e-13 * np.max(np.abs(Kref))
    Lref = np.linalg.cholesky(Kref)
    L = cholesky(Kref)
    assert np.max(np.abs(L - Lref)) < 1e-10 * np.max(np.abs(Lref))
    fit = DeviceFit(X, kernel_kind=kind, nu=nu, has_const=has_const, has_noise=True, jitter=1e-10)
    lml, grad = fit.lml(y, th)
    lo, go = O.lml_and_grad(X, y, th, spec)
    assert abs(lml - lo) <= 1e-8 * max(1.0, abs(lo))
    assert np.max(np.abs(grad - go)) <= 1e-6 * max(1.0, np.max(np.abs(go)))
    Lf, alpha, lml2 = fit.factor(y, th)
    gp = O.gp_fit_at_theta(X, y, th, spec, 1e-10)
    assert np.max(np.abs(Lf - gp.L)) < 1e-9 * np.max(np.abs(gp.L))
    assert np.max(np.abs(alpha - gp.alpha)) < 1e-7 * max(np.max(np.abs(gp.alpha)), 1e-300)
    assert abs(lml2 - lo) <= 1e-8 * max(1.0, abs(lo))
    fit.close()


@pytest.mark.parametrize("N,F", [(2, 2), (3, 7), (9, 4), (40, 40), (65, 33), (130, 70), (70, 130)])
def test_pca_shape_sweep(N, F):
    """Device scaler + PCA for tall, wide and square matrices: singular spectrum, reconstruction and
    the sign rule's decisions against the oracle (numpy SVD)."""
    from gpemu.fit import pca_fit
    rng = np.random.default_rng(N * 1009 + F)
    Y = rng.normal(size=(N, 3)) @ rng.normal(size=(3, F)) + 0.1 * rng.normal(size=(N, F))
    out = pca_fit(Y)
    mean, scale, var = O.scaler_fit(Y)
    np.testing.assert_allclose(out["scaler_mean"], mean, rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(out["scaler_scale"], scale, rtol=1e-13)
    ref = O.pca_fit((Y - mean) / scale)
    nc = min(N, F)
    ev = ref["explained_variance"]
    assert np.max(np.abs(out["explained_variance"] - ev)) < 1e-11 * ev[0]
    # compare only well-separated, non-null components (the last one of a centred matrix is null when N <= F)
    good = [i for i in range(nc) if ev[i] > 1e-8 * ev[0]
            and (i == 0 or ev[i - 1] - ev[i] > 1e-6 * ev[0]) and (i == nc - 1 or ev[i] - ev[i + 1] > 1e-6 * ev[0])]
    assert len(good) >= 1
    for i in good:
        assert np.max(np.abs(out["components"][i] - ref["components"][i])) < 1e-7
        assert np.max(np.abs(out["Y_pca"][:, i] - ref["Y_pca"][:, i])) < 1e-7 * max(1.0, np.max(np.abs(ref["Y_pca"][:, i])))
        assert int(out["flip_argmax"][i]) == int(ref["flip_argmax"][i])
    # scores times components reproduce the standardised matrix
    Ys = (Y - mean) / scale
    rec = out["Y_pca"] @ out["components"] + out["pca_mean"]
    assert np.max(np.abs(rec - Ys)) < 1e-10 * max(1.0, np.max(np.abs(Ys)))


@pytest.mark.parametrize("name,B", [("g2_rbf_noise", 300), ("g1_matern15_noise", 1027)])
def test_predict_full_writer_forms_agree(name, B, monkeypatch):
    """emulation.predict's covariance (ref: emulation.py:504-548) by the three device forms -- the matrix-core writer
    (default for even F <= 512, k <= 16), the VALU writer it replaced (GPEMU_PREDICT_VALU) and the CU-partitioned
    pipeline kept as a measured negative (GPEMU_PREDICT_SPLIT) -- against the oracle's per-sample loops, at batch sizes
    that are no multiple of a sample group, a row block or a pipeline chunk."""
    from gpemu import synthetic
    g = GU.load(name)
    model = GU.group_model(g)
    dm = GU.device_model(model)
    X = synthetic.make_walkers(B, seed=9, lo=g["lo"], hi=g["hi"])
    ref = O.predict_group(X[:40], model, O.cov_unexplained(model) * (40.0 / B))     # / n_samples quirk: n = B
    out = {}
    for form, env in (("mfma", {}), ("valu", {"GPEMU_PREDICT_VALU": "1"}),
                      ("pipeline", {"GPEMU_PREDICT_SPLIT": "16", "GPEMU_PREDICT_CHUNK": "128"})):
        for key in ("GPEMU_PREDICT_VALU", "GPEMU_PREDICT_SPLIT", "GPEMU_PREDICT_CHUNK"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        cv, cov = dm.predict_full(X, n_div=float(B))
        out[form] = (cv, cov)
        scale = np.max(np.abs(ref["cov"]))
        assert np.max(np.abs(cv[:40] - ref["central_value"])) < 1e-10 * np.max(np.abs(ref["central_value"])), form
        assert np.max(np.abs(cov[:40] - ref["cov"])) < 1e-10 * scale, form
        assert np.all(cov == np.swapaxes(cov, 1, 2)) or np.max(np.abs(cov - np.swapaxes(cov, 1, 2))) < 1e-13 * scale
    for form in ("valu", "pipeline"):
        np.testing.assert_allclose(out[form][0], out["mfma"][0], rtol=1e-12, atol=1e-13 * np.max(np.abs(out["mfma"][0])))
        np.testing.assert_allclose(out[form][1], out["mfma"][1], rtol=1e-11, atol=1e-13 * np.max(np.abs(out["mfma"][1])))
    dm.close()


@pytest.mark.parametrize("kind,nu", [(O.RBF, np.inf), (O.MATERN, 1.5), (O.MATERN, 2.5)])
def test_noise_free_emulator_with_length_scales_at_their_bounds(kind, nu):
    """ADVICE r4 (predict_dev.h): K_* is formed on the matrix cores from the EXPANDED distance x.q - |x|^2/2 - |q|^2/2,
    whose absolute error in the exponent grows like d (range / 2 ls)^2 eps; a noise-free emulator (alpha = 1e-10, no
    WhiteKernel) amplifies an error in k_* by |L^-1| in var = kdiag - |L^-1 k_*|^2.  Worst case by construction: one
    length scale at the reference's LOWER bound (0.01 x range: scaled coordinates up to +-50), the others at the upper
    one (100 x range: the design points nearly coincide there, K ill conditioned), queries ON training points, 1e-9 / 1e-6 /
    1e-3 of the range beside them, inside the box and five box widths outside.  Mean and variance against the oracle's
    cdist form (skl kernels.py:1553-1582, 1708-1781; _gpr.py:441-494) at the north star's 1e-6."""
    rng = np.random.default_rng(17)
    N, d, k = 200, 6, 3
    lo = np.array([0.1, 1.0, 0.0067, 0.0067, 0.0, 0.05])
    hi = np.array([0.5, 10.0, 10.0, 10.0, 1.5, 100.0])
    design = rng.uniform(lo, hi, (N, d))
    spec = O.KernelSpec(kind=kind, nu=nu, has_const=False, has_noise=False)
    gps = []
    for p in range(k):
        fac = np.full(d, 100.0)
        fac[p] = 0.01                                  # PC p: dimension p at the lower bound (cond(K) ~1e7 for RBF)
        y = np.sin(3.0 * (design[:, p] - lo[p]) / (hi[p] - lo[p])) + 0.1 * rng.normal(size=N)
        gps.append(O.gp_fit_at_theta(design, y, np.log((hi - lo) * fac), spec, 1e-10))
    F = 4
    model = O.GroupModel(X_train=design, spec=spec, gps=gps, components=np.eye(k, F), explained_variance=np.ones(k),
                         scaler_mean=np.zeros(F), scaler_scale=np.ones(F), n_pc=k)
    rows = [design[:40]]
    for eps_rel in (1e-9, 1e-6, 1e-3):
        rows.append(design[40:80] + eps_rel * (hi - lo) * rng.choice([-1.0, 1.0], (40, d)))
    rows.append(rng.uniform(lo, hi, (60, d)))
    rows.append(hi + 5.0 * (hi - lo) * rng.uniform(0.0, 1.0, (20, d)))
    Xq = np.concatenate(rows)
    dm = GU.device_model(model, with_cov_unexplained=False)
    m, v = dm.gp_predict(Xq)
    dm.close()
    mo, vo = O.gp_predict_all(Xq, model)
    em = np.max(np.abs(m - mo)) / max(np.max(np.abs(mo)), 1e-300)
    ev = np.max(np.abs(v - vo)) / max(1.0, np.max(vo))
    print(f"[noise-free, bounds, kind {kind} nu {nu}] mean rel err {em:.2e}, variance err {ev:.2e} "
          f"(cond-amplified: max |alpha| {max(np.max(np.abs(g.alpha)) for g in gps):.1e})")
    assert em < 1e-6 and ev < 1e-6
    assert np.all(v >= 0.0)
