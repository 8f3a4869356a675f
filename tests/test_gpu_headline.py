"""-m gpu parity tests at the headline sizes of BASELINE.json: the device StandardScaler + PCA at C3
(1000 x 500, against the reference-generated golden G4) and at C5 (5000 x 2000, against the oracle's LAPACK
SVD), the C5 model built end to end from DEVICE pieces (PCA, factorisation) with 10 PCs, the full-covariance
predict at F = 2000, the truncation covariance on the matrix cores, and the whole fit at C2 against G2."""
import numpy as np
import pytest

import dropin_util as DU
import golden_util as GU
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-8


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def test_pca_c3_size_vs_reference_golden():
    """ref: emulation.py:109-123 at N_design = 1000, N_obs = 500: the svd_flip decision indices of the leading
    components are bit-exact against the reference run (G4), the leading explained variances to 1e-11."""
    from gpemu import synthetic
    from gpemu.fit import pca_fit
    g = GU.load("g4_c3_fixed_theta")
    N, F, k = int(g["N"]), int(g["F"]), int(g["n_pc"])
    prob = synthetic.make_problem(N, F, seed=int(g["seed"]))
    out = pca_fit(prob["Y"])
    assert out["components"].shape == (min(N, F), F)
    np.testing.assert_array_equal(out["flip_argmax"][:k], g["flip_argmax"])
    np.testing.assert_allclose(out["explained_variance"][:k], g["explained_variance_head"], rtol=1e-11)
    # every component: the entry the sign decision picked is the largest in magnitude and positive
    rows = np.arange(out["components"].shape[0])
    picked = out["components"][rows, out["flip_argmax"]]
    assert np.all(picked > 0) and np.array_equal(np.argmax(np.abs(out["components"]), axis=1), out["flip_argmax"])
    # truncation covariance (ref: emulation.py:246-249) from the device components, on the device
    from gpemu.fit import truncation_cov
    cu = truncation_cov(out["components"], out["explained_variance"], k)
    np.testing.assert_allclose(np.diag(cu), g["cov_unexplained_diag"], rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("name", ["g1_rbf_noise", "g2_rbf_noise", "g3_realdata_matern15"])
def test_truncation_cov_vs_reference(name):
    """gpemu_truncation_cov (one MFMA GEMM) against the reference's compute_emulator_group_cov_unexplained."""
    from gpemu.fit import truncation_cov
    g = GU.load(name)
    k = int(g["n_pc"])
    cu = truncation_cov(g["pca_components"], g["pca_explained_variance"], k)
    assert relerr(cu, g["cov_unexplained"]) < 1e-12
    assert np.all(truncation_cov(g["pca_components"][:k], g["pca_explained_variance"][:k], k) == 0.0)


@pytest.fixture(scope="module")
def c5():
    """C5 (N_design = 5000, N_obs = 2000, 10 PCs): device PCA, oracle PCA, device factors -- built once."""
    from gpemu import estimators, synthetic
    from gpemu.fit import DeviceFit
    N, F, k = 5000, 2000, 10
    prob = synthetic.make_problem(N, F, seed=3)
    scaler, pca, Y_pca = estimators.scale_and_pca(prob["Y"])
    mean, scale, _ = O.scaler_fit(prob["Y"])
    ref = O.pca_fit((prob["Y"] - mean) / scale)
    theta = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.5, 0.05])
    fit = DeviceFit(prob["design"], kernel_kind=0, has_noise=True, jitter=1e-10)
    Ls, alphas = [], []
    for i in range(k):
        L, a, _ = fit.factor(Y_pca[:, i], theta)
        Ls.append(L)
        alphas.append(a)
    fit.close()
    return dict(N=N, F=F, k=k, prob=prob, scaler=scaler, pca=pca, Y_pca=Y_pca, ref=ref, mean=mean, scale=scale,
                theta=theta, Ls=Ls, alphas=alphas)


def test_pca_c5_size_vs_oracle(c5):
    """Device Jacobi SVD at 5000 x 2000 against the oracle (LAPACK gesdd + v-based svd_flip)."""
    k, pca, ref = c5["k"], c5["pca"], c5["ref"]
    np.testing.assert_allclose(c5["scaler"].mean_, c5["mean"], rtol=1e-14)
    np.testing.assert_allclose(c5["scaler"].scale_, c5["scale"], rtol=1e-13)
    np.testing.assert_array_equal(pca.flip_argmax_[:k], ref["flip_argmax"][:k])           # integer decisions: exact
    ev = ref["explained_variance"]
    assert np.max(np.abs(pca.explained_variance_ - ev)) < 1e-11 * ev[0]
    assert relerr(pca.components_[:k], ref["components"][:k]) < 1e-9
    assert relerr(c5["Y_pca"][:, :k], ref["Y_pca"][:, :k]) < 1e-9
    # all 2000 sign decisions agree wherever the leading entry is not a near-tie
    comp = ref["components"]
    srt = np.sort(np.abs(comp), axis=1)
    clear = srt[:, -1] - srt[:, -2] > 1e-6
    gap = np.minimum(np.r_[np.inf, ev[:-1] - ev[1:]], np.r_[ev[:-1] - ev[1:], np.inf]) > 1e-8 * ev[0]
    sel = clear & gap
    assert sel.sum() > 100
    np.testing.assert_array_equal(pca.flip_argmax_[sel], ref["flip_argmax"][sel])


def test_stress_c5_model_from_device_pieces_k10(c5):
    """BASELINE configs[4] with 10 PCs: device PCA -> device factorisation -> device model; predictive
    mean / variance, predict_full at F = 2000 and the log-posterior in both forms against the oracle fed the
    same (device) PCA and its OWN factorisation."""
    from gpemu import synthetic
    from gpemu.model import DeviceModel
    from gpemu.fit import truncation_cov
    k, prob, pca = c5["k"], c5["prob"], c5["pca"]
    spec = O.KernelSpec(kind=O.RBF, nu=np.inf, has_const=False, has_noise=True)
    n_oracle = 2                 # PCs factorised by the oracle as well (each is a 5000^3/3 LAPACK Cholesky)
    gps = []
    for i in range(k):
        if i < n_oracle:
            gp = O.gp_fit_at_theta(prob["design"], c5["Y_pca"][:, i], c5["theta"], spec, 1e-10)
            assert relerr(c5["Ls"][i], gp.L) < 1e-9
            assert relerr(c5["alphas"][i], gp.alpha) < 1e-6
        else:
            gp = O.GP(ls=gps[0].ls, const=gps[0].const, noise=gps[0].noise, L=c5["Ls"][i], alpha=c5["alphas"][i])
        gps.append(gp)
    model = O.GroupModel(X_train=prob["design"], spec=spec, gps=gps, components=pca.components_,
                         explained_variance=pca.explained_variance_, scaler_mean=c5["scaler"].mean_,
                         scaler_scale=c5["scaler"].scale_, n_pc=k)
    cu = truncation_cov(pca.components_, pca.explained_variance_, k)
    assert relerr(cu, O.cov_unexplained(model)) < 1e-11
    dm = DeviceModel(X_train=prob["design"], ls=np.stack([g.ls for g in gps]), alpha=np.stack(c5["alphas"]),
                     L=np.stack(c5["Ls"]), components=pca.components_[:k], scaler_mean=c5["scaler"].mean_,
                     scaler_scale=c5["scaler"].scale_, kernel_kind=0, noise=np.array([g.noise for g in gps]),
                     cov_unexplained=cu)
    X = synthetic.make_walkers(700, seed=4)
    m, v = dm.gp_predict(X)
    mo, vo = O.gp_predict_all(X[:3], model)
    assert np.max(np.abs(m[:3] - mo)) < TOL * np.max(np.abs(mo)) and np.max(np.abs(v[:3] - vo)) < TOL
    # predict_full at F = 2000: two rows, batch semantics (truncation covariance / 2)
    cv, cov = dm.predict_full(X[:2], n_div=2)
    po = O.predict_group(X[:2], model)
    assert relerr(cv, po["central_value"]) < TOL and relerr(cov, po["cov"]) < TOL
    dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    lp = dm.logpost(X)
    ref = np.array([O.log_posterior(X[i], {"g": model}, prob["lo"], prob["hi"], prob["y_exp"], prob["y_err"])[0]
                    for i in range(2)])
    np.testing.assert_allclose(lp[:2], ref, rtol=TOL)
    np.testing.assert_allclose(dm.logpost(X[:2], mode=1), ref, rtol=TOL)
    dm.close()


def test_fit_emulator_group_end_to_end_c2(tmp_path):
    """BASELINE configs[1] (N_design = 200, N_obs = 100, 5 PCs): emulation.fit_emulators through the
    reference's call signature lands on the reference's optimum (G2: same restart seed) and predicts its
    central values and covariances."""
    from bayesian_inference import emulation
    g = GU.load("g2_rbf_noise")
    DU.install_fake_data_IO(g["Y"], g["design"], g["y_exp"], g["y_err"], {})
    path, analysis = DU.write_config(tmp_path, kernels_active=("rbf", "noise"), n_pc=5, n_restarts=1)
    ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", path, analysis)
    np.random.seed(12345)
    emulation.fit_emulators(ec)
    cfg = ec.emulation_groups_config["main"]
    res = emulation.read_emulators(cfg)
    np.testing.assert_array_equal(res["PCA"]["pca"].flip_argmax_[:5], g["flip_argmax"][:5])
    assert relerr(res["PCA"]["Y_pca_truncated"], g["Y_pca_truncated"]) < 1e-9
    assert relerr(res["PCA"]["Y_reconstructed_truncated_unscaled"], g["Y_reconstructed_truncated_unscaled"]) < 1e-9
    jitter = float(g["gpr_alpha"])
    agree, dth = DU.certify_fit_against_reference(res["emulators"], g["theta"], g["lml_value"], "C2 fit (G2)", g["design"],
                                                  g["Y_pca_truncated"], jitter)
    cu = emulation.compute_emulator_group_cov_unexplained(cfg, res)
    assert relerr(cu, g["cov_unexplained"]) < 1e-9
    # per GP: where the optimiser stopped at the reference's theta, that PC's predictive mean / variance at 1e-6
    Xq = g["Xq"]
    for i, e in enumerate(res["emulators"]):
        if agree[i]:
            m, sd = e.predict(Xq, return_std=True)
            assert np.max(np.abs(m - g["gp_mean"][:, i])) < 1e-6 * max(1.0, np.max(np.abs(g["gp_mean"][:, i])))
            assert np.max(np.abs(sd ** 2 - g["gp_var"][:, i])) < 1e-6 * max(1.0, np.max(np.abs(g["gp_var"][:, i])))
    p = emulation.predict_emulation_group(Xq, res, cfg, emulator_group_cov_unexplained=cu)
    nh = g["batch_cov_head"].shape[0]
    # every GP, wherever its optimiser stopped: the reference's arithmetic AT THE DEVICE'S THETA, 1e-6
    om = DU.oracle_group_at(res["emulators"], g["design"], g["Y_pca_truncated"], g["pca_components"],
                            g["pca_explained_variance"], g["scaler_mean"], g["scaler_scale"], jitter)
    po = O.predict_group(Xq, om)
    assert relerr(p["central_value"], po["central_value"]) < 1e-6
    assert relerr(p["cov"], po["cov"]) < 1e-6
    if agree.all():                                  # ... and the reference's own outputs where the optima coincide
        assert relerr(p["central_value"], g["batch_central_value"]) < 1e-6
        assert relerr(p["cov"][:nh], g["batch_cov_head"]) < 1e-6
