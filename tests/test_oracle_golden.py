"""Pin the CPU oracle against golden vectors produced by the reference itself.

The goldens (tests/golden/*.npz) are outputs of /root/reference's own emulation.py /
log_posterior.py run in the build container by tests/golden/make_goldens.py.  If the oracle
reproduces them, it is a faithful restatement and the -m gpu tests may use it as the checker.
"""
import numpy as np
import pytest

import golden_util as GU
from oracle import gp_oracle as O

SYN = ["g1_rbf_noise", "g1_matern15_noise", "g1_matern25_const_noise", "g1_rbf_only", "g2_rbf_noise"]
RTOL = 1e-9   # the oracle follows the same arithmetic; observed differences are ~1e-13


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("name", SYN)
def test_scaler_and_pca_exact_decisions(name):
    g = GU.load(name)
    mean, scale, var = O.scaler_fit(g["Y"])
    np.testing.assert_allclose(mean, g["scaler_mean"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(scale, g["scaler_scale"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(var, g["scaler_var"], rtol=1e-13, atol=0)
    pca = O.pca_fit((g["Y"] - mean) / scale)
    k = int(g["n_pc"])
    # integer decisions: bit-exact (sign-decision index of every component the emulator uses)
    assert np.array_equal(pca["flip_argmax"][:k], g["flip_argmax"][:k])
    # leading components: same LAPACK driver => agreement to rounding
    assert relerr(pca["components"][:k], g["pca_components"][:k]) < 1e-10
    assert relerr(pca["explained_variance"], g["pca_explained_variance"]) < 1e-12
    assert relerr(pca["explained_variance_ratio"], g["pca_explained_variance_ratio"]) < 1e-12
    assert relerr(pca["Y_pca"][:, :k], g["Y_pca_truncated"]) < 1e-10


@pytest.mark.parametrize("name", ["g1", "g2", "g3"])
def test_pca_u_based_sign_rule_of_the_pinned_sklearn(name):
    """The oracle's u-based svd_flip (the rule of the scikit-learn 1.3.0 the reference pins, ref: pdm.lock:1998-1999)
    against sklearn's own ``svd_flip(u, v, u_based_decision=True)`` on the inputs of G1-G3 (golden_svd_flip_u)."""
    g = GU.load("g_svd_flip_u")
    Y = g[name + "_Y"]
    mean, scale, _ = O.scaler_fit(Y)
    k = g[name + "_components_u"].shape[0]
    pu = O.pca_fit((Y - mean) / scale, u_based=True)
    pv = O.pca_fit((Y - mean) / scale)
    assert np.array_equal(pu["flip_argmax"][:k], g[name + "_flip_u_argmax"][:k])
    assert np.array_equal(pv["flip_argmax"][:k], g[name + "_flip_v_argmax"][:k])
    np.testing.assert_allclose(pu["components"][:k], g[name + "_components_u"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(pu["Y_pca"][:, :k], g[name + "_Y_pca_u"], rtol=0, atol=1e-10)
    ratio = np.sign(np.sum(pu["components"][:k] * pv["components"][:k], axis=1)).astype(np.int64)
    assert np.array_equal(ratio, g[name + "_u_over_v_sign"][:k])


@pytest.mark.parametrize("name", SYN)
def test_fit_at_theta_L_alpha_lml_grad(name):
    g = GU.load(name)
    spec = GU.spec_of(g)
    X, ytr = g["design"], g["Y_pca_truncated"]
    for j, i in enumerate(g["L_index"]):
        gp = O.gp_fit_at_theta(X, ytr[:, i], g["theta"][i], spec, float(g["gpr_alpha"]))
        assert relerr(gp.L, g["L"][j]) < 1e-9
        # alpha_ = K^-1 y amplifies rounding by cond(K) (up to 1e9 for the noise-free kernel)
        assert relerr(gp.alpha, g["alpha"][i]) < (1e-5 if not spec.has_noise else 1e-8)
    for i in range(int(g["n_pc"])):
        for th, lml_key, grad_key in ((g["theta"][i], "lml_at_theta", "grad_at_theta"),
                                      (g["theta2"][i], "lml_at_theta2", "grad_at_theta2")):
            lml, grad = O.lml_and_grad(X, ytr[:, i], th, spec, float(g["gpr_alpha"]))
            assert abs(lml - g[lml_key][i]) <= 1e-8 * max(1.0, abs(g[lml_key][i]))
            scale_g = max(1.0, np.max(np.abs(g[grad_key][i])))
            assert np.max(np.abs(grad - g[grad_key][i])) <= 1e-6 * scale_g
        assert abs(g["lml_value"][i] - g["lml_at_theta"][i]) <= 1e-8 * max(1, abs(g["lml_value"][i]))


@pytest.mark.parametrize("name", SYN + ["g3_realdata_matern15"])
def test_predict_and_logposterior(name):
    g = GU.load(name)
    design = GU.load("observables_fixture")["design"] if name.startswith("g3") else None
    model = GU.group_model(g, design=design)
    Xq = g["Xq"]
    m, v = O.gp_predict_all(Xq, model)
    assert relerr(m, g["gp_mean"]) < RTOL
    assert np.max(np.abs(v - g["gp_var"])) < RTOL * max(1.0, np.max(g["gp_var"]))
    cu = O.cov_unexplained(model)
    assert relerr(cu, g["cov_unexplained"]) < RTOL
    pb = O.predict_group(Xq, model)
    assert relerr(pb["central_value"], g["batch_central_value"]) < RTOL
    nh = g["batch_cov_head"].shape[0]
    assert relerr(pb["cov"][:nh], g["batch_cov_head"]) < RTOL
    for i in range(g["single_central_value"].shape[0]):
        p1 = O.predict_group(Xq[i:i + 1], model)
        assert relerr(p1["central_value"][0], g["single_central_value"][i]) < RTOL
        if i < g["single_cov_head"].shape[0]:
            assert relerr(p1["cov"][0], g["single_cov_head"][i]) < RTOL
    lo, hi, ye, yr = g["lo"], g["hi"], g["y_exp"], g["y_err"]
    per = np.array([O.log_posterior(Xq[i], {"g": model}, lo, hi, ye, yr)[0]
                    for i in range(g["logpost_per_walker"].shape[0])])
    np.testing.assert_allclose(per, g["logpost_per_walker"], rtol=1e-8)
    np.testing.assert_allclose(O.log_posterior(Xq, {"g": model}, lo, hi, ye, yr), g["logpost_batched"], rtol=1e-8)
    mixed = O.log_posterior(g["X_mixed"], {"g": model}, lo, hi, ye, yr)
    assert np.array_equal(np.isneginf(mixed), np.isneginf(g["logpost_mixed"]))
    fin = np.isfinite(mixed)
    np.testing.assert_allclose(mixed[fin], g["logpost_mixed"][fin], rtol=1e-8)
    # the /n_samples quirk is real: batched != per-walker
    assert not np.allclose(g["logpost_batched"][:per.size], g["logpost_per_walker"], rtol=1e-6)


@pytest.mark.parametrize("name", SYN)
def test_lowrank_form_equals_exact_form(name):
    g = GU.load(name)
    model = GU.group_model(g)
    Xq = g["Xq"]
    m, v = O.gp_predict_all(Xq, model)
    st = O.lowrank_setup(model, g["y_exp"], g["y_err"], n_div=1)
    lr = np.array([O.loglik_lowrank(m[i], v[i], st) for i in range(Xq.shape[0])])
    np.testing.assert_allclose(lr, g["logpost_per_walker"], rtol=1e-9)
    stB = O.lowrank_setup(model, g["y_exp"], g["y_err"], n_div=Xq.shape[0])
    lrB = np.array([O.loglik_lowrank(m[i], v[i], stB) for i in range(Xq.shape[0])])
    np.testing.assert_allclose(lrB, g["logpost_batched"], rtol=1e-9)


def test_multigroup_merge():
    g = GU.load("g5_multigroup")
    models = {}
    for grp in ("g1", "g2"):
        models[grp] = GU.group_model(g, prefix=grp + "_")
    mapping = {"A": ("g1", slice(0, 10), slice(0, 10)),
               "B": ("g2", slice(10, 18), slice(0, 8)),
               "C": ("g1", slice(18, 30), slice(10, 22))}
    Xq = g["Xq"]
    go = {k: O.predict_group(Xq, mdl) for k, mdl in models.items()}
    merged = O.merge_groups(go, mapping, 30)
    assert relerr(merged["central_value"], g["merged_central_value"]) < RTOL
    assert relerr(merged["cov"][:2], g["merged_cov_head"]) < RTOL
    lp = O.log_posterior(Xq, models, g["lo"], g["hi"], g["y_exp"], g["y_err"], mapping)
    np.testing.assert_allclose(lp, g["logpost_batched"], rtol=1e-8)
    per = np.array([O.log_posterior(Xq[i], models, g["lo"], g["hi"], g["y_exp"], g["y_err"], mapping)[0]
                    for i in range(Xq.shape[0])])
    np.testing.assert_allclose(per, g["logpost_per_walker"], rtol=1e-8)


def test_multigroup_lowrank_blocks_equal_reference_merge():
    """Sum over groups and observable blocks of the low-rank form == the reference's merged value."""
    g = GU.load("g5_multigroup")
    Xq = g["Xq"]
    total = np.zeros(Xq.shape[0])
    for grp, cols, bs in (("g1", g["cols_g1"], [0, 10, 22]), ("g2", g["cols_g2"], [0, 8])):
        model = GU.group_model(g, prefix=grp + "_")
        m, v = O.gp_predict_all(Xq, model)
        sts = O.lowrank_setup_blocks(model, g["y_exp"][cols], g["y_err"][cols], bs, n_div=1)
        total += np.array([O.loglik_lowrank_blocks(m[i], v[i], sts) for i in range(Xq.shape[0])])
    np.testing.assert_allclose(total, g["logpost_per_walker"], rtol=1e-9)


def test_shipped_three_group_configuration_g7():
    """G7: the reference run end to end on its own fixture in the shape of its shipped analysis -- three emulation
    groups with 5 / 11 / 25 PCs, Matern-1.5 + White, merged by the real SortEmulationGroupObservables through the real
    data_IO (ref: config/jet_substructure.yaml:243-278; tests/golden/make_g7_shipped.py).  The oracle at the
    reference's fitted theta reproduces the factors, the merged prediction and the log-posterior in all three calling
    forms; the low-rank block form (what the device evaluates) equals the reference's value."""
    g = GU.load("g7_shipped_config")
    names, mapping, block_start, cols = GU.g7_groups(g)
    assert [int(g[n + "_n_pc"]) for n in names] == [5, 11, 25]
    assert int(g["map_shape"][1]) == 215 and len(mapping) == 16
    X = g["design"]
    for n in names:                                   # fit-side parity at identical theta, every PC of every group
        spec = GU.spec_of(g, n + "_")
        assert spec.kind == O.MATERN and spec.nu == 1.5 and spec.has_noise and not spec.has_const
        mean, scale, var = O.scaler_fit(g[n + "_Y"])
        np.testing.assert_allclose(mean, g[n + "_scaler_mean"], rtol=1e-14, atol=0)
        np.testing.assert_allclose(scale, g[n + "_scaler_scale"], rtol=1e-14, atol=0)
        pca = O.pca_fit((g[n + "_Y"] - mean) / scale)
        k = int(g[n + "_n_pc"])
        assert np.array_equal(pca["flip_argmax"][:k], g[n + "_flip_argmax"][:k])
        assert relerr(pca["Y_pca"][:, :k], g[n + "_Y_pca_truncated"]) < 1e-9
        for i in range(k):
            gp = O.gp_fit_at_theta(X, g[n + "_Y_pca_truncated"][:, i], g[n + "_theta"][i], spec, 1e-10)
            assert relerr(gp.alpha, g[n + "_alpha"][i]) < 1e-7
            chk = np.array([gp.L.sum(), (gp.L ** 2).sum(), np.abs(gp.L).max()])
            np.testing.assert_allclose(chk, g[n + "_L_checksum"][i], rtol=1e-9)
            lml, grad = O.lml_and_grad(X, g[n + "_Y_pca_truncated"][:, i], g[n + "_theta"][i], spec, 1e-10)
            assert abs(lml - g[n + "_lml_at_theta"][i]) <= 1e-8 * max(1.0, abs(lml))
            assert np.max(np.abs(grad - g[n + "_grad_at_theta"][i])) <= 1e-6 * max(1.0, np.max(np.abs(grad)))
    models = GU.g7_models(g)
    for n in names:
        assert relerr(O.cov_unexplained(models[n]), g[n + "_cov_unexplained"]) < RTOL
    Xq = g["Xq"]
    merged = O.merge_groups({n: O.predict_group(Xq, m) for n, m in models.items()}, mapping, 215)
    assert relerr(merged["central_value"], g["merged_central_value"]) < RTOL
    assert relerr(merged["cov"][0], g["merged_cov_first"]) < RTOL
    assert relerr(np.stack([np.diag(c) for c in merged["cov"]]), g["merged_cov_diag"]) < RTOL
    one = O.merge_groups({n: O.predict_group(Xq[:1], m) for n, m in models.items()}, mapping, 215)
    assert relerr(one["cov"][0], g["merged1_cov"]) < RTOL
    lo, hi, ye, yr = g["lo"], g["hi"], g["y_exp"], g["y_err"]
    np.testing.assert_allclose(O.log_posterior(Xq, models, lo, hi, ye, yr, mapping), g["logpost_batched"], rtol=1e-8)
    per = np.array([O.log_posterior(Xq[i], models, lo, hi, ye, yr, mapping)[0] for i in range(Xq.shape[0])])
    np.testing.assert_allclose(per, g["logpost_per_walker"], rtol=1e-8)
    mixed = O.log_posterior(g["X_mixed"], models, lo, hi, ye, yr, mapping)
    assert np.array_equal(np.isneginf(mixed), np.isneginf(g["logpost_mixed"])) and np.isneginf(mixed).sum() == 3
    fin = np.isfinite(mixed)
    np.testing.assert_allclose(mixed[fin], g["logpost_mixed"][fin], rtol=1e-8)
    # the device's form: per group, per observable block, low rank (k = 25 > 16 takes the LDS likelihood variant there)
    total = np.zeros(Xq.shape[0])
    for n in names:
        m, v = O.gp_predict_all(Xq, models[n])
        sts = O.lowrank_setup_blocks(models[n], ye[cols[n]], yr[cols[n]], block_start[n], n_div=1)
        total += np.array([O.loglik_lowrank_blocks(m[i], v[i], sts) for i in range(Xq.shape[0])])
    np.testing.assert_allclose(total, g["logpost_per_walker"], rtol=1e-9)


def test_c3_fixed_theta_golden():
    """C3 shape (N=1000, F=500, k=10): factors regenerated by the oracle from the seed."""
    g = GU.load("g4_c3_fixed_theta")
    model, prob, pca = GU.fixed_theta_model(int(g["N"]), int(g["F"]), int(g["n_pc"]), seed=int(g["seed"]))
    k = int(g["n_pc"])
    assert np.array_equal(pca["flip_argmax"][:k], g["flip_argmax"])
    assert relerr(pca["explained_variance"][:k], g["explained_variance_head"]) < 1e-11
    assert relerr(np.stack([gp.alpha[:8] for gp in model.gps]), g["alpha_head"]) < 1e-8
    Xq = g["Xq"]
    m, v = O.gp_predict_all(Xq, model)
    assert relerr(m, g["gp_mean"]) < 1e-9
    assert np.max(np.abs(v - g["gp_var"])) < 1e-9
    st = O.lowrank_setup(model, g["y_exp"], g["y_err"], n_div=1)
    lr = np.array([O.loglik_lowrank(m[i], v[i], st) for i in range(Xq.shape[0])])
    np.testing.assert_allclose(lr, g["logpost_per_walker"], rtol=1e-8)
    p1 = O.predict_group(Xq[:1], model)
    assert relerr(p1["central_value"][0], g["single_central_value"][0]) < 1e-9
    assert relerr(np.diag(p1["cov"][0]), g["single_cov_diag"][0]) < 1e-9
    assert relerr(p1["cov"][0][0], g["single_cov_row0"][0]) < 1e-9
