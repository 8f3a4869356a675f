"""CPU tests of the whole-fit certificate the -m gpu fit tests use (tests/dropin_util.py:
certify_fit_against_reference): it must accept the optima scipy's L-BFGS-B reaches in the reference's arithmetic --
the reference's own, and the slightly different ones of a run whose path differs -- and reject a theta that is merely
close.  ref: emulation.py:169-172 -> skl _gpr.py:299-364."""
import types

import numpy as np
import pytest
import scipy.optimize

import dropin_util as DU
import golden_util as GU
from oracle import gp_oracle as O


def _fake_emulators(g, thetas):
    from gpemu import estimators as E
    spec = GU.spec_of(g)
    d = g["design"].shape[1]
    ls0 = g["hi"] - g["lo"]
    out = []
    for i, th in enumerate(thetas):
        t = np.exp(th)
        kern = E.ARDKernel(spec.kind, t[:d], np.outer(ls0, (0.01, 100.0)), nu=spec.nu,
                           noise_level=t[d] if spec.has_noise else None, noise_level_bounds=(1e-3, 10.0))
        lml = O.lml_and_grad(g["design"], g["Y_pca_truncated"][:, i], th, spec, float(g["gpr_alpha"]))[0]
        out.append(types.SimpleNamespace(kernel_=kern, log_marginal_likelihood_value_=lml))
    return out


@pytest.fixture(scope="module")
def g1():
    return GU.load("g1_rbf_noise")


def test_certificate_accepts_the_references_own_optima(g1):
    emus = _fake_emulators(g1, g1["theta"])
    agree, d = DU.certify_fit_against_reference(emus, g1["theta"], g1["lml_value"], "G1 self", g1["design"],
                                                g1["Y_pca_truncated"], float(g1["gpr_alpha"]))
    assert agree.all() and np.all(d == 0.0)


def test_certificate_accepts_an_equivalent_optimum_and_rejects_a_near_miss(g1):
    spec = GU.spec_of(g1)
    jit = float(g1["gpr_alpha"])
    bounds = _fake_emulators(g1, g1["theta"][:1])[0].kernel_.bounds
    # Another path to the same optima, as a device build takes it: L-BFGS-B from a start 0.2 away in every log
    # hyper-parameter, on evaluations that differ from the oracle's by relative 1e-13 (value) / 1e-11 (gradient).
    # (GP 1 of this golden has a flat ridge -- such a run ends 0.2 away, 1e-4 lower -- and is left out.)
    idx = [0, 2, 3, 4]
    sub = dict(g1)
    sub["Y_pca_truncated"] = g1["Y_pca_truncated"][:, idx]
    theta_ref, lml_ref = g1["theta"][idx], g1["lml_value"][idx]
    thetas = []
    for j, i in enumerate(idx):
        y = g1["Y_pca_truncated"][:, i]
        rng = np.random.default_rng(i)

        def neg(t):
            v, gr = O.lml_and_grad(g1["design"], y, t, spec, jit)
            return -v * (1 + 1e-13 * rng.standard_normal()), -gr * (1 + 1e-11 * rng.standard_normal(gr.size))
        start = np.clip(theta_ref[j] + 0.2 * (-1.0) ** np.arange(theta_ref.shape[1]), bounds[:, 0], bounds[:, 1])
        thetas.append(scipy.optimize.minimize(neg, start, method="L-BFGS-B", jac=True, bounds=bounds).x)
    thetas = np.stack(thetas)
    agree, d = DU.certify_fit_against_reference(_fake_emulators(sub, thetas), theta_ref, lml_ref, "G1 other path",
                                                g1["design"], sub["Y_pca_truncated"], jit)
    assert not agree.any() and np.all(d < 1e-4), "the other path was meant to end ~1e-5 from the reference's theta"
    # a theta 0.02 off in one free coordinate is close too -- but its LML is worse: rejected
    free = np.flatnonzero((theta_ref[0] > bounds[:, 0] + 1e-6) & (theta_ref[0] < bounds[:, 1] - 1e-6))
    off = theta_ref.copy()
    off[0, free[0]] += 0.02
    with pytest.raises(AssertionError, match="WORSE optimum"):
        DU.certify_fit_against_reference(_fake_emulators(sub, off), theta_ref, lml_ref, "G1 near miss", g1["design"],
                                         sub["Y_pca_truncated"], jit)
