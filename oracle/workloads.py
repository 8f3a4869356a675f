"""TEST INFRASTRUCTURE (like everything under oracle/): synthetic workloads built with the CPU oracle only.

Used by the tests (through tests/golden_util.py), by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` -- never by the product path.
"""
import os
import sys

import numpy as np

from . import gp_oracle as O

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesian-inference_amd")


def fixed_theta_model(N, F, k, seed=0, ls_factor=0.5, noise=0.05, jitter=1e-10, kind=O.RBF, nu=np.inf):
    """The C3-style model of SURVEY 8(d): synthetic data (the product's own generator, ``gpemu.synthetic`` -- inputs
    only), fixed hyper-parameters, scaler / PCA / Cholesky by the oracle (numpy / scipy, no sklearn).  Regenerates what
    tests/golden/g4_c3_fixed_theta.npz was produced from.  Returns (GroupModel, problem dict, pca dict)."""
    if _PKG not in sys.path:
        sys.path.insert(0, _PKG)
    from gpemu import synthetic
    prob = synthetic.make_problem(N, F, seed=seed)
    mean, scale, var = O.scaler_fit(prob["Y"])
    Ys = (prob["Y"] - mean) / scale
    pca = O.pca_fit(Ys)
    spec = O.KernelSpec(kind=kind, nu=nu, has_const=False, has_noise=True)
    ls = (prob["hi"] - prob["lo"]) * ls_factor
    theta = np.log(np.r_[ls, noise])
    gps = [O.gp_fit_at_theta(prob["design"], pca["Y_pca"][:, i], theta, spec, jitter) for i in range(k)]
    model = O.GroupModel(X_train=prob["design"], spec=spec, gps=gps, components=pca["components"],
                         explained_variance=pca["explained_variance"], scaler_mean=mean,
                         scaler_scale=scale, n_pc=k)
    return model, prob, pca
