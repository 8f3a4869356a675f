"""Helpers to turn the committed golden .npz files into oracle models (tests only)."""
from __future__ import annotations

import os

import numpy as np

from oracle import gp_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def spec_of(g, prefix=""):
    return O.KernelSpec(kind=int(g[prefix + "kernel_kind"]), nu=float(g[prefix + "nu"]),
                        has_const=bool(g[prefix + "has_const"]), has_noise=bool(g[prefix + "has_noise"]))


def group_model(g, prefix="", design=None, use_golden_L=True):
    """GroupModel at the golden's fitted theta.  L_/alpha_ come from the golden where stored
    (use_golden_L) or are rebuilt by the oracle from theta."""
    design = g["design"] if design is None else design
    spec = spec_of(g, prefix)
    theta = g[prefix + "theta"]
    k = int(g[prefix + "n_pc"])
    ytr = g[prefix + "Y_pca_truncated"]
    stored = {int(i): j for j, i in enumerate(g[prefix + "L_index"])}
    gps = []
    for i in range(k):
        gp = O.gp_fit_at_theta(design, ytr[:, i], theta[i], spec, float(g["gpr_alpha"]))
        if use_golden_L and i in stored:
            gp.L = g[prefix + "L"][stored[i]]
            gp.alpha = g[prefix + "alpha"][i]
        gps.append(gp)
    return O.GroupModel(X_train=design, spec=spec, gps=gps,
                        components=g[prefix + "pca_components"],
                        explained_variance=g[prefix + "pca_explained_variance"],
                        scaler_mean=g[prefix + "scaler_mean"], scaler_scale=g[prefix + "scaler_scale"],
                        n_pc=k)


def g7_groups(g):
    """G7 (the reference's shipped three-group shape on its own fixture): group names, the mapping the reference's
    real sorter learned {observable: (group, slice in the merged matrix, slice in the group matrix)}, and per group
    the observable block starts inside the group matrix and the group's columns of the merged matrix."""
    names = [str(n) for n in g["group_names"]]
    mapping = {}
    for obs, grp, a, b, c, d in zip(g["map_observables"], g["map_group"], g["map_out_start"], g["map_out_stop"],
                                    g["map_grp_start"], g["map_grp_stop"]):
        mapping[str(obs)] = (str(grp), slice(int(a), int(b)), slice(int(c), int(d)))
    block_start, cols = {}, {}
    for n in names:
        mine = sorted((sg.start, sg.stop, so.start) for (grp, so, sg) in mapping.values() if grp == n)
        block_start[n] = [m[0] for m in mine] + [mine[-1][1]]
        cols[n] = np.concatenate([np.arange(m[2], m[2] + m[1] - m[0]) for m in mine])
    return names, mapping, block_start, cols


def g7_models(g, use_golden_L=True):
    return {n: group_model(g, prefix=n + "_", design=g["design"], use_golden_L=use_golden_L)
            for n in g7_groups(g)[0]}


from oracle.workloads import fixed_theta_model  # noqa: E402,F401  (kept under this name for the tests)


def device_model(model, device=0, with_cov_unexplained=True):
    """Upload an oracle GroupModel to the GPU through the C ABI (tests only)."""
    from gpemu.model import DeviceModel
    spec = model.spec
    k = model.n_pc
    return DeviceModel(
        X_train=model.X_train,
        ls=np.stack([gp.ls for gp in model.gps]),
        alpha=np.stack([gp.alpha for gp in model.gps]),
        L=np.stack([gp.L for gp in model.gps]),
        components=model.components[:k],
        scaler_mean=model.scaler_mean, scaler_scale=model.scaler_scale,
        kernel_kind=spec.kind, nu=spec.nu,
        const=np.array([gp.const for gp in model.gps]) if spec.has_const else None,
        noise=np.array([gp.noise for gp in model.gps]) if spec.has_noise else None,
        cov_unexplained=O.cov_unexplained(model) if with_cov_unexplained else None,
        device=device)
