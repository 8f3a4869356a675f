"""Test helpers for the drop-in modules: a fake ``data_IO`` (the reference's HDF5 layer is out of
scope and needs silx, which is not installed) and a config written to a temp dir."""
import os
import sys
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))


class FakeObservableFilter:
    def __init__(self, include_list, exclude_list=None):
        self.include_list = list(include_list)
        self.exclude_list = list(exclude_list or [])

    def accept_observable(self, observable_name):
        return True


def install_fake_data_IO(Y, design, y_exp, y_err, written):
    """Register a stand-in ``bayesian_inference.data_IO`` serving in-memory matrices; ``written``
    collects what ``write_dict_to_h5`` receives."""
    import bayesian_inference
    m = types.ModuleType("bayesian_inference.data_IO")
    m.predictions_matrix_from_h5 = lambda *a, **k: Y
    m.design_array_from_h5 = lambda *a, **k: design

    def data_array_from_h5(output_dir, filename, pseudodata_index=-1, observable_filter=None):
        if pseudodata_index < 0:
            return {"y": y_exp, "y_err": y_err}
        # closure test: a validation point's prediction smeared with the experimental uncertainty
        # (ref: data_IO.py:362-372), drawn from numpy's global state like the reference
        centre = Y[pseudodata_index % Y.shape[0]]
        return {"y": centre + np.random.normal(loc=0.0, scale=y_err), "y_err": y_err}
    m.data_array_from_h5 = data_array_from_h5
    m.ObservableFilter = FakeObservableFilter

    def write_dict_to_h5(results, output_dir, filename, verbose=True):
        # what the reference's data_IO does (dicttoh5), through the silx-free writer; the dict is kept for the checks
        from gpemu import h5io
        written[os.path.join(output_dir, filename)] = results
        h5io.write_dict_to_h5(results, output_dir, filename, verbose=verbose)
    m.write_dict_to_h5 = write_dict_to_h5
    sys.modules["bayesian_inference.data_IO"] = m
    bayesian_inference.data_IO = m
    return m


def write_config(tmp_path, kernels_active=("rbf", "noise"), n_pc=5, n_restarts=1):
    cfg = yaml.safe_load(open(os.path.join(HERE, "fixtures", "analysis.yaml")))
    cfg["output_dir"] = str(tmp_path / "out")
    em = cfg["test_analysis"]["parameters"]["emulators"]["main"]
    em["kernels"]["active"] = list(kernels_active)
    em["n_pc"] = n_pc
    em["GPR"]["n_restarts"] = n_restarts
    path = tmp_path / "analysis.yaml"
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return str(path), cfg["test_analysis"]


class TrivialSort:
    """Single group whose matrix is the merged matrix."""

    def __init__(self, name):
        self.name = name

    def convert(self, group_matrices):
        return group_matrices[self.name]


def results_at_golden_theta(g, kernels_active=None, design=None):
    """Results dict in the reference's schema with our estimator objects at the golden's fitted theta."""
    from gpemu import estimators as E
    design = g["design"] if design is None else design
    import golden_util as GU
    spec = GU.spec_of(g)
    k = int(g["n_pc"])
    d = design.shape[1]
    scaler = E.StandardScaler()
    scaler.mean_, scaler.scale_, scaler.var_ = g["scaler_mean"], g["scaler_scale"], g["scaler_var"]
    pca = E.PCA()
    pca.components_, pca.explained_variance_ = g["pca_components"], g["pca_explained_variance"]
    pca.explained_variance_ratio_ = g["pca_explained_variance_ratio"]
    pca.mean_ = g["pca_mean"] if "pca_mean" in g else np.zeros(g["pca_components"].shape[1])
    emus = []
    for i in range(k):
        th = np.exp(g["theta"][i])
        kern = E.ARDKernel(spec.kind, th[:d], np.outer(th[:d], [0.01, 100]), nu=spec.nu,
                           constant_value=th[d] if spec.has_const else None, constant_value_bounds=(1e-3, 1e3),
                           noise_level=th[d + int(spec.has_const)] if spec.has_noise else None,
                           noise_level_bounds=(1e-3, 10))
        gp = E.GaussianProcessRegressor(kern, alpha=float(g["gpr_alpha"]), optimizer=None, copy_X_train=False)
        gp.fit(design, g["Y_pca_truncated"][:, i])
        emus.append(gp)
    return {"PCA": {"pca": pca, "scaler": scaler, "Y_pca_truncated": g["Y_pca_truncated"]}, "emulators": emus}


