#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference and scikit-learn):

    python tests/golden/make_goldens.py            # writes tests/golden/*.npz

What runs: the reference's own ``bayesian_inference.emulation`` and
``bayesian_inference.log_posterior`` modules, imported unchanged from
/root/reference/src (``fit_emulator_group``, ``predict_emulation_group``,
``compute_emulator_group_cov_unexplained``, ``predict``, ``log_posterior``).
Only arrays (inputs + expected outputs) are written; no reference code or
pickled reference objects are stored.

Two things are bypassed because they are OUTSIDE the hot path:
  * ``silx`` (HDF5 dict I/O, data_IO.py:32,232,251) is not installed; the two
    names data_IO imports from it are registered as never-called placeholders so
    that the import statement succeeds.  No arithmetic goes through them.
  * the two HDF5 readers ``fit_emulator_group`` calls (emulation.py:76,126) are
    pointed at in-memory synthetic matrices.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bayesian-inference_amd"))
sys.path.insert(0, "/root/reference/src")

# --- HDF5 I/O placeholders (never called) ------------------------------------------------------
_silx = types.ModuleType("silx")
_silx_io = types.ModuleType("silx.io")
_silx_dd = types.ModuleType("silx.io.dictdump")


def _not_on_path(*a, **k):  # pragma: no cover
    raise RuntimeError("HDF5 I/O is outside the hot path and must not be called here")


_silx_dd.dicttoh5 = _not_on_path
_silx_dd.h5todict = _not_on_path
_silx.io = _silx_io
_silx_io.dictdump = _silx_dd
sys.modules.setdefault("silx", _silx)
sys.modules.setdefault("silx.io", _silx_io)
sys.modules.setdefault("silx.io.dictdump", _silx_dd)

from bayesian_inference import data_IO, emulation, log_posterior  # noqa: E402  (the reference)
from gpemu import synthetic  # noqa: E402

warnings.filterwarnings("ignore")  # ConvergenceWarnings from the optimiser at bounds


class GroupCfg:
    """The attributes fit_emulator_group / predict_emulation_group read (emulation.py:64-172)."""

    def __init__(self, n_pc, lo, hi, active_kernels, n_restarts, alpha=1e-10, max_n=None):
        self.emulation_outputfile = "/nonexistent/emulation.pkl"
        self.force_retrain = False
        self.output_dir = "/nonexistent"
        self.observables_filename = "observables.h5"
        self.observable_filter = None
        self.max_n_components_to_calculate = max_n
        self.n_pc = n_pc
        self.parameterization = "p"
        self.analysis_config = {"parameterization": {"p": {"min": list(lo), "max": list(hi)}}}
        self.active_kernels = active_kernels
        self.n_restarts = n_restarts
        self.alpha = alpha


class TrivialSort:
    def __init__(self, name):
        self.name = name

    def convert(self, group_matrices):
        return group_matrices[self.name]


class EmuCfg:
    def __init__(self, groups, sorter):
        self.emulation_groups_config = groups
        self.sort_observables_in_matrix = sorter


KERNELS = {
    "rbf_noise": {
        "rbf": {"length_scale_bounds_factor": [0.01, 100]},
        "noise": {"type": "white", "args": {"noise_level": 0.1, "noise_level_bounds": [1e-3, 1e1]}},
    },
    "matern15_noise": {
        "matern": {"length_scale_bounds_factor": [0.01, 100], "nu": 1.5},
        "noise": {"type": "white", "args": {"noise_level": 0.1, "noise_level_bounds": [1e-3, 1e1]}},
    },
    "matern25_const_noise": {
        "matern": {"length_scale_bounds_factor": [0.01, 100], "nu": 2.5},
        "constant": {"constant_value": 1.0, "constant_value_bounds": [1e-3, 1e3]},
        "noise": {"type": "white", "args": {"noise_level": 0.1, "noise_level_bounds": [1e-3, 1e1]}},
    },
    "rbf_only": {
        "rbf": {"length_scale_bounds_factor": [0.01, 100]},
    },
}


def kernel_spec(active):
    kind = 0 if "rbf" in active else 1  # 0 = rbf, 1 = matern
    nu = active["matern"]["nu"] if kind == 1 else np.inf
    return dict(kernel_kind=np.int64(kind), nu=np.float64(nu),
                has_const=np.int64("constant" in active), has_noise=np.int64("noise" in active))


def fit_with_reference(Y, design, cfg):
    data_IO.predictions_matrix_from_h5 = lambda *a, **k: Y
    emulation.data_IO.predictions_matrix_from_h5 = data_IO.predictions_matrix_from_h5
    data_IO.design_array_from_h5 = lambda *a, **k: design
    emulation.data_IO.design_array_from_h5 = data_IO.design_array_from_h5
    return emulation.fit_emulator_group(cfg)


def pack_fit(res, cfg, full_L=None):
    """Arrays out of the reference's results dict."""
    pca, scaler, emus = res["PCA"]["pca"], res["PCA"]["scaler"], res["emulators"]
    out = dict(
        scaler_mean=scaler.mean_, scaler_scale=scaler.scale_, scaler_var=scaler.var_,
        pca_components=pca.components_, pca_explained_variance=pca.explained_variance_,
        pca_explained_variance_ratio=pca.explained_variance_ratio_, pca_mean=pca.mean_,
        flip_argmax=np.argmax(np.abs(pca.components_), axis=1).astype(np.int64),
        Y_pca_truncated=np.ascontiguousarray(res["PCA"]["Y_pca_truncated"]),
        Y_reconstructed_truncated_unscaled=res["PCA"]["Y_reconstructed_truncated_unscaled"],
        theta=np.stack([e.kernel_.theta for e in emus]),
        alpha=np.stack([e.alpha_ for e in emus]),
        lml_value=np.array([e.log_marginal_likelihood_value_ for e in emus]),
        n_pc=np.int64(cfg.n_pc),
    )
    idx = range(len(emus)) if full_L is None else full_L
    out["L_index"] = np.array(list(idx), dtype=np.int64)
    out["L"] = np.stack([emus[i].L_ for i in idx])
    out["L_checksum"] = np.array([[e.L_.sum(), (e.L_ ** 2).sum(), np.abs(e.L_).max()] for e in emus])
    # LML and gradient at the fitted theta and at a perturbed theta (identical-theta parity)
    th2 = out["theta"] + 0.1
    lml1, g1, lml2, g2 = [], [], [], []
    for e, t2 in zip(emus, th2):
        a, b = e.log_marginal_likelihood(e.kernel_.theta, eval_gradient=True)
        lml1.append(a); g1.append(b)
        a, b = e.log_marginal_likelihood(t2, eval_gradient=True)
        lml2.append(a); g2.append(b)
    out.update(lml_at_theta=np.array(lml1), grad_at_theta=np.stack(g1), theta2=th2,
               lml_at_theta2=np.array(lml2), grad_at_theta2=np.stack(g2))
    return out


def pack_predict(res, cfg, Xq, n_cov=4, n_single=8, n_single_cov=2):
    emus = res["emulators"]
    m = np.stack([e.predict(Xq, return_std=True)[0] for e in emus], axis=1)
    v = np.stack([e.predict(Xq, return_std=True)[1] ** 2 for e in emus], axis=1)
    cu = emulation.compute_emulator_group_cov_unexplained(cfg, res)
    pb = emulation.predict_emulation_group(Xq, res, cfg, emulator_group_cov_unexplained=cu)
    out = dict(Xq=Xq, gp_mean=m, gp_var=v, cov_unexplained=cu,
               batch_central_value=pb["central_value"], batch_cov_head=pb["cov"][:n_cov].copy())
    cv1, cov1 = [], []
    for i in range(n_single):
        p1 = emulation.predict_emulation_group(Xq[i:i + 1], res, cfg, emulator_group_cov_unexplained=cu)
        cv1.append(p1["central_value"][0])
        if i < n_single_cov:
            cov1.append(p1["cov"][0])
    out.update(single_central_value=np.stack(cv1), single_cov_head=np.stack(cov1))
    return out


def pack_logpost(res_by_group, emu_cfg, lo, hi, y_exp, y_err, Xq, n_single=None):
    log_posterior.initialize_pool_variables(lo, hi, emu_cfg, res_by_group,
                                            {"y": y_exp, "y_err": y_err}, None)
    n_single = Xq.shape[0] if n_single is None else n_single
    per_walker = np.array([log_posterior.log_posterior(Xq[i])[0] for i in range(n_single)])
    batched = log_posterior.log_posterior(Xq)
    # a batch with rows outside the box (strict inequality, log_posterior.py:63-64)
    Xo = Xq[:8].copy()
    Xo[1, 0] = lo[0]            # on the boundary -> outside
    Xo[3, 2] = hi[2] + 1.0      # beyond -> outside
    Xo[6, 5] = lo[5] - 1e-9
    mixed = log_posterior.log_posterior(Xo)
    return dict(y_exp=y_exp, y_err=y_err, logpost_per_walker=per_walker, logpost_batched=batched,
                X_mixed=Xo, logpost_mixed=mixed)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")


def golden_synthetic(tag, N, F, k, kern_name, n_restarts, n_query=64, full_L=None, seed=0):
    prob = synthetic.make_problem(N, F, seed=seed)
    lo, hi = prob["lo"], prob["hi"]
    active = KERNELS[kern_name]
    cfg = GroupCfg(k, lo, hi, active, n_restarts)
    np.random.seed(12345)  # restarts draw from the global RNG (sklearn _gpr.py:327)
    res = fit_with_reference(prob["Y"], prob["design"], cfg)
    Xq = synthetic.make_walkers(n_query, seed=1, lo=lo, hi=hi)
    out = dict(Y=prob["Y"], design=prob["design"], lo=lo, hi=hi, gpr_alpha=np.float64(cfg.alpha))
    out.update(kernel_spec(active))
    out.update(pack_fit(res, cfg, full_L=full_L))
    out.update(pack_predict(res, cfg, Xq))
    emu_cfg = EmuCfg({"g": cfg}, TrivialSort("g"))
    out.update(pack_logpost({"g": res}, emu_cfg, lo, hi, prob["y_exp"], prob["y_err"], Xq))
    save(f"{tag}.npz", **out)


def golden_fixed_theta(tag, N, F, k, n_query=16, seed=0):
    """C3 shape: fixed hyper-parameters, reference predict + log_posterior only.

    The results dict is assembled with the same sklearn calls as emulation.py:109-172 but with
    optimizer=None, so kernel_ == the prototype; the 80 MB of factors are regenerated by the
    tests from the seed, only the expected outputs are stored.
    """
    import sklearn.decomposition as skd
    import sklearn.gaussian_process as skg
    import sklearn.preprocessing as skp

    prob = synthetic.make_problem(N, F, seed=seed)
    lo, hi = prob["lo"], prob["hi"]
    scaler = skp.StandardScaler()
    pca = skd.PCA(svd_solver="full", whiten=False)
    Y_pca = pca.fit_transform(scaler.fit_transform(prob["Y"]))
    ls = (hi - lo) * 0.5
    kernel = skg.kernels.RBF(length_scale=ls) + skg.kernels.WhiteKernel(noise_level=0.05)
    emus = [skg.GaussianProcessRegressor(kernel=kernel, alpha=1e-10, optimizer=None,
                                         copy_X_train=False).fit(prob["design"], y)
            for y in Y_pca[:, :k].T]
    res = {"PCA": {"pca": pca, "scaler": scaler}, "emulators": emus}
    cfg = GroupCfg(k, lo, hi, {"rbf": {}, "noise": {}}, 0)
    Xq = synthetic.make_walkers(n_query, seed=1, lo=lo, hi=hi)
    m = np.stack([e.predict(Xq, return_std=True)[0] for e in emus], axis=1)
    v = np.stack([e.predict(Xq, return_std=True)[1] ** 2 for e in emus], axis=1)
    cu = emulation.compute_emulator_group_cov_unexplained(cfg, res)
    p1 = [emulation.predict_emulation_group(Xq[i:i + 1], res, cfg, emulator_group_cov_unexplained=cu)
          for i in range(4)]
    emu_cfg = EmuCfg({"g": cfg}, TrivialSort("g"))
    lp = pack_logpost({"g": res}, emu_cfg, lo, hi, prob["y_exp"], prob["y_err"], Xq)
    save(f"{tag}.npz", N=np.int64(N), F=np.int64(F), n_pc=np.int64(k), seed=np.int64(seed),
         length_scale=ls, noise_level=np.float64(0.05), gpr_alpha=np.float64(1e-10),
         Xq=Xq, gp_mean=m, gp_var=v,
         single_central_value=np.stack([p["central_value"][0] for p in p1]),
         single_cov_diag=np.stack([np.diag(p["cov"][0]) for p in p1]),
         single_cov_row0=np.stack([p["cov"][0][0] for p in p1]),
         cov_unexplained_diag=np.diag(cu).copy(),
         flip_argmax=np.argmax(np.abs(pca.components_), axis=1).astype(np.int64)[:k],
         explained_variance_head=pca.explained_variance_[:k],
         alpha_head=np.stack([e.alpha_[:8] for e in emus]),
         **lp)


def golden_multigroup(tag, N=60, F=30, seed=3):
    """Two emulation groups whose observables interleave in the sorted order (emulation.py:346-406)."""
    prob = synthetic.make_problem(N, F, seed=seed)
    lo, hi = prob["lo"], prob["hi"]
    # observables A: bins 0-9, B: 10-17, C: 18-29; group g1 = {A, C}, g2 = {B}
    cols = {"g1": np.r_[0:10, 18:30], "g2": np.r_[10:18]}
    mapping = {
        "A": ("g1", slice(0, 10), slice(0, 10)),
        "B": ("g2", slice(10, 18), slice(0, 8)),
        "C": ("g1", slice(18, 30), slice(10, 22)),
    }
    sorter = emulation.SortEmulationGroupObservables(emulation_group_to_observable_matrix=mapping,
                                                     shape=(N, F))
    cfgs = {"g1": GroupCfg(4, lo, hi, KERNELS["matern15_noise"], 1),
            "g2": GroupCfg(3, lo, hi, KERNELS["rbf_noise"], 1)}
    np.random.seed(777)
    res = {g: fit_with_reference(np.ascontiguousarray(prob["Y"][:, cols[g]]), prob["design"], cfgs[g])
           for g in cfgs}
    Xq = synthetic.make_walkers(16, seed=1, lo=lo, hi=hi)
    emu_cfg = EmuCfg(cfgs, sorter)
    merged = emulation.predict(Xq, emu_cfg, emulation_group_results=res)
    merged1 = emulation.predict(Xq[:1], emu_cfg, emulation_group_results=res)
    out = dict(Y=prob["Y"], design=prob["design"], lo=lo, hi=hi, Xq=Xq, gpr_alpha=np.float64(1e-10),
               cols_g1=cols["g1"].astype(np.int64), cols_g2=cols["g2"].astype(np.int64),
               merged_central_value=merged["central_value"], merged_cov_head=merged["cov"][:2].copy(),
               merged1_central_value=merged1["central_value"], merged1_cov=merged1["cov"][0])
    for g in cfgs:
        spec = kernel_spec(cfgs[g].active_kernels)
        fit = pack_fit(res[g], cfgs[g])
        for kk, vv in {**spec, **fit}.items():
            out[f"{g}_{kk}"] = vv
    out.update(pack_logpost(res, emu_cfg, lo, hi, prob["y_exp"], prob["y_err"], Xq))
    save(f"{tag}.npz", **out)


def golden_realdata(tag):
    """Real JETSCAPE fixture (ref: tests/test_data/observables.h5) dumped to npz by
    tests/golden/dump_observables_h5.py (needs h5py; run with /opt/conda/bin/python3.9)."""
    src = os.path.join(HERE, "observables_fixture.npz")
    if not os.path.exists(src):
        print("skip real-data golden: run dump_observables_h5.py first")
        return
    fx = np.load(src)
    Y, design, y_exp, y_err = fx["Y"], fx["design"], fx["y"], fx["y_err"]
    # the shipped box for this 6-parameter design (ref: config/jet_substructure.yaml:130-131)
    lo, hi = synthetic.BOX_LO.copy(), synthetic.BOX_HI.copy()
    lo = np.minimum(lo, design.min(0) - 1e-6)
    hi = np.maximum(hi, design.max(0) + 1e-6)
    cfg = GroupCfg(10, lo, hi, KERNELS["matern15_noise"], 0)
    np.random.seed(4242)
    res = fit_with_reference(Y, design, cfg)
    Xq = synthetic.make_walkers(32, seed=1, lo=design.min(0), hi=design.max(0))
    out = dict(lo=lo, hi=hi, gpr_alpha=np.float64(cfg.alpha))
    out.update(kernel_spec(cfg.active_kernels))
    out.update(pack_fit(res, cfg, full_L=[0]))
    out.update(pack_predict(res, cfg, Xq, n_cov=1, n_single=4, n_single_cov=1))
    emu_cfg = EmuCfg({"g": cfg}, TrivialSort("g"))
    out.update(pack_logpost({"g": res}, emu_cfg, lo, hi, y_exp, y_err, Xq))
    save(f"{tag}.npz", **out)


def golden_svd_flip_u(tag="g_svd_flip_u"):
    """The sign rule of the scikit-learn the reference PINS (ref: pdm.lock:1998-1999 -> 1.3.0): its PCA._fit_full calls
    ``svd_flip(U, Vt)`` with the u-based decision (per COLUMN of U the sign of its max-|.| entry), where 1.5+ -- the
    version the other goldens were made with -- decides on the rows of Vt.  The installed 1.7.2 still exports that
    rule: ``sklearn.utils.extmath.svd_flip(u, v, u_based_decision=True)``.  Same steps as ``_fit_full`` otherwise
    (ref: emulation.py:109-117): StandardScaler, centre, LAPACK gesdd thin SVD.  Inputs: those of G1, G2 and G3."""
    import scipy.linalg
    import sklearn.preprocessing as skp
    from sklearn.utils.extmath import svd_flip
    out = {}
    inputs = {"g1": synthetic.make_problem(50, 30, seed=0)["Y"], "g2": synthetic.make_problem(200, 100, seed=0)["Y"]}
    src = os.path.join(HERE, "observables_fixture.npz")
    if os.path.exists(src):
        inputs["g3"] = np.load(src)["Y"]
    for name, Y in inputs.items():
        Ys = skp.StandardScaler().fit_transform(Y)
        Xc = Ys - Ys.mean(axis=0)
        U, S, Vt = scipy.linalg.svd(Xc, full_matrices=False)
        Uu, Vu = svd_flip(U.copy(), Vt.copy(), u_based_decision=True)
        Uv, Vv = svd_flip(U.copy(), Vt.copy(), u_based_decision=False)
        k = 10
        out[f"{name}_Y"] = Y
        out[f"{name}_flip_u_argmax"] = np.argmax(np.abs(U), axis=0).astype(np.int64)        # row of U that decides
        out[f"{name}_flip_v_argmax"] = np.argmax(np.abs(Vt), axis=1).astype(np.int64)
        out[f"{name}_u_over_v_sign"] = np.sign(np.sum(Vu * Vv, axis=1)).astype(np.int64)    # +1: both rules agree
        out[f"{name}_components_u"] = Vu[:k]
        out[f"{name}_Y_pca_u"] = (Uu * S)[:, :k]
        out[f"{name}_explained_variance"] = (S ** 2 / (Y.shape[0] - 1))[:k]
    save(f"{tag}.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "flipu"]
    if "flipu" in which:
        golden_svd_flip_u()
    if "g1" in which:
        golden_synthetic("g1_rbf_noise", 50, 30, 5, "rbf_noise", 2)
        golden_synthetic("g1_matern15_noise", 50, 30, 5, "matern15_noise", 2)
        golden_synthetic("g1_matern25_const_noise", 50, 30, 5, "matern25_const_noise", 1)
        golden_synthetic("g1_rbf_only", 50, 30, 5, "rbf_only", 1)
    if "g2" in which:
        golden_synthetic("g2_rbf_noise", 200, 100, 5, "rbf_noise", 1, full_L=[0, 1])
    if "g3" in which:
        golden_realdata("g3_realdata_matern15")
    if "g4" in which:
        golden_fixed_theta("g4_c3_fixed_theta", 1000, 500, 10)
    if "g5" in which:
        golden_multigroup("g5_multigroup")
