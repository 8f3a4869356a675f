"""Drop-in replacement of ``bayesian_inference.emulation`` with the arithmetic on an MI355X.

Same public names, arguments and return shapes as the reference module
(ref: src/bayesian_inference/emulation.py): ``fit_emulators``, ``fit_emulator_group``,
``read_emulators``, ``write_emulators``, ``compute_emulator_cov_unexplained``,
``compute_emulator_group_cov_unexplained``, ``nd_block_diag``, ``SortEmulationGroupObservables``,
``predict``, ``predict_emulation_group``, ``EmulationGroupConfig``, ``EmulationConfig``.

What runs where
  * standardise + PCA, GP fit (kernel matrix, Cholesky, LML + gradient), GP predict, the
    back-projection and the covariance assembly run in libgpemu (HIP, gfx950) via ``gpemu``;
  * configuration parsing, HDF5 / pickle I/O and the observable bookkeeping are host Python, as in
    the reference.  ``data_IO`` is imported lazily from the reference package (it is untouched).
The results dict has the reference's keys; the objects in it are ``gpemu.estimators`` classes
(same attribute names as the sklearn objects, picklable without scikit-learn).

Deliberate differences (results identical): ``compute_emulator_cov_unexplained`` returns the dict it
builds (the reference forgets the ``return``, ref: emulation.py:214-224, so callers got ``None`` and
recomputed the matrix on every predict call).
"""
from __future__ import annotations

import logging
import os
import pickle
from pathlib import Path
from typing import Any

import numpy as np
import yaml

from gpemu import estimators
from gpemu.model import DeviceModel

logger = logging.getLogger(__name__)


def _data_IO():
    """The reference's data_IO module (HDF5 readers; out of scope here and left untouched)."""
    from bayesian_inference import data_IO
    return data_IO


####################################################################################################
def _rank_world():
    """(rank, world); joins the launcher's process group on first use (``gpemu.dist``)."""
    from gpemu import dist as gdist
    return gdist.rank_world()


def fit_emulators(emulation_config: "EmulationConfig") -> None:
    """PCA + GP fit for every emulation group; writes one pickle per group (ref: emulation.py:38-50).

    One process per GPU (torch.distributed initialised): the groups are independent, group i is fitted and
    written by rank i % world on its own GPU; all ranks leave together so that the next stage finds every file."""
    rank, world = _rank_world()
    for i, (name, group_config) in enumerate(emulation_config.emulation_groups_config.items()):
        if i % world != rank:
            continue
        result = fit_emulator_group(group_config)
        if result:   # an existing emulator is not overwritten (ref: emulation.py:46-48)
            write_emulators(config=group_config, output_dict=result)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def build_kernel(config) -> estimators.ARDKernel:
    """Kernel prototype in the order of ``kernels.active`` (ref: emulation.py:129-162)."""
    par = config.analysis_config['parameterization'][config.parameterization]
    lo, hi = np.array(par['min'], dtype=np.float64), np.array(par['max'], dtype=np.float64)
    kw: dict[str, Any] = {}
    kind = None
    for kernel_type, kernel_args in config.active_kernels.items():
        if kernel_type in ("matern", "rbf"):
            length_scale = hi - lo
            bounds = np.outer(length_scale, tuple(kernel_args['length_scale_bounds_factor']))
            kind = estimators.MATERN_KIND if kernel_type == "matern" else estimators.RBF_KIND
            kw.update(length_scale=length_scale, length_scale_bounds=bounds)
            if kernel_type == "matern":
                kw["nu"] = kernel_args['nu']
        elif kernel_type == "constant":
            kw.update(constant_value=kernel_args["constant_value"],
                      constant_value_bounds=kernel_args["constant_value_bounds"])
        elif kernel_type == "noise":
            kw.update(noise_level=kernel_args["args"]["noise_level"],
                      noise_level_bounds=kernel_args["args"]["noise_level_bounds"])
    if kind is None:
        raise ValueError("Must provide exactly one of 'matern', 'rbf' kernel")
    return estimators.ARDKernel(kind, **kw)


def fit_emulator_group(config: "EmulationGroupConfig") -> dict[str, Any]:
    """Standardise, PCA, fit one GP per retained PC (ref: emulation.py:53-192).  Returns the results dict
    (schema in SURVEY 8b), or {} when a pickle exists and ``force_retrain`` is off."""
    target = Path(config.emulation_outputfile)
    if target.exists():
        if not config.force_retrain:
            logger.info(f'{target} exists; keeping it (force_retrain is off)')
            return {}
        target.unlink()
        logger.info(f'{target} deleted (force_retrain)')

    io = _data_IO()
    observables = io.predictions_matrix_from_h5(config.output_dir, filename=config.observables_filename,
                                                observable_filter=config.observable_filter)
    n_keep = config.n_pc
    n_calc = config.max_n_components_to_calculate           # None: all min(N, F) components
    logger.info('Scaling + PCA on the device' + ('' if n_calc is None else f' (first {n_calc} components)') + '...')
    scaler, pca, scores = estimators.scale_and_pca(observables, n_components=n_calc)
    kept_scores = scores[:, :n_keep]
    back_projected = kept_scores @ pca.components_[:n_keep]
    logger.info(f'  {n_keep} components explain {pca.explained_variance_ratio_[:n_keep].sum():.6f} of the variance')

    design = io.design_array_from_h5(config.output_dir, filename=config.observables_filename)
    logger.info(f'Fitting {n_keep} GPs on {design.shape[0]} design points x {design.shape[1]} parameters '
                f'({config.n_restarts} restarts each)...')
    # the GPs and their restarts are independent optimisations: they run concurrently on the device
    emulators = estimators.fit_gps(design, kept_scores, build_kernel(config), alpha=config.alpha,
                                   n_restarts_optimizer=config.n_restarts, copy_X_train=False)
    for index, gp in enumerate(emulators):
        logger.info(f'  PC {index}: {gp.kernel_}')

    return {
        'PCA': {
            'Y': observables,
            'Y_pca': scores,
            'Y_pca_truncated': kept_scores,
            'Y_reconstructed_truncated': back_projected,
            'Y_reconstructed_truncated_unscaled': scaler.inverse_transform(back_projected),
            'pca': pca,
            'scaler': scaler,
        },
        'emulators': emulators,
    }


####################################################################################################
def read_emulators(config: "EmulationGroupConfig") -> dict[str, Any]:
    return pickle.loads(Path(config.emulation_outputfile).read_bytes())


def write_emulators(config: "EmulationGroupConfig", output_dict: dict[str, Any]) -> None:
    target = Path(config.emulation_outputfile)
    target.parent.mkdir(parents=True, exist_ok=True)
    # written under a private name and moved into place: a reader (or another rank) never sees half a pickle
    scratch = target.with_name(f'{target.name}.{os.getpid()}.tmp')
    scratch.write_bytes(pickle.dumps(output_dict))
    os.replace(scratch, target)


####################################################################################################
def compute_emulator_cov_unexplained(emulation_config, emulation_results) -> dict[str, np.ndarray]:
    """Truncation covariance of every group (ref: emulation.py:214-224; returned here, see module doc)."""
    if not emulation_results:
        emulation_results = emulation_config.read_all_emulator_groups()
    return {name: compute_emulator_group_cov_unexplained(cfg, emulation_results.get(name))
            for name, cfg in emulation_config.emulation_groups_config.items()}


def compute_emulator_group_cov_unexplained(emulation_group_config, emulation_group_result) -> np.ndarray:
    """S_{>k} diag(explained_variance_{>k}) S_{>k}^T (ref: emulation.py:227-251): one F x F x (n_comp - k)
    product on the device's f64 matrix cores (``gpemu_truncation_cov``), computed once per group."""
    return estimators.truncation_covariance(emulation_group_result['PCA']['pca'], emulation_group_config.n_pc)


####################################################################################################
def nd_block_diag(arrays):
    """Stack (..., r_i, c_i) blocks on the diagonal of a (..., sum r, sum c) array (ref: emulation.py:254-270)."""
    lead = np.amax(np.array([a.shape[:-2] for a in arrays]), axis=0) if arrays[0].ndim > 2 else ()
    rows = sum(a.shape[-2] for a in arrays)
    cols = sum(a.shape[-1] for a in arrays)
    out = np.zeros(tuple(lead) + (rows, cols))
    r = c = 0
    for a in arrays:
        out[..., r:r + a.shape[-2], c:c + a.shape[-1]] = a
        r += a.shape[-2]
        c += a.shape[-1]
    return out


class SortEmulationGroupObservables:
    """Mapping between the per-group matrices and the globally sorted observable order
    (ref: emulation.py:274-406).  ``emulation_group_to_observable_matrix`` is
    {observable: (group, slice in the merged matrix, slice in the group matrix)} in sorted order."""

    def __init__(self, emulation_group_to_observable_matrix, shape):
        self.emulation_group_to_observable_matrix = emulation_group_to_observable_matrix
        self.shape = tuple(shape)
        self._available_value_types = None

    @classmethod
    def learn_mapping(cls, emulation_config: "EmulationConfig") -> "SortEmulationGroupObservables":
        data_IO = _data_IO()
        prediction_key = "Prediction"
        all_observables = data_IO.read_dict_from_h5(emulation_config.output_dir, 'observables.h5')
        position = 0
        observable_slices = {}
        for key in data_IO.sorted_observable_list_from_dict(all_observables[prediction_key]):
            n_bins = all_observables[prediction_key][key]['y'].shape[0]
            observable_slices[key] = slice(position, position + n_bins)
            position += n_bins
        mapping = {}
        for group_name, group_config in emulation_config.emulation_groups_config.items():
            keys = data_IO.sorted_observable_list_from_dict(all_observables[prediction_key],
                                                            observable_filter=group_config.observable_filter)
            group_bin = 0
            for key in keys:
                sl = observable_slices[key]
                width = sl.stop - sl.start
                mapping[key] = (group_name, sl, slice(group_bin, group_bin + width))
                group_bin += width
        mapping = {k: mapping[k] for k in observable_slices}
        last = list(observable_slices)[-1]
        n_design = all_observables[prediction_key][last]['y'].shape[1]
        return cls(mapping, (n_design, observable_slices[last].stop))

    def group_layout(self, group_name):
        """(columns of this group in the merged matrix, observable block starts inside the group)."""
        entries = sorted(((sg.start, so, sg) for (g, so, sg) in self.emulation_group_to_observable_matrix.values()
                          if g == group_name), key=lambda e: e[0])
        cols = np.concatenate([np.arange(so.start, so.stop) for _, so, _ in entries])
        starts = [sg.start for _, _, sg in entries] + [entries[-1][2].stop]
        return cols, np.array(starts, dtype=np.int64)

    def convert(self, group_matrices):
        if self._available_value_types is None:
            self._available_value_types = set(vt for group in group_matrices.values() for vt in group)
        output = {}
        if "cov" in self._available_value_types:
            blocks = {}
            for _, (group_name, slice_out, slice_group) in self.emulation_group_to_observable_matrix.items():
                blocks[slice_out.start] = group_matrices[group_name]["cov"][:, slice_group, slice_group]
            output["cov"] = nd_block_diag([blocks[s] for s in sorted(blocks)])
        for value_type in self._available_value_types:
            if value_type == "cov":
                continue
            out = None
            for _, (group_name, slice_out, slice_group) in self.emulation_group_to_observable_matrix.items():
                m = group_matrices[group_name][value_type]
                if out is None:
                    out = np.zeros((m.shape[0], *self.shape[1:]))
                out[:, slice_out] = m[:, slice_group]
            output[value_type] = out
        return output


####################################################################################################
# Device models are built once per (results dict, n_pc, truncation covariance) and reused (the reference rebuilds
# nothing either: its sklearn objects live in the dict).  The cache holds at most GPEMU_MODEL_CACHE entries
# (default 4), least recently used first out; an evicted model is released as soon as no sampler uses it.
_DEVICE_MODELS: "dict[tuple[int, int], tuple[Any, np.ndarray | None, DeviceModel]]" = {}


def _model_cache_limit() -> int:
    return max(1, int(os.environ.get("GPEMU_MODEL_CACHE", "4")))


def release_device_models() -> None:
    """Forget every cached device model (their HBM is freed once nothing else refers to them)."""
    _DEVICE_MODELS.clear()


def device_model_for(results: dict[str, Any], n_pc: int, cov_unexplained: np.ndarray | None = None) -> DeviceModel:
    """The DeviceModel (GP factors, PCA, scaler resident in HBM) of one emulation group's results dict."""
    key = (id(results), int(n_pc))
    hit = _DEVICE_MODELS.get(key)
    if hit is not None and hit[0] is results:
        same_cov = (hit[1] is cov_unexplained) or (
            hit[1] is not None and cov_unexplained is not None and np.array_equal(hit[1], cov_unexplained))
        if same_cov:
            _DEVICE_MODELS[key] = _DEVICE_MODELS.pop(key)      # most recently used last
            return hit[2]
    emulators = results['emulators'][:n_pc]
    pca, scaler = results['PCA']['pca'], results['PCA']['scaler']
    k0 = emulators[0].kernel_
    given = cov_unexplained
    if cov_unexplained is None:
        cov_unexplained = estimators.truncation_covariance(pca, n_pc)
    dm = DeviceModel(
        X_train=emulators[0].X_train_,
        ls=np.stack([e.kernel_.length_scale for e in emulators]),
        alpha=np.stack([e.alpha_ for e in emulators]),
        L=np.stack([e.L_ for e in emulators]),
        components=pca.components_[:n_pc], scaler_mean=scaler.mean_, scaler_scale=scaler.scale_,
        kernel_kind=k0.kind, nu=k0.nu,
        const=np.array([e.kernel_.constant_value for e in emulators]) if k0.has_const else None,
        noise=np.array([e.kernel_.noise_level for e in emulators]) if k0.has_noise else None,
        cov_unexplained=cov_unexplained)
    _DEVICE_MODELS.pop(key, None)
    _DEVICE_MODELS[key] = (results, given, dm)
    while len(_DEVICE_MODELS) > _model_cache_limit():
        _DEVICE_MODELS.pop(next(iter(_DEVICE_MODELS)))
    return dm


def predict(parameters, emulation_config: "EmulationConfig", merge_predictions_over_groups: bool = True,
            emulation_group_results: dict[str, dict[str, Any]] | None = None,
            emulator_cov_unexplained: dict | None = None) -> dict[str, np.ndarray]:
    """{'central_value': (B,F), 'cov': (B,F,F)} over all groups (ref: emulation.py:410-462)."""
    emulation_group_results = emulation_group_results or {}
    emulator_cov_unexplained = emulator_cov_unexplained or {}
    predict_output = {}
    for group_name, group_config in emulation_config.emulation_groups_config.items():
        group_result = emulation_group_results.get(group_name)
        if group_result is None:
            group_result = read_emulators(group_config)
        cov_un = emulator_cov_unexplained[group_name] if emulator_cov_unexplained else None
        predict_output[group_name] = predict_emulation_group(parameters, group_result, group_config,
                                                             emulator_group_cov_unexplained=cov_un)
    if not merge_predictions_over_groups:
        return predict_output
    return emulation_config.sort_observables_in_matrix.convert(group_matrices=predict_output)


def predict_emulation_group(parameters, results, emulation_group_config, emulator_group_cov_unexplained=None):
    """Central values (B,F) and covariances (B,F,F) of one group (ref: emulation.py:466-548).
    The truncation covariance is divided by the number of rows passed, like the reference
    (ref: emulation.py:531-532)."""
    parameters = np.array(parameters, ndmin=2, dtype=np.float64)
    dm = device_model_for(results, emulation_group_config.n_pc, emulator_group_cov_unexplained)
    cv, cov = dm.predict_full(parameters, n_div=parameters.shape[0])
    return {'central_value': cv, 'cov': cov}


####################################################################################################
class _Base:
    """Attribute bag (the reference derives its config classes from common_base.CommonBase)."""

    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)

    def set_attribute(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)

    def __str__(self):
        body = '\n .  '.join(f'{k} = {v}' for k, v in self.__dict__.items())
        return f"[i] {self.__class__.__name__} with \n .  {body}"


_TOP_LEVEL_KEYS = ('observable_table_dir', 'observable_config_dir', 'observables_filename')


def _read_yaml(path):
    with open(path, 'r') as handle:
        return yaml.safe_load(handle)


def _check_kernels(active):
    """The reference's validity rules for the ``kernels`` block (ref: emulation.py:583-603)."""
    base = [name for name in ('matern', 'rbf') if name in active]
    assert len(base) == 1, "Must provide exactly one of 'matern', 'rbf' kernel"
    noise = active.get('noise')
    if noise is None:
        return
    assert 'type' in noise and 'args' in noise, "Noise configuration must have keys 'type' and 'args'"
    if noise['type'] != 'white':
        raise ValueError("Unsupported noise kernel")
    assert set(noise['args']) == {'noise_level', 'noise_level_bounds'}, \
        "Must provide arguments 'noise_level' and 'noise_level_bounds' for white noise kernel"


class EmulationGroupConfig(_Base):
    """Settings of one emulation group from the analysis YAML (ref: emulation.py:551-622); the attribute
    names are the reference's.  ``emulation_group_name=None`` reads an un-grouped ``emulators`` block."""

    def __init__(self, analysis_name='', parameterization='', analysis_config='', config_file='',
                 emulation_group_name: str | None = None):
        self.analysis_name, self.parameterization = analysis_name, parameterization
        self.analysis_config, self.config_file = analysis_config, config_file
        top = _read_yaml(config_file)
        for key in _TOP_LEVEL_KEYS:
            setattr(self, key, top[key])

        block = analysis_config['parameters']['emulators']
        if emulation_group_name is not None:
            block = block[emulation_group_name]
        self.force_retrain, self.n_pc = block['force_retrain'], block['n_pc']
        self.max_n_components_to_calculate = block.get('max_n_components_to_calculate')
        self.n_restarts, self.alpha = block['GPR']['n_restarts'], block['GPR']['alpha']

        kernels = block['kernels']
        self.active_kernels = {name: kernels[name] for name in kernels['active']}     # order of `active` is kept
        _check_kernels(self.active_kernels)

        wanted, unwanted = block.get('observable_list', []), block.get('observable_exclude_list', [])
        self.observable_filter = None
        if wanted or unwanted:
            self.observable_filter = _data_IO().ObservableFilter(include_list=wanted, exclude_list=unwanted)

        self.output_dir = os.path.join(top['output_dir'], f'{analysis_name}_{parameterization}')
        pickle_name = 'emulation.pkl' if emulation_group_name is None else f'emulation_group_{emulation_group_name}.pkl'
        self.emulation_outputfile = os.path.join(self.output_dir, pickle_name)


class EmulationConfig(_Base):
    """The emulation groups of one analysis (ref: emulation.py:624-709); attribute and method names kept."""

    def __init__(self, analysis_name: str, parameterization: str, config_file, analysis_config=None,
                 emulation_groups_config=None):
        self.analysis_name, self.parameterization = analysis_name, parameterization
        self.config_file = Path(config_file)
        self.analysis_config = {} if analysis_config is None else analysis_config
        self.emulation_groups_config = {} if emulation_groups_config is None else emulation_groups_config
        self.config = _read_yaml(self.config_file)
        for key in _TOP_LEVEL_KEYS:
            setattr(self, key, self.config[key])
        self.output_dir = os.path.join(self.config['output_dir'], f'{analysis_name}_{parameterization}')
        self._observable_filter = None
        self._sort_observables_in_matrix = None

    @classmethod
    def from_config_file(cls, analysis_name: str, parameterization: str, config_file, analysis_config):
        self = cls(analysis_name=analysis_name, parameterization=parameterization, config_file=config_file,
                   analysis_config=analysis_config)
        for group in analysis_config['parameters']['emulators']:
            self.emulation_groups_config[group] = EmulationGroupConfig(
                analysis_name=analysis_name, parameterization=parameterization, analysis_config=analysis_config,
                config_file=self.config_file, emulation_group_name=group)
        return self

    def _need_groups(self, what):
        if not self.emulation_groups_config:
            raise ValueError(f"Need to specify emulation groups to provide {what}")

    def read_all_emulator_groups(self):
        return {name: read_emulators(cfg) for name, cfg in self.emulation_groups_config.items()}

    @property
    def observable_filter(self):
        """Union of the groups' filters plus the global exclude list (built once)."""
        if self._observable_filter is None:
            self._need_groups("an observable filter")
            keep: list[str] = []
            drop: list[str] = self.config.get('global_observable_exclude_list', [])
            for group in self.emulation_groups_config.values():
                keep += group.observable_filter.include_list
                drop += group.observable_filter.exclude_list
            self._observable_filter = _data_IO().ObservableFilter(include_list=keep, exclude_list=drop)
        return self._observable_filter

    @property
    def sort_observables_in_matrix(self) -> SortEmulationGroupObservables:
        if self._sort_observables_in_matrix is None:
            self._need_groups("an sorting for observable group observables")
            self._sort_observables_in_matrix = SortEmulationGroupObservables.learn_mapping(self)
        return self._sort_observables_in_matrix
