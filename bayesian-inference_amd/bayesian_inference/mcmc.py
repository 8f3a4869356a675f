"""Drop-in replacement of ``bayesian_inference.mcmc`` (ref: src/bayesian_inference/mcmc.py).

Public surface kept: ``run_mcmc(config, closure_index=-1)``, ``credible_interval``, ``map_parameters``,
``LoggingEnsembleSampler`` and ``MCMCConfig`` (same attribute names), and the on-disk results -- ``mcmc.h5``
with ``chain``, ``acceptance_fraction``, ``log_prob``, ``autocorrelation_time`` (+ the closure extras) and the
pickled sampler.  What changes is where the work happens: the ensemble, the stretch move and the
log-posterior live on the GPU(s) (``gpemu.sampler.EnsembleSampler``) instead of an emcee sampler feeding a
multiprocessing pool (ref: mcmc.py:77-85).  One process per GPU under ``torch.distributed``: the production
chain shards its walkers over the ranks, closure chains run whole, one per rank.
"""
from __future__ import annotations

import logging
import os
import pickle
from pathlib import Path

import numpy as np
import yaml

from bayesian_inference import emulation, log_posterior
from gpemu import dist as gdist
from gpemu.sampler import EnsembleSampler, walkers_independent

logger = logging.getLogger(__name__)

_MCMC_KEYS = ("n_walkers", "n_burn_steps", "n_sampling_steps", "n_logging_steps")


def _data_IO():
    from bayesian_inference import data_IO
    return data_IO


def _rank_world():
    """(rank, world); joins the launcher's process group on first use (the reference's steering script
    initialises none)."""
    return gdist.rank_world()


def _rank():
    return _rank_world()[0]


def closure_owner(closure_index, world):
    """Closure tests (ref: steer_analysis.py:168-183) are independent chains, one per validation design
    point: with one process per GPU every rank walks the caller's loop, rank ``closure_index % world``
    runs that chain whole on its GPU and writes its own ``closure/results/<index>/`` files, the others skip
    it -- N chains in flight on N GPUs, no communication.  GPEMU_CLOSURE_REPLICAS=0 restores walker
    sharding for closure runs too."""
    if closure_index < 0 or world <= 1 or os.environ.get("GPEMU_CLOSURE_REPLICAS", "1") == "0":
        return None
    return closure_index % world


def _same_on_all_ranks(arr):
    """Rank 0's copy of ``arr`` on every rank (identity without torch.distributed)."""
    return gdist.rank0_array(arr)


def _best_distinct(sampler, count):
    """Positions of the ``count`` highest *distinct* log-probabilities seen so far (the reference restarts the
    second burn-in stage from them, ref: mcmc.py:99)."""
    _, first_seen = np.unique(sampler.flatlnprobability, return_index=True)   # ascending in log-probability
    return sampler.flatchain[first_seen[-count:]]


####################################################################################################
# Closure tests as ONE batched run.  The reference's loop (ref: steer_analysis.py:168-183) calls
# run_mcmc(config, closure_index=i) once per validation design point; the chains are independent and share the
# emulators.  The first call a rank receives runs ALL of that rank's closure chains stacked in one multi-chain
# device sampler and writes every chain's files; the later calls find their chain done and return.
#   * only the call for the FIRST index this rank owns starts a batch (the reference's loop starts at 0); it always
#     reruns, so a second closure pass in the same process -- e.g. after re-fitting the emulators -- produces fresh
#     files like the reference's; a call for any other index that no batch has produced runs that chain alone;
#   * a finished index is handed out once: asking for the same index again reruns it;
#   * the batch is keyed on the emulator files' modification times as well, so chains of older emulators never count.
_closure_done: "dict[tuple, set]" = {}


def _closure_key(config):
    stamps = []
    try:
        for group in config.analysis_config['parameters']['emulators']:
            path = os.path.join(config.output_dir, f'emulation_group_{group}.pkl')
            stamps.append((path, os.path.getmtime(path) if os.path.exists(path) else None))
    except (KeyError, TypeError):
        pass
    return (config.output_dir, config.analysis_name, config.parameterization, tuple(stamps))


def _closure_sub_batches(config, indices, n_par):
    """Split ``indices`` so that one stacked run's chain storage (C x steps x W x (d + 1) doubles on the device, and
    again on the host) stays within GPEMU_CLOSURE_CHAIN_GIB (default 16)."""
    budget = float(os.environ.get("GPEMU_CLOSURE_CHAIN_GIB", "16")) * 2 ** 30
    steps = max(config.n_sampling_steps, config.n_burn_steps)
    per_chain = 8.0 * steps * config.n_walkers * (n_par + 1)
    n = max(1, int(budget // max(per_chain, 1.0)))
    return [indices[i:i + n] for i in range(0, len(indices), n)]


def _closure_batch_enabled():
    return os.environ.get("GPEMU_CLOSURE_BATCH", "1") != "0"


def _closure_indices(config, first, rank, world):
    """Closure indices this rank still has to run, from ``first`` on (validation_indices of the analysis)."""
    lo, hi = config.analysis_config['validation_indices']
    return [j for j in range(first, hi - lo) if closure_owner(j, world) in (None, rank)]


def _run_closure_batch(config, indices):
    """The chains of ``indices`` (ref: mcmc.py:34-134 each) as one stacked run: per chain the reference's pseudo-data
    draw and start positions (numpy's global state, in the reference's order), its two-stage burn-in with the
    restart from its own best points, production, and its own ``closure/results/<index>/`` outputs."""
    from gpemu.sampler import DeviceSampler
    box = config.analysis_config['parameterization'][config.parameterization]
    lower, upper = box['min'], box['max']
    n_par, n_walk, n_ch = len(box['names']), config.n_walkers, len(indices)
    emu_cfg = emulation.EmulationConfig.from_config_file(
        analysis_name=config.analysis_name, parameterization=config.parameterization,
        analysis_config=config.analysis_config, config_file=config.config_file)
    emu_results = emu_cfg.read_all_emulator_groups()
    truncation_cov = emulation.compute_emulator_cov_unexplained(emu_cfg, emu_results)
    io = _data_IO()
    datas, starts = [], []
    for j in indices:           # the reference's order of draws: pseudo-data of chain j, then its start positions
        datas.append(io.data_array_from_h5(config.output_dir, 'observables.h5', pseudodata_index=j,
                                           observable_filter=emu_cfg.observable_filter))
        starts.append(np.random.uniform(lower, upper, (n_walk, n_par)))
    y_err = np.asarray(datas[0]['y_err'], dtype=np.float64)
    for dat in datas[1:]:
        if not np.array_equal(np.asarray(dat['y_err'], dtype=np.float64), y_err):
            raise ValueError("closure chains must share the experimental uncertainties to be stacked")
    log_posterior.initialize_pool_variables(lower, upper, emu_cfg, emu_results, datas[0], truncation_cov)
    models = log_posterior.device_models_for_chains(np.stack([np.asarray(dat['y'], dtype=np.float64) for dat in datas]))
    seeds = [int(np.random.randint(0, 2 ** 31 - 1)) for _ in indices]
    sampler = DeviceSampler(models, n_walk, seeds=seeds)
    logger.info(f'Closure tests {indices[0]}..{indices[-1]}: {n_ch} chains x {n_walk} walkers stacked in one sampler')

    def per_chain(arr):         # (steps, C W, ...) -> list of (steps, W, ...)
        return [arr[:, c * n_walk:(c + 1) * n_walk] for c in range(n_ch)]

    def advance(X0, steps):
        # emcee's initial-state checks, per chain (the single-chain path makes them in EnsembleSampler.advance)
        for c, x0 in enumerate(X0):
            if not walkers_independent(x0):
                raise ValueError(f"closure chain {indices[c]}: Initial state has a large condition number. Make sure "
                                 "that your walkers are linearly independent for the best performance")
        sampler.set_state(np.concatenate(X0))
        if np.any(np.isnan(sampler.get_state()[1])):
            raise ValueError("The initial log_prob was NaN")
        done = 0
        while done < steps:
            block = min(config.n_logging_steps - done % config.n_logging_steps, steps - done)
            sampler.run(block)
            done += block
            if done % config.n_logging_steps == 0 or done == steps:
                nacc, it, _ = sampler.counts()
                frac = nacc / float(max(it, 1))
                logger.info(f'  step {done}: acceptance fraction over {n_ch} chains: mean {frac.mean()}, '
                            f'min {frac.min()}, max {frac.max()}')

    first_stage = config.n_burn_steps // 2
    advance(starts, first_stage)
    chain, lps = sampler.get_chain()
    restart = []
    for c, (ch, lp) in enumerate(zip(per_chain(chain), per_chain(lps))):     # ref: mcmc.py:99, per chain
        _, first_seen = np.unique(lp.reshape(-1), return_index=True)
        restart.append(ch.reshape(-1, n_par)[first_seen[-n_walk:]])
    sampler.reset()
    advance(restart, config.n_burn_steps - first_stage)
    state = sampler.get_state()[0]
    sampler.reset()
    advance([state[c * n_walk:(c + 1) * n_walk] for c in range(n_ch)], config.n_sampling_steps)
    chain, lps = sampler.get_chain()
    nacc, iters, _ = sampler.counts()
    # every chain's autocorrelation time from the chain as it sits on the device (its own walkers only)
    taus = []
    for c in range(n_ch):
        try:
            taus.append(sampler.integrated_time(w0=c * n_walk, nw=n_walk))
        except Exception as err:
            logger.info(f'No autocorrelation time (closure {indices[c]}): {err}')
            taus.append(None)
    sampler.close()

    validation_design = io.design_array_from_h5(config.output_dir, filename='observables.h5', validation_set=True)
    for c, j in enumerate(indices):
        cfg_j = MCMCConfig(analysis_name=config.analysis_name, parameterization=config.parameterization,
                           analysis_config=config.analysis_config, config_file=config.config_file, closure_index=j)
        one = LoggingEnsembleSampler(n_walk, n_par, log_posterior.log_posterior, seed=seeds[c], sharded=False)
        one._cache = (per_chain(chain)[c].copy(), per_chain(lps)[c].copy(),
                      nacc[c * n_walk:(c + 1) * n_walk].copy(), iters)
        one._frozen = True
        tau = taus[c]
        results = {'chain': one.get_chain(), 'acceptance_fraction': one.acceptance_fraction,
                   'log_prob': one.get_log_prob(), 'autocorrelation_time': tau,
                   'design_point': validation_design[j], 'experimental_pseudodata': datas[c]}
        logger.info(f'Writing {cfg_j.mcmc_outputfile}')
        io.write_dict_to_h5(results, cfg_j.mcmc_output_dir, 'mcmc.h5', verbose=True)
        pickle_path = Path(cfg_j.sampler_outputfile)
        pickle_path.parent.mkdir(parents=True, exist_ok=True)
        pickle_path.write_bytes(pickle.dumps(one))


def run_mcmc(config, closure_index=-1):
    """Calibrate the parameters against the data (or, for ``closure_index >= 0``, against the pseudo-data
    of that validation point) with the affine-invariant ensemble sampler (ref: mcmc.py:34-134)."""
    rank, world = _rank_world()
    owner = closure_owner(closure_index, world)
    if owner is not None and owner != rank:
        logger.info(f'closure test {closure_index}: runs on rank {owner}')
        return
    alone = owner is not None          # this rank runs the whole chain by itself
    if closure_index >= 0 and (alone or world == 1) and _closure_batch_enabled() \
            and 'validation_indices' in config.analysis_config:
        key = _closure_key(config)
        done = _closure_done.setdefault(key, set())
        owned = _closure_indices(config, 0, rank, world)
        if owned and closure_index == owned[0] and log_posterior.chains_can_stack(config):
            for stale in [k for k in _closure_done if k[:3] == key[:3] and k != key]:
                del _closure_done[stale]
            done.clear()                      # a new pass: everything reruns, like the reference
            n_par = len(config.analysis_config['parameterization'][config.parameterization]['names'])
            for batch in _closure_sub_batches(config, owned, n_par):
                _run_closure_batch(config, batch)
                done.update(batch)
            done.discard(closure_index)
            return
        if closure_index in done:
            done.discard(closure_index)       # handed out once; a repeated request reruns the chain by itself
            logger.info(f'closure test {closure_index}: already run with the stacked chains')
            return

    box = config.analysis_config['parameterization'][config.parameterization]
    lower, upper = box['min'], box['max']
    n_par = len(box['names'])
    n_walk = config.n_walkers

    emu_cfg = emulation.EmulationConfig.from_config_file(
        analysis_name=config.analysis_name, parameterization=config.parameterization,
        analysis_config=config.analysis_config, config_file=config.config_file)
    emu_results = emu_cfg.read_all_emulator_groups()
    truncation_cov = emulation.compute_emulator_cov_unexplained(emu_cfg, emu_results)

    io = _data_IO()
    data = io.data_array_from_h5(config.output_dir, 'observables.h5', pseudodata_index=closure_index,
                                 observable_filter=emu_cfg.observable_filter)
    if closure_index >= 0 and world > 1 and not alone:
        # walker-sharded closure run: the pseudo-data carries random smearing (ref: data_IO.py:371), so every
        # rank conditions on rank 0's draw
        data = dict(data)
        for key in ('y', 'y_err'):
            data[key] = _same_on_all_ranks(np.asarray(data[key], dtype=np.float64))

    # upstream copies this state into every pool worker (ref: mcmc.py:77-78); here it goes to the device once
    log_posterior.initialize_pool_variables(lower, upper, emu_cfg, emu_results, data, truncation_cov)
    sampler = LoggingEnsembleSampler(n_walk, n_par, log_posterior.log_posterior, sharded=False if alone else None)
    logger.info(f'Sampler ready: {n_walk} walkers, {n_par} parameters, {sampler.world_size} GPU process(es)')

    start = np.random.uniform(lower, upper, (n_walk, n_par))       # ref: mcmc.py:88
    if not alone:
        start = _same_on_all_ranks(start)

    # burn-in in two stages; the second restarts from the best distinct points of the first
    first_stage = config.n_burn_steps // 2
    logger.info(f'Burn-in, stage 1 ({first_stage} steps)...')
    sampler.run_mcmc(start, first_stage, n_logging_steps=config.n_logging_steps)
    restart = _best_distinct(sampler, n_walk)
    sampler.reset()
    logger.info(f'Burn-in, stage 2 ({config.n_burn_steps - first_stage} steps) from the best {n_walk} points...')
    state = sampler.run_mcmc(restart, config.n_burn_steps - first_stage, n_logging_steps=config.n_logging_steps)
    sampler.reset()

    logger.info(f'Production ({config.n_sampling_steps} steps)...')
    sampler.run_mcmc(state[0], config.n_sampling_steps, n_logging_steps=config.n_logging_steps)

    if rank != 0 and not alone:
        return
    try:
        tau = sampler.get_autocorr_time()
    except Exception as err:        # chain too short for a reliable estimate (emcee's AutocorrError upstream)
        logger.info(f'No autocorrelation time: {err}')
        tau = None
    results = {'chain': sampler.get_chain(), 'acceptance_fraction': sampler.acceptance_fraction,
               'log_prob': sampler.get_log_prob(), 'autocorrelation_time': tau}
    if closure_index >= 0:
        validation_design = io.design_array_from_h5(config.output_dir, filename='observables.h5', validation_set=True)
        results['design_point'] = validation_design[closure_index]
        results['experimental_pseudodata'] = data
    # the two big outputs -- mcmc.h5 (ref: mcmc.py:125) and the pickled sampler (ref: mcmc.py:131-132), ~0.5 GB each at
    # the shipped length -- are written side by side: the file writes release the interpreter lock
    logger.info(f'Writing {config.mcmc_outputfile}')
    pickle_path = Path(config.sampler_outputfile)
    pickle_path.parent.mkdir(parents=True, exist_ok=True)

    def write_pickle():
        with open(pickle_path, 'wb') as handle:
            pickle.dump(sampler, handle, protocol=pickle.HIGHEST_PROTOCOL)     # streamed: no 0.5 GB bytes object first

    import threading
    failure = []

    def guarded():
        try:
            write_pickle()
        except BaseException as err:     # re-raised in the caller's thread
            failure.append(err)
    side = threading.Thread(target=guarded, name='gpemu-sampler-pickle')
    side.start()
    try:
        io.write_dict_to_h5(results, config.mcmc_output_dir, 'mcmc.h5', verbose=True)
    finally:
        side.join()
    if failure:
        raise failure[0]
    logger.info('MCMC finished.')


####################################################################################################
def credible_interval(samples, confidence=0.9, interval_type='quantile'):
    """(low, high) of a 1-D sample array (ref: mcmc.py:137-164).

    'quantile': equal tails.  'hpd': the narrowest interval holding ``confidence`` of the samples, searched
    over the windows that drop i points at the bottom and ``n_out - i`` at the top.
    """
    x = np.asarray(samples)
    if interval_type == 'quantile':
        tail = 0.5 * (1.0 - confidence)
        return np.quantile(x, [tail, 1.0 - tail])
    if interval_type == 'hpd':
        n_out = int((1 - confidence) * x.size)           # points left outside the interval
        ordered = np.sort(x)
        bottoms, tops = ordered[:n_out], ordered[x.size - n_out:]
        narrowest = int(np.argmin(tops - bottoms))
        return bottoms[narrowest], tops[narrowest]
    raise ValueError(f"unknown interval_type {interval_type!r}")


def map_parameters(posterior, method='quantile'):
    """Point estimate per parameter: the mean of the samples inside the central 1 % quantile band of that
    parameter's marginal (ref: mcmc.py:167-184).  ``posterior``: (n_samples, n_parameters)."""
    if method != 'quantile':
        raise ValueError(f"unknown method {method!r}")
    half_band = 0.005
    estimate = np.empty(posterior.shape[1])
    for j, column in enumerate(np.asarray(posterior).T):
        q_lo, q_hi = np.quantile(column, [0.5 - half_band, 0.5 + half_band])
        estimate[j] = column[(column >= q_lo) & (column <= q_hi)].mean()
    return estimate


####################################################################################################
class LoggingEnsembleSampler(EnsembleSampler):
    """Ensemble sampler that reports the acceptance fraction every ``n_logging_steps`` steps
    (ref: mcmc.py:187-204)."""

    def run_mcmc(self, X0, n_sampling_steps, n_logging_steps=100, **kwargs):
        logger.info(f'  running {self.nwalkers} walkers for {n_sampling_steps} steps')
        state, done = None, 0
        # advance in blocks that end on the logging steps, so that the device runs ahead of the host
        while done < n_sampling_steps:
            block = min(n_logging_steps - done % n_logging_steps, n_sampling_steps - done)
            # (the ensemble itself is only wanted at the end: the shipped settings log every 10 steps, and two blocking
            # downloads per block were a tenth of a 50 000-step run)
            state = self.advance(X0 if done == 0 else None, block, want_state=done + block >= n_sampling_steps, **kwargs)
            done += block
            if done % n_logging_steps == 0 or done == n_sampling_steps:
                frac = self.acceptance_fraction
                logger.info(f'  step {done}: acceptance fraction: mean {frac.mean()}, std {frac.std()}, '
                            f'min {frac.min()}, max {frac.max()}')
        return state


####################################################################################################
class MCMCConfig:
    """Settings of one MCMC run, read from the analysis YAML (ref: mcmc.py:207-245; same attribute names)."""

    def __init__(self, analysis_name='', parameterization='', analysis_config='', config_file='',
                 closure_index=-1, **kwargs):
        self.set_attribute(**kwargs)
        self.analysis_name, self.parameterization = analysis_name, parameterization
        self.analysis_config, self.config_file = analysis_config, config_file

        with open(config_file, 'r') as handle:
            top = yaml.safe_load(handle)
        for key in ('observable_table_dir', 'observable_config_dir', 'observables_filename'):
            setattr(self, key, top[key])
        for key in _MCMC_KEYS:
            setattr(self, key, analysis_config['parameters']['mcmc'][key])

        # <output_dir>/<analysis>_<parameterization>[/closure/results/<index>]/{mcmc.h5, mcmc_sampler.pkl}
        self.output_dir = os.path.join(top['output_dir'], f'{analysis_name}_{parameterization}')
        self.emulation_outputfile = os.path.join(self.output_dir, 'emulation.pkl')
        self.mcmc_output_dir = self.output_dir if closure_index < 0 else \
            os.path.join(self.output_dir, f'closure/results/{closure_index}')
        self.mcmc_outputfilename = 'mcmc.h5'
        self.mcmc_outputfile = os.path.join(self.mcmc_output_dir, self.mcmc_outputfilename)
        self.sampler_outputfile = os.path.join(self.mcmc_output_dir, 'mcmc_sampler.pkl')

        # parameter names are used as (raw) plot labels downstream
        par = self.analysis_config['parameterization'][self.parameterization]
        par['names'] = [str(name) for name in par['names']]

    def set_attribute(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)
