"""Drop-in replacement of ``bayesian_inference.mcmc`` (ref: src/bayesian_inference/mcmc.py).

``run_mcmc(config, closure_index=-1)`` keeps the reference's flow -- uniform start in the box, burn-in
in two halves with repositioning on the n_walkers best unique log-probabilities, production, then
``mcmc.h5`` (``chain``, ``acceptance_fraction``, ``log_prob``, ``autocorrelation_time``, closure extras)
and the pickled sampler -- but the emcee ensemble + multiprocessing pool (ref: mcmc.py:77-85) is
replaced by ``gpemu.sampler.EnsembleSampler``: the walkers, the stretch move and the log-posterior
live on the GPU(s).  With ``torch.distributed`` initialised (one process per GPU) the proposing
half-ensemble is sharded over the ranks and only rank 0 writes the outputs.
"""
from __future__ import annotations

import logging
import os
import pickle

import numpy as np
import yaml

from bayesian_inference import emulation, log_posterior
from gpemu.sampler import EnsembleSampler

logger = logging.getLogger(__name__)


def _data_IO():
    from bayesian_inference import data_IO
    return data_IO


def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


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


####################################################################################################
def run_mcmc(config, closure_index=-1):
    """Markov chain Monte Carlo calibration with the affine-invariant ensemble sampler
    (ref: mcmc.py:34-134)."""
    par = config.analysis_config['parameterization'][config.parameterization]
    names, lo, hi = par['names'], par['min'], par['max']
    ndim = len(names)
    rank, world = _rank_world()
    owner = closure_owner(closure_index, world)
    if owner is not None and owner != rank:
        logger.info(f'closure test {closure_index}: runs on rank {owner}')
        return
    replica = owner is not None          # this rank runs the whole chain alone

    emulation_config = emulation.EmulationConfig.from_config_file(
        analysis_name=config.analysis_name, parameterization=config.parameterization,
        analysis_config=config.analysis_config, config_file=config.config_file)
    emulation_results = emulation_config.read_all_emulator_groups()
    emulator_cov_unexplained = emulation.compute_emulator_cov_unexplained(emulation_config, emulation_results)

    data_IO = _data_IO()
    experimental_results = data_IO.data_array_from_h5(config.output_dir, 'observables.h5', pseudodata_index=closure_index,
                                                      observable_filter=emulation_config.observable_filter)

    if closure_index >= 0 and world > 1 and not replica:
        # walker-sharded closure run: the pseudo-data carries random smearing (ref: data_IO.py:371), every
        # rank must condition on rank 0's draw
        experimental_results = dict(experimental_results)
        for key in ('y', 'y_err'):
            experimental_results[key] = _broadcast_from_rank0(np.asarray(experimental_results[key], dtype=np.float64))
    # the reference replicates this state into every pool worker (mcmc.py:77-78); here it is uploaded once
    log_posterior.initialize_pool_variables(lo, hi, emulation_config, emulation_results, experimental_results,
                                            emulator_cov_unexplained)
    logger.info('Initializing sampler...')
    sampler = LoggingEnsembleSampler(config.n_walkers, ndim, log_posterior.log_posterior,
                                     sharded=False if replica else None)

    random_pos = np.random.uniform(lo, hi, (config.n_walkers, ndim))
    if not replica:
        random_pos = _broadcast_from_rank0(random_pos)

    logger.info(f'Parallelizing over {sampler.world_size} GPU process(es)...')
    logger.info('Starting initial burn-in...')
    nburn0 = config.n_burn_steps // 2
    sampler.run_mcmc(random_pos, nburn0, n_logging_steps=config.n_logging_steps)

    logger.info('Resampling walker positions...')
    X0 = sampler.flatchain[np.unique(sampler.flatlnprobability, return_index=True)[1][-config.n_walkers:]]
    sampler.reset()
    X0 = sampler.run_mcmc(X0, config.n_burn_steps - nburn0, n_logging_steps=config.n_logging_steps)[0]
    sampler.reset()
    logger.info('Burn-in complete.')

    logger.info('Starting production...')
    sampler.run_mcmc(X0, config.n_sampling_steps, n_logging_steps=config.n_logging_steps)

    if rank != 0 and not replica:
        return
    logger.info('Writing chain to file...')
    output_dict = {}
    output_dict['chain'] = sampler.get_chain()
    output_dict['acceptance_fraction'] = sampler.acceptance_fraction
    output_dict['log_prob'] = sampler.get_log_prob()
    try:
        output_dict['autocorrelation_time'] = sampler.get_autocorr_time()
    except Exception as e:
        output_dict['autocorrelation_time'] = None
        logger.info(f"Could not compute autocorrelation time: {str(e)}")
    if closure_index >= 0:
        design_point = data_IO.design_array_from_h5(config.output_dir, filename='observables.h5',
                                                    validation_set=True)[closure_index]
        output_dict['design_point'] = design_point
        output_dict['experimental_pseudodata'] = experimental_results
    data_IO.write_dict_to_h5(output_dict, config.mcmc_output_dir, 'mcmc.h5', verbose=True)

    os.makedirs(os.path.dirname(config.sampler_outputfile) or ".", exist_ok=True)
    with open(config.sampler_outputfile, 'wb') as f:
        pickle.dump(sampler, f)
    logger.info('Done.')


def _broadcast_from_rank0(arr):
    """All ranks must start from the same ensemble (only rank 0's draw is used)."""
    try:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
            dist.broadcast(t, src=0)
            return t.cpu().numpy()
    except ImportError:
        pass
    return arr


####################################################################################################
def credible_interval(samples, confidence=0.9, interval_type='quantile'):
    """Credible interval of a 1-D array of samples: 'hpd' or 'quantile' (ref: mcmc.py:137-164)."""
    if interval_type == 'hpd':
        nci = int((1 - confidence) * samples.size)
        argp = np.argpartition(samples, [nci, samples.size - nci])
        lows = np.sort(samples[argp[:nci]])
        highs = np.sort(samples[argp[-nci:]])
        i = np.argmin(highs - lows)
        ci = lows[i], highs[i]
    elif interval_type == 'quantile':
        ci = np.quantile(samples, [(1 - confidence) / 2, 1 - (1 - confidence) / 2])
    return ci


def map_parameters(posterior, method='quantile'):
    """MAP estimate: mean of the samples in a narrow central quantile band per parameter
    (ref: mcmc.py:167-184)."""
    if method == 'quantile':
        central_quantile = 0.01
        lower = np.quantile(posterior, 0.5 - central_quantile / 2, axis=0)
        upper = np.quantile(posterior, 0.5 + central_quantile / 2, axis=0)
        mask = (posterior >= lower) & (posterior <= upper)
        map_parameters = np.array([posterior[mask[:, i], i].mean() for i in range(posterior.shape[1])])
    return map_parameters


####################################################################################################
class LoggingEnsembleSampler(EnsembleSampler):
    """Ensemble sampler with the reference's acceptance-fraction log line (ref: mcmc.py:187-204)."""

    def run_mcmc(self, X0, n_sampling_steps, n_logging_steps=100, **kwargs):
        logger.info(f'  running {self.nwalkers} walkers for {n_sampling_steps} steps')
        result = None
        done = 0
        # advance in blocks that end on the logging steps, so the device runs ahead of the host
        while done < n_sampling_steps:
            block = min(n_logging_steps - done % n_logging_steps, n_sampling_steps - done)
            result = self.advance(X0 if done == 0 else None, block, **kwargs)
            done += block
            if done % n_logging_steps == 0 or done == n_sampling_steps:
                af = self.acceptance_fraction
                logger.info(f'  step {done}: acceptance fraction: mean {af.mean()}, std {af.std()}, '
                            f'min {af.min()}, max {af.max()}')
        return result


####################################################################################################
class MCMCConfig:
    """MCMC settings read from the YAML (ref: mcmc.py:207-245); same attribute names."""

    def __init__(self, analysis_name='', parameterization='', analysis_config='', config_file='',
                 closure_index=-1, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)
        self.analysis_name = analysis_name
        self.parameterization = parameterization
        self.analysis_config = analysis_config
        self.config_file = config_file
        with open(self.config_file, 'r') as stream:
            config = yaml.safe_load(stream)
        self.observable_table_dir = config['observable_table_dir']
        self.observable_config_dir = config['observable_config_dir']
        self.observables_filename = config["observables_filename"]

        mcmc_configuration = analysis_config["parameters"]["mcmc"]
        self.n_walkers = mcmc_configuration['n_walkers']
        self.n_burn_steps = mcmc_configuration['n_burn_steps']
        self.n_sampling_steps = mcmc_configuration['n_sampling_steps']
        self.n_logging_steps = mcmc_configuration['n_logging_steps']

        self.output_dir = os.path.join(config['output_dir'], f'{analysis_name}_{parameterization}')
        self.emulation_outputfile = os.path.join(self.output_dir, 'emulation.pkl')
        self.mcmc_outputfilename = 'mcmc.h5'
        if closure_index < 0:
            self.mcmc_output_dir = self.output_dir
        else:
            self.mcmc_output_dir = os.path.join(self.output_dir, f'closure/results/{closure_index}')
        self.mcmc_outputfile = os.path.join(self.mcmc_output_dir, 'mcmc.h5')
        self.sampler_outputfile = os.path.join(self.mcmc_output_dir, 'mcmc_sampler.pkl')

        unformatted = self.analysis_config['parameterization'][self.parameterization]['names']
        self.analysis_config['parameterization'][self.parameterization]['names'] = [rf'{s}' for s in unformatted]

    def set_attribute(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)
