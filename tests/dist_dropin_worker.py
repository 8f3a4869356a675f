"""Worker of test_gpu_dropin.py::test_two_process_launch_one_writer_per_file, started as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tests/dist_dropin_worker.py <dir>
It does what the reference's steering script does (fit_emulators, then run_mcmc for the production chain and for
two closure chains) WITHOUT initialising a process group itself -- the drop-in modules join the launcher's group --
and logs every file this rank writes to <dir>/writes_rank<r>.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bayesian-inference_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import dropin_util as DU  # noqa: E402
import golden_util as GU  # noqa: E402


def main(out_dir):
    from pathlib import Path
    from bayesian_inference import emulation, mcmc
    rank = int(os.environ["RANK"])
    log = open(os.path.join(out_dir, f"writes_rank{rank}.txt"), "w")

    def note(path):
        log.write(str(path) + "\n")
        log.flush()

    g = GU.load("g1_rbf_noise")
    written = {}
    dio = DU.install_fake_data_IO(g["Y"], g["design"], g["y_exp"], g["y_err"], written)
    inner_h5 = dio.write_dict_to_h5
    dio.write_dict_to_h5 = lambda results, output_dir, filename, verbose=True: (
        note(os.path.join(output_dir, filename)), inner_h5(results, output_dir, filename, verbose=verbose))[1]
    inner_emu = emulation.write_emulators
    emulation.write_emulators = lambda config, output_dict: (note(config.emulation_outputfile), inner_emu(config, output_dict))[1]
    real_write_bytes = Path.write_bytes

    def logging_write_bytes(self, data):
        if self.name == "mcmc_sampler.pkl":
            note(self)
        return real_write_bytes(self, data)
    Path.write_bytes = logging_write_bytes
    # the production chain's sampler pickle is streamed with pickle.dump(sampler, handle)
    import pickle as real_pickle
    import types
    proxy = types.SimpleNamespace(**{n: getattr(real_pickle, n) for n in dir(real_pickle) if not n.startswith("__")})

    def logging_dump(obj, handle, *a, **k):
        if os.path.basename(getattr(handle, "name", "")) == "mcmc_sampler.pkl":
            note(handle.name)
        return real_pickle.dump(obj, handle, *a, **k)
    proxy.dump = logging_dump
    mcmc.pickle = proxy

    import yaml
    path = os.path.join(out_dir, "analysis.yaml")          # written by the test before the launch
    analysis = yaml.safe_load(open(path))["test_analysis"]
    analysis["validation_indices"] = [0, 2]
    ec = emulation.EmulationConfig.from_config_file("test_analysis", "exponential", path, analysis)
    np.random.seed(1 + rank)            # the ranks' numpy states differ, as under a real launch
    emulation.fit_emulators(ec)
    emulation.EmulationConfig.sort_observables_in_matrix = property(lambda self: DU.TrivialSort("main"))
    emulation.EmulationConfig.observable_filter = property(lambda self: None)
    cfg = mcmc.MCMCConfig("test_analysis", "exponential", analysis, path)
    mcmc.run_mcmc(cfg)
    for j in range(2):
        c = mcmc.MCMCConfig("test_analysis", "exponential", analysis, path, closure_index=j)
        mcmc.run_mcmc(c, closure_index=j)
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_world_size() == 2      # joined by the drop-in, not by this script
    dist.barrier()
    log.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
