#!/usr/bin/env python3
"""The C3 workload (N = 1000, 500 observables' bins, 10 PCs, 1024 walkers) with its covariance block diagonal over nb
observables (ref: emulation.py:370-388): ms per stretch-move step with the likelihood's blocks taken in turn by one wave per
proposal (GPEMU_NO_LOGLIK_TASKS=1) and as tasks on waves of their own.   python tools/time_c3_blocks.py [blocks] [walkers] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import numpy as np  # noqa: E402

import bench  # noqa: E402
from gpemu import synthetic  # noqa: E402
from gpemu.model import DeviceModel  # noqa: E402
from gpemu.sampler import DeviceSampler  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
wl = bench.build_workload(0, 1000, 500, 10, seed=0)
prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                 scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"], cov_unexplained=wl["cun"], device=0)
blocks = [int(round(i * 500 / nb)) for i in range(nb + 1)]
dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0, block_start=blocks if nb > 1 else None)
X0 = synthetic.make_walkers(W, seed=3)
res = {}
for name, env in (("serial", {"GPEMU_NO_LOGLIK_TASKS": "1"}), ("tasks", {"GPEMU_LOGLIK_TASKS_MAX_ROWS": "100000"}), ("default", {})):
    for k in ("GPEMU_NO_LOGLIK_TASKS", "GPEMU_LOGLIK_TASKS_MAX_ROWS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ds = DeviceSampler([dm], W, seed=11)
    ds.set_state(X0)
    ds.run(100, store=False)
    dm.sync()
    t0 = time.perf_counter()
    ds.run(steps, store=False)
    dm.sync()
    dt = time.perf_counter() - t0
    res[name] = ds.get_state()
    print(f"{nb} observable blocks, {W} walkers, {name:8s}: {dt / steps * 1e3:.4f} ms per step", flush=True)
    ds.close()
print("same chain:", all(np.array_equal(a, b) for a, b in zip(res["serial"], res["tasks"])))
dm.close()
