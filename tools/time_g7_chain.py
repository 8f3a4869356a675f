#!/usr/bin/env python3
"""One chain on the reference's shipped three-group configuration WITH its observable blocks (golden G7: 200 design points,
groups of 5 / 11 / 25 PCs whose covariance is block diagonal over 2 / 4 / 10 observables, ref: emulation.py:370-388):
microseconds per stretch-move step of the device sampler.   python tools/time_g7_chain.py [walkers] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import golden_util as GU  # noqa: E402
from gpemu import synthetic  # noqa: E402
from gpemu.sampler import DeviceSampler  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
g = GU.load("g7_shipped_config")
names, mapping, block_start, cols = GU.g7_groups(g)
dms = []
for n in names:
    model = GU.group_model(g, prefix=n + "_")
    dm = GU.device_model(model)
    dm.likelihood_setup(g["y_exp"][cols[n]], g["y_err"][cols[n]], g["lo"], g["hi"], 1.0, block_start=block_start[n])
    dms.append(dm)
    print(f"{n}: {model.X_train.shape[0]} design points, {model.n_pc} PCs, {len(cols[n])} features in {len(block_start[n]) - 1} observable blocks")
s = DeviceSampler(dms, W, seed=11)
s.set_state(synthetic.make_walkers(W, seed=3, lo=g["lo"], hi=g["hi"]))
s.run(200, store=False)
for rep in range(2):
    dms[0].sync()
    t0 = time.perf_counter()
    s.run(steps, store=False)
    dms[0].sync()
    dt = time.perf_counter() - t0
    print(f"G7, {W} walkers: {dt / steps * 1e6:7.2f} us per step, {W * steps / dt / 1e6:6.3f} M evaluations/s; acceptance {s.counts()[0].sum() / (W * s.counts()[1]):.3f}", flush=True)
s.close()
