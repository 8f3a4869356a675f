#!/usr/bin/env python3
"""Soak of the overlapped half-step at the C3 size: blocks of steps with a short bound on the cross-stream
waits; a block that falls back (mode 2) is reported with what expired, and the sampler is re-created so that the soak goes
on overlapped.  The chain is compared with a serial sampler's at the end.   python tools/soak_overlap.py [blocks steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import numpy as np  # noqa: E402

import bench  # noqa: E402
from gpemu import _lib, synthetic  # noqa: E402
from gpemu.model import DeviceModel  # noqa: E402
from gpemu.sampler import DeviceSampler  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
short = os.environ.pop("GPEMU_OVERLAP_TIMEOUT_MS", "100")      # after a warm-up block under the default bound
wl = bench.build_workload(0)
prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                 scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"],
                 cov_unexplained=wl["cun"], device=0)
dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
W = 1024
X0 = synthetic.make_walkers(W, seed=1)
ds = DeviceSampler([dm], W, seed=1)
ds.set_state(X0)
ds.run(10, store=False)              # first launches (code objects, the second stream's queue) under the default bound
dm.sync()
print(f"warm-up block: mode {ds.last_run_mode()}" + (f"  <- {_lib.last_error()}" if ds.last_run_mode() == 2 else ""), flush=True)
if ds.last_run_mode() != 1:
    X, lp = ds.get_state()
    ds.close()
    ds = DeviceSampler([dm], W, seed=77)
    ds.set_state(X, lp)
os.environ["GPEMU_OVERLAP_TIMEOUT_MS"] = short
fell = 0
for b in range(blocks):
    t0 = time.perf_counter()
    ds.run(steps, store=False)
    dm.sync()
    dt = time.perf_counter() - t0
    mode = ds.last_run_mode()
    print(f"block {b}: {dt / steps * 1e3:.4f} ms per step, mode {mode}" + (f"  <- {_lib.last_error()}" if mode == 2 else ""), flush=True)
    if mode != 1:
        fell += 1
        X, lp = ds.get_state()
        ds.close()
        ds = DeviceSampler([dm], W, seed=1 + b)
        ds.set_state(X, lp)
print(f"{fell} of {blocks} blocks fell back")
