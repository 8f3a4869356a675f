#!/usr/bin/env python3
"""rocprofv3 driver: C3 model, `reps` calls of emulation.predict's device path with B rows (outputs stay in HBM)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayesian-inference_amd"))
import bench  # noqa: E402
from gpemu.model import DeviceModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wl = bench.build_workload()
prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"],
                 components=wl["components"], scaler_mean=wl["mean"], scaler_scale=wl["scale"],
                 kernel_kind=0, noise=wl["noise"], cov_unexplained=wl["cun"])
print(bench.measure_predict(dm, n_samples=B, reps=reps))
