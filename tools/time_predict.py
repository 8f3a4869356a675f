#!/usr/bin/env python3
"""Metric 2 (emulation.predict GB/s, outputs resident in HBM) at several batch sizes on the C3 model."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayesian-inference_amd"))
import bench  # noqa: E402
from gpemu.model import DeviceModel  # noqa: E402

wl = bench.build_workload(0)
prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                 scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"],
                 cov_unexplained=wl["cun"], device=0)
for B in [int(a) for a in (sys.argv[1:] or ["512", "1024", "2048", "4096"])]:
    r = bench.measure_predict(dm, n_samples=B, reps=20)
    print(f"B={B}: {r['ms_per_batch']:.3f} ms  {r['value']:.0f} GB/s  ({r['roofline']['frac']:.3f} of 8 TB/s)  "
          f"{r['samples_per_s'] / 1e6:.3f} M samples/s", flush=True)
dm.close()
