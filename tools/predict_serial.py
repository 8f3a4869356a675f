#!/usr/bin/env python3
"""emulation.predict (metric 2) at B = 1024, serial form, several repetitions with the per-call time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import torch
import bench
from gpemu.model import DeviceModel
wl = bench.build_workload(0); prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                 scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"],
                 cov_unexplained=wl["cun"], device=0)
for r in range(4):
    out = bench.measure_predict(dm, n_samples=int(os.environ.get("PRED_B", "1024")), reps=10)
    print(f"B {out['batch']}: {out['ms_per_batch']:.4f} ms = {out['value']:.0f} GB/s ({out['roofline']['frac']:.3f} of 8 TB/s) "
          f"[CBW={os.environ.get('GPEMU_PM_CBW')}]", flush=True)
dm.close()
