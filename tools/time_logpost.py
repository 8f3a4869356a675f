#!/usr/bin/env python3
"""Time the log-posterior pipeline (kstar + triangular GEMM + likelihood) at several batch sizes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayesian-inference_amd"))
import bench  # noqa: E402
from gpemu import synthetic  # noqa: E402
from gpemu.model import DeviceModel  # noqa: E402

wl = bench.build_workload()
prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"],
                 components=wl["components"], scaler_mean=wl["mean"], scaler_scale=wl["scale"],
                 kernel_kind=0, noise=wl["noise"], cov_unexplained=wl["cun"])
dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
for B in [int(a) for a in (sys.argv[1:] or ["64", "128", "256", "512"])]:
    X = synthetic.make_walkers(B, seed=1)
    dm.logpost(X)
    dm.profile(True)
    t0 = time.perf_counter()
    for _ in range(20):
        lp = dm.logpost(X)
    dt = (time.perf_counter() - t0) / 20
    pr = dm.profile_read()
    dm.profile(False)
    print(f"B={B:5d}  trmm {pr['trmm_vsq'][0] / pr['trmm_vsq'][1] * 1e3:8.1f} us  kstar {pr['kstar'][0] / pr['kstar'][1] * 1e3:6.1f} us"
          f"  host round trip {dt * 1e6:8.1f} us  sum={lp.sum():.6f}")
