#!/usr/bin/env python3
"""Back-to-back (no host sync) log-posterior launches at small batch sizes: per-kernel HIP-event times of the
triangular GEMM and the cross-kernel kernel under sustained load.   python tools/time_small.py 64 128 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayesian-inference_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from gpemu import synthetic  # noqa: E402
from gpemu.model import DeviceModel  # noqa: E402

wl = bench.build_workload()
prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"],
                 components=wl["components"], scaler_mean=wl["mean"], scaler_scale=wl["scale"],
                 kernel_kind=0, noise=wl["noise"], cov_unexplained=wl["cun"])
dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
dev = torch.device("cuda", 0)
for B in [int(a) for a in (sys.argv[1:] or ["64", "128"])]:
    X = torch.from_numpy(synthetic.make_walkers(B, seed=1)).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    for _ in range(300):                                   # clock ramp
        dm.logpost_dev(X.data_ptr(), B, out.data_ptr())
    dm.sync()
    dm.profile(True)
    for _ in range(300):
        dm.logpost_dev(X.data_ptr(), B, out.data_ptr())
    dm.sync()
    pr = dm.profile_read()
    dm.profile(False)
    lp = out.cpu().numpy()
    print(f"B={B:5d}  trmm {pr['trmm_vsq'][0] / pr['trmm_vsq'][1] * 1e3:8.2f} us  kstar {pr['kstar'][0] / pr['kstar'][1] * 1e3:6.2f} us"
          f"  sum={lp.sum():.6f}", flush=True)
