#!/usr/bin/env python3
"""One configuration of tools/stage_shares.py, for rocprofv3: GPEMU_SERIAL_MASK from the environment, B = 512."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import torch
import bench
from gpemu import synthetic
from gpemu.model import DeviceModel
wl = bench.build_workload(0); prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                 scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"],
                 cov_unexplained=wl["cun"], device=0)
dev = torch.device("cuda", 0); F, k = dm.F, dm.k
B = int(os.environ.get("STAGE_B", "512"))
os.environ.setdefault("GPEMU_PREDICT_SPLIT", "0")
X = torch.from_numpy(synthetic.make_walkers(B, seed=2)).to(dev)
cv = torch.empty((B, F), dtype=torch.float64, device=dev); cov = torch.empty((B, F, F), dtype=torch.float64, device=dev)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    for _ in range(12):
        dm.predict_full_dev(X.data_ptr(), B, float(B), cv.data_ptr(), cov.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
dm.close()
