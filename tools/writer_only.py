#!/usr/bin/env python3
"""Times the covariance writer alone (GP stage results reused): gpemu_predict_full_dev minus the GP stage is not
exposed, so this times the whole call and the GP stage (gp_predict_dev) separately at B = 512."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import numpy as np, torch
import bench
from gpemu import synthetic
from gpemu.model import DeviceModel
wl = bench.build_workload(0); prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                 scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"],
                 cov_unexplained=wl["cun"], device=0)
dev = torch.device("cuda", 0); F, k = dm.F, dm.k
B = 512
os.environ["GPEMU_PREDICT_SPLIT"] = "0"
X = torch.from_numpy(synthetic.make_walkers(B, seed=2)).to(dev)
cv = torch.empty((B, F), dtype=torch.float64, device=dev); cov = torch.empty((B, F, F), dtype=torch.float64, device=dev)
mean = torch.empty((B, k), dtype=torch.float64, device=dev); var = torch.empty((B, k), dtype=torch.float64, device=dev)
st = torch.cuda.Stream(device=dev)
def timeit(fn, reps=10):
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps): fn()
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / reps
t_full = timeit(lambda: dm.predict_full_dev(X.data_ptr(), B, float(B), cv.data_ptr(), cov.data_ptr(), stream=st.cuda_stream))
import ctypes as C
from gpemu import _lib
t_gp = timeit(lambda: _lib.check(_lib.lib().gpemu_gp_predict_dev(dm.handle, B, C.c_void_p(X.data_ptr()), C.c_void_p(mean.data_ptr()), C.c_void_p(var.data_ptr()), C.c_void_p(st.cuda_stream))))
nb = 8 * B * F * F
print(f"B {B}: full {t_full*1e3:.1f} us, GP stage {t_gp*1e3:.1f} us, writer ~{(t_full-t_gp)*1e3:.1f} us = {nb/((t_full-t_gp)*1e-3)/1e12:.2f} TB/s  [PM_DBG={os.environ.get('GPEMU_PM_DBG')}, VALU={os.environ.get('GPEMU_PREDICT_VALU')}]")
