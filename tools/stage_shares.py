#!/usr/bin/env python3
"""Each stage of emulation.predict ALONE on a share of the chip (GPEMU_SERIAL_MASK = "g,w": GP stage on g CUs per XCD,
then the covariance writer on w): what the two stages cost on the CUs a partition would give them."""
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
B = 512
os.environ["GPEMU_PREDICT_SPLIT"] = "0"
X = torch.from_numpy(synthetic.make_walkers(B, seed=2)).to(dev)
cv = torch.empty((B, F), dtype=torch.float64, device=dev); cov = torch.empty((B, F, F), dtype=torch.float64, device=dev)
st = torch.cuda.Stream(device=dev)
def timeit(reps=10):
    fn = lambda: dm.predict_full_dev(X.data_ptr(), B, float(B), cv.data_ptr(), cov.data_ptr(), stream=st.cuda_stream)
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps): fn()
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
base = {}
for g, w in [(32, 32), (32, 8)]:
    os.environ["GPEMU_SERIAL_MASK"] = f"{g},{w}"
    t = timeit()
    base[(g, w)] = t
    note = ""
    if g == 32 and w < 32: note = f"writer on {8*w} CUs: +{t - base[(32, 32)]:.0f} us"
    if w == 32 and g < 32: note = f"GP stage on {8*g} CUs: +{t - base[(32, 32)]:.0f} us"
    print(f"B {B} GP stage on {g:2d}/XCD, writer on {w:2d}/XCD, serial: {t:.1f} us  {note}", flush=True)
