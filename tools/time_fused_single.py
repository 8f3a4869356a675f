#!/usr/bin/env python3
"""One GPU, C3: the three-launch half-step (cross-kernel, GEMM, likelihood + accept) against the fused two-launch
half-step of the sharded run taken at one rank (front kernel, GEMM): ms per step of each."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayesian-inference_amd"))
import bench  # noqa: E402
from gpemu import _lib, synthetic  # noqa: E402
from gpemu.model import DeviceModel  # noqa: E402
from gpemu.sampler import DeviceSampler  # noqa: E402

wl = bench.build_workload()
prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                 scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"], cov_unexplained=wl["cun"])
dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
L = _lib.lib()
W = 1024
X0 = synthetic.make_walkers(W, seed=1)
for fused in (False, True, False, True):
    ds = DeviceSampler([dm], W, seed=1)
    ds.set_state(X0)
    if fused:
        h = (C.c_char * 64)()
        _lib.check(L.gpemu_sampler_peer_export(ds._h, C.cast(h, C.c_void_p)))
        _lib.check(L.gpemu_sampler_peer_import(ds._h, 1, 0, C.cast(h, C.c_void_p)))
        run = lambda n: _lib.check(L.gpemu_sampler_run_peer(ds._h, n, 0))
    else:
        run = lambda n: ds.run(n, store=False)
    for _ in range(4):
        run(400)
    dm.sync()
    t0 = time.perf_counter()
    run(1000)
    dm.sync()
    print(f"{'fused two-launch' if fused else 'three-launch   '} half-step: {(time.perf_counter() - t0):.4f} ms per step", flush=True)
    ds.close()
dm.close()
