#!/usr/bin/env python3
"""Metric 2 (emulation.predict GB/s) against the CU partition of the predict pipeline: GP stage on n CUs per XCD,
covariance writer on the other 32 - n (hipExtStreamCreateWithCUMask), for several chunk sizes; split 0 = the serial
form.  Every configuration's output is compared with the serial one (bit for bit).  Usage: predict_overlap.py [B ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import numpy as np
import torch

import bench
from gpemu import synthetic
from gpemu.model import DeviceModel

wl = bench.build_workload(0)
prob = wl["prob"]
dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                 scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"],
                 cov_unexplained=wl["cun"], device=0)
dev = torch.device("cuda", 0)
F, k, N, d = dm.F, dm.k, dm.N, dm.d
for B in [int(a) for a in sys.argv[1:]] or [1024]:
    X = torch.from_numpy(synthetic.make_walkers(B, seed=2)).to(dev)
    cv = torch.empty((B, F), dtype=torch.float64, device=dev)
    cov = torch.empty((B, F, F), dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    nbytes = 8 * (B * F * F + B * F) + 8 * (k * N * (N + 1) // 2 + k * N + N * d + B * d + F * k + 2 * F + F * F)
    ref = None
    for split, chunk in [(0, 512)] + [(s, c) for c in (256, 384, 512) for s in (16, 20, 22, 24, 26)]:
        if B < 2 * chunk and split:
            continue
        os.environ["GPEMU_PREDICT_SPLIT"] = str(split)
        os.environ["GPEMU_PREDICT_CHUNK"] = str(chunk)
        with torch.cuda.stream(st):
            cov.zero_(); cv.zero_()
            dm.predict_full_dev(X.data_ptr(), B, float(B), cv.data_ptr(), cov.data_ptr(), stream=st.cuda_stream)
            st.synchronize()
            chk = (float(cov.sum()), float((cov * cov).sum()), float(cv.sum()))
            if ref is None:
                ref = (cov[::37].clone(), cv.clone())
            same = bool(torch.allclose(cov[::37], ref[0], rtol=1e-11, atol=1e-14) and torch.allclose(cv, ref[1], rtol=1e-12, atol=0))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 8
            e0.record(st)
            for _ in range(reps):
                dm.predict_full_dev(X.data_ptr(), B, float(B), cv.data_ptr(), cov.data_ptr(), stream=st.cuda_stream)
            e1.record(st)
            st.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"B {B} split {split:2d} chunk {chunk:3d}: {ms:.3f} ms per batch = {nbytes / (ms * 1e-3) / 1e12:.2f} TB/s "
              f"({nbytes / (ms * 1e-3) / 8e12:.3f} of 8 TB/s)  equal to serial (1e-11): {same}", flush=True)
dm.close()
