#!/usr/bin/env python3
"""One MCMC chain at the size the reference ships (ref: config/jet_substructure.yaml: ~150 design points, d = 6, 200
walkers; one emulation group of k PCs here): microseconds per stretch-move step of the device sampler -- the regime
where a step is a handful of ~5 us launches.   python tools/time_shipped_chain.py [N F k W steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import golden_util as GU  # noqa: E402
from gpemu import synthetic  # noqa: E402
from gpemu.sampler import DeviceSampler  # noqa: E402

a = [v for v in sys.argv[1:]]
multi = bool(a) and a[0] == "groups"          # "groups": the shipped three emulation groups (5 / 11 / 25 PCs) in one sampler
if multi:
    a = a[1:]
a = [int(v) for v in a]
N, F, k, W, steps = (a + [150, 215, 11, 200, 4000][len(a):])[:5]
dms = []
for gi, (Fg, kg) in enumerate([(60, 5), (120, 11), (215, 25)] if multi else [(F, k)]):
    model, prob, _ = GU.fixed_theta_model(N, Fg, kg, seed=gi)
    dmg = GU.device_model(model)
    dmg.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    dms.append(dmg)
dm = dms[0]
if multi:
    k = "5+11+25"
import ctypes as C  # noqa: E402
from gpemu import _lib  # noqa: E402
L = _lib.lib()
for fused in (False, True):
    s = DeviceSampler(dms, W, seed=11)
    s.set_state(synthetic.make_walkers(W, seed=3))
    if fused:       # the sharded run's two-launch half-step (front kernel + GEMM) taken at one rank
        h = (C.c_char * 64)()
        _lib.check(L.gpemu_sampler_peer_export(s._h, C.cast(h, C.c_void_p)))
        _lib.check(L.gpemu_sampler_peer_import(s._h, 1, 0, C.cast(h, C.c_void_p)))
        run = lambda n: _lib.check(L.gpemu_sampler_run_peer(s._h, n, 0))
    else:
        run = lambda n: s.run(n, store=False)
    run(200)
    for rep in range(2):
        dm.sync()
        t0 = time.perf_counter()
        run(steps)
        dm.sync()
        dt = time.perf_counter() - t0
        print(f"N={N} F={F} k={k} W={W} {'fused two-launch' if fused else 'three-launch':>16s}: {dt / steps * 1e6:7.2f} us per step, "
              f"{W * steps / dt / 1e6:6.3f} M evaluations/s", flush=True)
    s.close()
for d_ in dms:
    d_.close()
