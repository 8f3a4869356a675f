#!/usr/bin/env python3
"""Closure-test shape of the shipped analysis (ref: config/jet_substructure.yaml: ~150 design points after exclusions,
d = 6, k = 11 PCs, 200 walkers; ref: steer_analysis.py:168-183: one MCMC per validation design point, 30 of them):
log-posterior evaluations per second with the 30 chains run one after the other (one DeviceSampler each) and stacked
in ONE multi-chain sampler (gpemu_sampler_create_chains).   python tools/bench_closure.py [N F k W C steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import golden_util as GU  # noqa: E402
from gpemu import synthetic  # noqa: E402
from gpemu.sampler import DeviceSampler  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
N, F, k, W, C, steps = (a + [150, 215, 11, 200, 30, 400][len(a):])[:6]
model, prob, _ = GU.fixed_theta_model(N, F, k, seed=0)
dm = GU.device_model(model)
rng = np.random.default_rng(5)
ys = prob["y_exp"][None, :] + prob["y_err"][None, :] * rng.normal(size=(C, F))       # pseudo-data of C validation points
X0 = [synthetic.make_walkers(W, seed=100 + c) for c in range(C)]

# one chain after the other
dm.likelihood_setup(ys[0], prob["y_err"], prob["lo"], prob["hi"], 1.0)
warm = DeviceSampler([dm], W, seed=1); warm.set_state(X0[0]); warm.run(50, store=False); warm.close()
t0 = time.perf_counter()
for c in range(C):
    dm.likelihood_setup(ys[c], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    s = DeviceSampler([dm], W, seed=1000 + c)
    s.set_state(X0[c])
    s.run(steps)
    last = s.get_chain()[0][-1]
    s.close()
t_seq = time.perf_counter() - t0

# stacked
dm.likelihood_setup(ys, prob["y_err"], prob["lo"], prob["hi"], 1.0)
warm = DeviceSampler([dm], W, seeds=[7 + c for c in range(C)]); warm.set_state(np.concatenate(X0)); warm.run(50, store=False); warm.close()
ms = DeviceSampler([dm], W, seeds=[1000 + c for c in range(C)])
ms.set_state(np.concatenate(X0))
t0 = time.perf_counter()
ms.run(steps)
chain = ms.get_chain()[0]
t_stack = time.perf_counter() - t0
same = np.array_equal(chain[-1, (C - 1) * W:], last)
ms.close(); dm.close()
ev = C * W * steps
print(f"N={N} F={F} k={k}: {C} chains x {W} walkers x {steps} steps = {ev} log-posterior evaluations")
print(f"  one chain after the other: {t_seq:7.3f} s  {ev / t_seq / 1e6:7.3f} M evals/s")
print(f"  stacked in one sampler   : {t_stack:7.3f} s  {ev / t_stack / 1e6:7.3f} M evals/s   ({t_seq / t_stack:.1f}x)"
      f"   last chain identical: {same}")
