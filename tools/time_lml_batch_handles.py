#!/usr/bin/env python3
"""Batched LML + gradient evaluations through H device handles at once (one host thread each), as the group fit's lock-step
driver runs them:  python tools/time_lml_batch_handles.py N nb H [reps]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import numpy as np
from gpemu import synthetic
from gpemu.fit import DeviceFit

N, nb, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
prob = synthetic.make_problem(N, 8, seed=0)
y = prob["Y"][:, 0] - prob["Y"][:, 0].mean()
theta = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.5, 0.05])
rng = np.random.default_rng(0)
ys = np.stack([y] * nb)
thetas = np.stack([theta + 0.1 * rng.normal(size=theta.size) for _ in range(nb)])
fits = [DeviceFit(prob["design"], kernel_kind=0, has_noise=True, jitter=1e-10) for _ in range(H)]
for f in fits:
    f.lml_batch(ys, thetas)


def work(f):
    for _ in range(reps):
        f.lml_batch(ys, thetas, eval_gradient=True)


t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(f,)) for f in fits]
[t.start() for t in th]
[t.join() for t in th]
dt = time.perf_counter() - t0
n = H * reps * nb
print(f"N={N} nb={nb} handles={H}: {dt / n * 1e3:.3f} ms per problem, {n * N**3 / dt / 1e12:.2f} TFLOP/s ({n * N**3 / dt / 78.6e12:.3f} of peak)")
for f in fits:
    f.close()
