#!/usr/bin/env python3
"""Soak of the device Cholesky (the three-wave sweep hands columns from wave to wave through LDS flags): random SPD matrices
of many sizes and conditionings, each factored TWICE -- the two factors must have the same bits -- and compared with LAPACK;
batched LML + gradient evaluations repeated -- same bits every time.   python tools/soak_cholesky.py [n_matrices] [n_batches]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import numpy as np  # noqa: E402

from gpemu import synthetic  # noqa: E402
from gpemu.fit import DeviceFit, cholesky  # noqa: E402

nm = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(12345)
t0 = time.perf_counter()
worst = 0.0
for i in range(nm):
    n = int(rng.integers(1, 700))
    M = rng.normal(size=(n, n))
    cond = 10.0 ** rng.uniform(0, 8)
    A = M @ M.T / n + np.eye(n) / cond
    L1 = np.tril(cholesky(A))
    L2 = np.tril(cholesky(A))
    assert L1.tobytes() == L2.tobytes(), f"matrix {i} (n = {n}): two factorisations differ"
    ref = np.linalg.cholesky(A)
    err = np.max(np.abs(L1 - ref)) / np.max(np.abs(ref))
    worst = max(worst, err * min(cond, 1e8) ** -0.0)
    assert err < 1e-9 * max(1.0, cond * 1e-4), f"matrix {i} (n = {n}, cond ~ {cond:.1e}): {err:.2e} from LAPACK"
print(f"{nm} matrices (n = 1 .. 699, conditioning up to 1e8): factor twice -> same bits; worst deviation from LAPACK {worst:.2e}")
for N, nb in ((1000, 64), (300, 37)):
    prob = synthetic.make_problem(N, 8, seed=1)
    y = prob["Y"][:, 0] - prob["Y"][:, 0].mean()
    theta = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.5, 0.05])
    ys = np.stack([y] * nb)
    thetas = np.stack([theta + 0.2 * rng.normal(size=theta.size) for _ in range(nb)])
    fit = DeviceFit(prob["design"], kernel_kind=0, has_noise=True, jitter=1e-10)
    first = None
    for _ in range(nbatch):
        lml, grad, info = fit.lml_batch(ys, thetas)
        key = lml.tobytes() + grad.tobytes() + info.tobytes()
        first = first or key
        assert key == first, f"N = {N}: a batched evaluation differs from the first"
    one = fit.lml(ys[3], thetas[3], eval_gradient=True)
    assert one[0] == lml[3] and np.array_equal(one[1], grad[3]), "a single evaluation differs from its place in the batch"
    fit.close()
    print(f"N = {N}: {nbatch} batched evaluations of {nb} problems: same bits every time, = the single evaluation")
print(f"soak passed in {time.perf_counter() - t0:.1f} s")
