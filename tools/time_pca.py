"""Times gpemu_pca_fit at a given size and checks it against the oracle's LAPACK SVD.
    python tools/time_pca.py N F [k]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
from gpemu import fit, synthetic
from oracle import gp_oracle as O

N, F = int(sys.argv[1]), int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prob = synthetic.make_problem(N, F, seed=0)
Y = prob["Y"]
fit.pca_fit(Y[:64, :32])
t0 = time.time(); out = fit.pca_fit(Y); t1 = time.time()
print(f"device pca_fit {N}x{F}: {t1 - t0:.3f} s, sweeps {out['n_sweeps']}", flush=True)
sc = O.scaler_fit(Y); Ys = (Y - sc[0]) / sc[1]
t0 = time.time(); ref = O.pca_fit(Ys); t1 = time.time()
print(f"oracle pca_fit: {t1 - t0:.3f} s", flush=True)
fa = np.argmax(np.abs(ref["components"]), axis=1)
print("flip_argmax equal (first k):", np.array_equal(fa[:k], out["flip_argmax"][:k]),
      " all:", int(np.sum(fa != out["flip_argmax"])), "differ")
print("ev rel err (first k):", np.max(np.abs(out["explained_variance"][:k] / ref["explained_variance"][:k] - 1)))
print("comp abs err (first k):", np.max(np.abs(out["components"][:k] - ref["components"][:k])))
print("Y_pca abs err (first k):", np.max(np.abs(out["Y_pca"][:, :k] - ref["Y_pca"][:, :k])))
