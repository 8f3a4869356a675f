#!/usr/bin/env python3
"""Per-worker finish times of the dominant kernel (in-kernel stamps of launch 600, GPEMU_TRMM_STAMP_FILE) and the step
time of the same bench run.  Needs the diagnostic library: `make -C bayesian-inference_amd/csrc clean all STAMPS=1`
(the product build carries no stamp code; rebuild without STAMPS afterwards)."""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stamp = tempfile.mktemp(suffix=".txt")
env = dict(os.environ, GPEMU_TRMM_STAMP_FILE=stamp)
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "20", "--no-cpu-baseline", "--no-fit"],
                     env=env, capture_output=True, text=True)
d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
rows = [l.split() for l in open(stamp)]
fin = np.array([float(r[-1]) for r in rows]); n = np.array([int(r[1]) for r in rows])
print(f"step {d['ms_per_step']:.4f} ms, trmm {d['roofline']['avg_launch_us']:.2f} us (events), frac {d['roofline']['frac']:.3f}; "
      f"workers finish min {fin.min():.1f} med {np.median(fin):.1f} max {fin.max():.1f}; by items " +
      ", ".join(f"{c}: {fin[n == c].mean():.1f} ({int((n == c).sum())})" for c in np.unique(n)), flush=True)
