#!/usr/bin/env python3
"""The C3 emulator fit (bench.py's fit_c3 leg) alone:  python tools/time_fit_c3.py [n_restarts] [batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import bench
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for b in (sys.argv[2:] or [os.environ.get("GPEMU_FIT_BATCH", "32")]):
    os.environ["GPEMU_FIT_BATCH"] = b
    out = bench.measure_fit_c3(0, nr)
    print("batch", b, "driver", os.environ.get("GPEMU_FIT_DRIVER", "direct"), out, flush=True)
