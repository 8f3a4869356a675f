#!/usr/bin/env python3
"""cProfile of the C3 emulator fit (bench.py's fit_c3 leg): where the host side of fit_gps spends its time.
   python tools/profile_fit_c3.py [n_restarts]"""
import cProfile
import os
import pstats
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
import bench  # noqa: E402

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bench.measure_fit_c3(0, 2)          # warm: code objects, workspace
pr = cProfile.Profile()
pr.enable()
out = bench.measure_fit_c3(0, nr)
pr.disable()
print(out)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
