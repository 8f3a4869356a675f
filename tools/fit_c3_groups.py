#!/usr/bin/env python3
"""bench.py's fit_c3 leg, cold (as the driver's bench run has it) and warm, with the lock-step driver's one- and
two-group forms (GPEMU_FIT_GROUPS) and one or two device handles (GPEMU_FIT_HANDLES): one child process per setting.
   python tools/fit_c3_groups.py"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd")]
    import bench
    for label in ("cold", "warm", "warm"):
        out = bench.measure_fit_c3(0, 50)
        print(label, json.dumps({k: out[k] for k in ("seconds", "seconds_in_library", "seconds_host_optimiser",
                                                         "lml_evaluations", "mean_lml")}), flush=True)
    sys.exit(0)
for groups, handles in (("2", "1"), ("2", "2"), ("3", "3"), ("2", "1"), ("2", "2"), ("3", "3")):
    env = dict(os.environ, GPEMU_FIT_GROUPS=groups, GPEMU_FIT_HANDLES=handles)
    print("GPEMU_FIT_GROUPS =", groups, "GPEMU_FIT_HANDLES =", handles, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
