#!/usr/bin/env python3
"""The covariance writer's own time (HIP events around its launch) on a share of the CUs: GPEMU_SERIAL_MASK='32,w'."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for w in (32, 16, 12, 10, 8):
    env = dict(os.environ, GPEMU_SERIAL_MASK=f"32,{w}", GPEMU_SERIAL_MASK_PRINT="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stage_one.py")], env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stderr.splitlines() if l.startswith("writer on")]
    print(lines[-1] if lines else out.stderr[-500:], flush=True)
