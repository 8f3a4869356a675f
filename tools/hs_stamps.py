#!/usr/bin/env python3
"""Phase times inside halfstep_small_kernel (csrc/k_halfstep.hip) at the shipped shape: clock64 stamps of wave 0 of three
workgroups.  Needs the diagnostic library: make -C bayesian-inference_amd/csrc clean all HS_STAMPS=1 (rebuild without
afterwards).   python tools/hs_stamps.py [groups]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-inference_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import golden_util as GU  # noqa: E402
from gpemu import _lib, synthetic  # noqa: E402
from gpemu.sampler import DeviceSampler  # noqa: E402

multi = len(sys.argv) > 1 and sys.argv[1] == "groups"
dms = []
for gi, (Fg, kg) in enumerate([(60, 5), (120, 11), (215, 25)] if multi else [(215, 11)]):
    model, prob, _ = GU.fixed_theta_model(150, Fg, kg, seed=gi)
    dmg = GU.device_model(model)
    dmg.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    dms.append(dmg)
s = DeviceSampler(dms, 200, seed=11)
s.set_state(synthetic.make_walkers(200, seed=3))
L = _lib.lib()
f = L.gpemu_debug_hs_stamps
f.restype = C.c_int
names = ["entry -> proposal in LDS", "cross-kernel tiles", "triangular GEMM", "final sums + store"]
for rep in range(3):
    s.run(500, store=False)
    dms[0].sync()
    st = (C.c_longlong * 96)()
    assert f(st) == 0
    st = np.array(st[:]).reshape(3, 32)
    print("run", rep)
    for w in range(3):
        d = np.diff(st[w, :5]) / 1.0
        print(f"  workgroup {w}: " + ", ".join(f"{n} {v:.2f}" for n, v in zip(names, d)) + f"; total {(st[w, 4] - st[w, 0]) / 1.0:.2f};"
              f" start after workgroup 0: {(st[w, 0] - st[0, 0]) / 1.0:.2f}")
        if w == 0:
            t = st[0, 2]
            for rb in range(8):
                if st[0, 8 + rb] == 0:
                    break
                print(f"    row block {rb}: done {st[0, 8 + rb] - t} ticks after the one before")
                t = st[0, 8 + rb]

# wall-clock picture of the last launch: start / end of every workgroup (wall_clock64: one 100 MHz counter for the chip)
g = L.gpemu_debug_hs_wall
g.restype = C.c_int
wl = (C.c_longlong * 2048)()
assert g(wl) == 0
wl = np.array(wl[:]).reshape(512, 4)
idx = np.flatnonzero((wl[:, 0] > 0) & (wl[:, 2] > wl[:, 0]))
t0 = wl[idx, 0].min()
starts, ends = (wl[idx, 0] - t0) / 100.0, (wl[idx, 2] - t0) / 100.0
print(f"{len(idx)} workgroups: start {starts.min():.2f} .. {starts.max():.2f} us, end min {ends.min():.2f} median {np.median(ends):.2f} max {ends.max():.2f} us "
      "after the first start; the five latest: " + ", ".join(f"{idx[j]}: {ends[j]:.2f}" for j in np.argsort(ends)[-5:]))
s.close()
