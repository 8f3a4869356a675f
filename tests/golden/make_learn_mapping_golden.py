#!/usr/bin/env python3
"""Golden G6: the reference's ``SortEmulationGroupObservables.learn_mapping`` (ref: emulation.py:289-344) run on the
reference's own fixture ``tests/test_data/observables.h5`` for a two-group split of its 16 observables.

Run in the build container only (needs /root/reference):   python tests/golden/make_learn_mapping_golden.py

The reference modules are imported unchanged from /root/reference/src.  silx is not installed: the two names data_IO
imports from it are served by gpemu.h5io (whose reader is pinned against h5py on this very file by
tests/test_h5io.py::test_reader_on_the_reference_fixture).  Only arrays are stored: observable names, group names,
slice bounds, shape.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bayesian-inference_amd"))
from gpemu import h5io  # noqa: E402

h5io.install_silx_shim()
sys.path.insert(0, "/root/reference/src")
for name in [m for m in sys.modules if m.startswith("bayesian_inference")]:
    del sys.modules[name]
from bayesian_inference import data_IO, emulation  # noqa: E402  (the reference)

assert emulation.__file__.startswith("/root/reference/"), emulation.__file__

GROUPS = {                      # group name -> (include_list, exclude_list): substrings of the observable labels
    "rhic": (["200__AuAu"], []),
    "lhc": (["2760__PbPb", "5020__PbPb"], []),
}


class GroupCfg:
    def __init__(self, include, exclude):
        self.observable_filter = data_IO.ObservableFilter(include_list=include, exclude_list=exclude)


class EmuCfg:
    output_dir = "/root/reference/tests/test_data"
    emulation_groups_config = {g: GroupCfg(*f) for g, f in GROUPS.items()}


m = emulation.SortEmulationGroupObservables.learn_mapping(EmuCfg())
keys = list(m.emulation_group_to_observable_matrix)
rows = [m.emulation_group_to_observable_matrix[k] for k in keys]
np.savez_compressed(
    os.path.join(HERE, "g6_learn_mapping.npz"),
    observables=np.array(keys), group=np.array([r[0] for r in rows]),
    out_start=np.array([r[1].start for r in rows], dtype=np.int64), out_stop=np.array([r[1].stop for r in rows], dtype=np.int64),
    grp_start=np.array([r[2].start for r in rows], dtype=np.int64), grp_stop=np.array([r[2].stop for r in rows], dtype=np.int64),
    shape=np.array(m.shape, dtype=np.int64),
    group_names=np.array(list(GROUPS)), include=np.array([";".join(GROUPS[g][0]) for g in GROUPS]))
print(len(keys), "observables", m.shape, {g: int(sum(r[0] == g for r in rows)) for g in GROUPS})
