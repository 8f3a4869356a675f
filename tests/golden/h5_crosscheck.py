#!/usr/bin/env python3
"""Cross-check of gpemu.h5io against the HDF5 library (h5py).  Run in the build container with the interpreter that
has h5py:

    /opt/conda/bin/python3.9 tests/golden/h5_crosscheck.py

1. reads tests/golden/h5_native_writer.h5 -- written by gpemu.h5io's OWN writer (python3 tests/golden/h5_crosscheck.py
   --write-native, run with the main interpreter, no h5py) -- through h5py and compares every dataset, group and
   empty group with the expected tree;
2. writes the same tree with h5py to tests/golden/h5_h5py_writer.h5, the fixture gpemu.h5io's own READER is tested on
   (a genuine library-written file: old-style groups, one of them spread over several symbol-table nodes).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NATIVE = os.path.join(HERE, "h5_native_writer.h5")
LIBRARY = os.path.join(HERE, "h5_h5py_writer.h5")


def tree():
    """An mcmc.h5-shaped results dict (ref: mcmc.py:111-125) plus the value kinds the writer supports."""
    rng = np.random.default_rng(20260101)
    return {
        "chain": rng.normal(size=(5, 4, 3)),
        "acceptance_fraction": rng.uniform(size=4),
        "log_prob": rng.normal(size=(5, 4)),
        "autocorrelation_time": None,                              # silx: an empty group
        "design_point": rng.uniform(size=3),
        "experimental_pseudodata": {"y": rng.normal(size=7), "y_err": rng.uniform(size=7)},
        "extras": {
            "counts": np.arange(12, dtype=np.int64).reshape(3, 4),
            "small_ints": np.array([-3, 0, 9], dtype=np.int32),
            "single": np.float32(2.5),
            "scalar": 3.25,
            "label": "pt_ch_alice",
            "empty_array": np.zeros((0, 3)),
            "deep": {"deeper": {"value": np.array([1.0, 2.0])}},
        },
        "many": {f"obs_{i:02d}": np.full(2, float(i)) for i in range(21)},    # > 8 members: several symbol nodes
    }


def compare(got, want, where=""):
    assert set(got) == set(want), (where, sorted(got), sorted(want))
    for key, w in want.items():
        g = got[key]
        if w is None or (isinstance(w, dict) and not w):
            assert isinstance(g, dict) and not g, (where, key, g)
        elif isinstance(w, dict):
            compare(g, w, f"{where}/{key}")
        elif isinstance(w, str):
            assert (g.decode() if isinstance(g, bytes) else g) == w, (where, key, g)
        else:
            w = np.asarray(w)
            g = np.asarray(g)
            assert g.shape == w.shape and g.dtype == w.dtype, (where, key, g.shape, g.dtype, w.shape, w.dtype)
            assert np.array_equal(g, w), (where, key)


def h5py_tree(group):
    import h5py
    out = {}
    for key, item in group.items():
        out[key] = h5py_tree(item) if isinstance(item, h5py.Group) else item[()]
    return out


if __name__ == "__main__":
    if "--write-native" in sys.argv:
        sys.path.insert(0, os.path.join(HERE, "..", "..", "bayesian-inference_amd"))
        os.environ["GPEMU_NO_H5PY"] = "1"
        from gpemu import h5io
        h5io.dicttoh5(tree(), NATIVE)
        print("wrote", NATIVE, os.path.getsize(NATIVE), "bytes")
        sys.exit(0)
    import h5py
    with h5py.File(NATIVE, "r") as f:
        compare(h5py_tree(f), tree())
    print(f"h5py {h5py.__version__} reads {os.path.basename(NATIVE)}: all datasets, groups and empty groups as expected")
    with h5py.File(LIBRARY, "w") as f:
        def put(t, g):
            for k, v in t.items():
                if v is None or (isinstance(v, dict) and not v):
                    g.create_group(k)
                elif isinstance(v, dict):
                    put(v, g.create_group(k))
                elif isinstance(v, str):
                    g.create_dataset(k, data=np.array(v.encode(), dtype=f"S{len(v)}"))
                else:
                    g.create_dataset(k, data=v)
        put(tree(), f)
    print("wrote", LIBRARY, os.path.getsize(LIBRARY), "bytes")
