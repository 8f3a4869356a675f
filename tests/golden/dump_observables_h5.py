#!/usr/bin/env python3
"""Dump the reference's real-data test fixture to a plain .npz (data only, no code).

    /opt/conda/bin/python3.9 tests/golden/dump_observables_h5.py

Reads ref: tests/test_data/observables.h5 (the fixture of ref: tests/test_data_IO.py) with h5py
and stacks it like data_IO.predictions_matrix_from_h5 / data_array_from_h5 do
(ref: data_IO.py:260-297, 345-388): bins of all observables concatenated;
Y = (design points x bins), design = (design points x parameters), data y / y_err = (bins,).
Observables are ordered by plain label sort here (the reference sorts by label fields,
data_IO.py:531-549); the emulator path is invariant to a permutation of the feature columns,
so the fixture is simply "a real 200 x 215 physics matrix with its data vector".
"""
import os

import h5py
import numpy as np

SRC = "/root/reference/tests/test_data/observables.h5"
HERE = os.path.dirname(os.path.abspath(__file__))

with h5py.File(SRC, "r") as f:
    design = np.asarray(f["Design"], dtype=np.float64)
    labels = sorted(f["Prediction"].keys())
    Y = np.concatenate([np.asarray(f["Prediction"][k]["y"], dtype=np.float64).T for k in labels], axis=1)
    y = np.concatenate([np.asarray(f["Data"][k]["y"], dtype=np.float64) for k in labels])
    y_err = np.concatenate([np.asarray(f["Data"][k]["y_err"], dtype=np.float64) for k in labels])
    nbins = np.array([f["Prediction"][k]["y"].shape[0] for k in labels], dtype=np.int64)

print("labels", labels)
print("Y", Y.shape, "design", design.shape, "y", y.shape, "finite", np.isfinite(Y).all())
np.savez_compressed(os.path.join(HERE, "observables_fixture.npz"), Y=Y, design=design, y=y, y_err=y_err,
                    nbins=nbins, labels=np.array(labels))
