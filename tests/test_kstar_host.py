"""CPU test of the matrix-core cross-kernel's host-built operands (csrc/kstar_host.h): the augmented, centred,
fragment-ordered training rows reproduce the squared scaled distance of sklearn's ARD kernels
(ref: emulation.py:497 -> skl kernels.py:1553-1582, 1708-1781) when contracted the way the device does it."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("N,d,k", [(203, 6, 3), (64, 7, 2), (130, 8, 2), (17, 1, 1)])
def test_augmented_product_recovers_the_scaled_distance(tmp_path, N, d, k):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "kstar_host_check"
    subprocess.run([gxx, "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(HERE, "native", "kstar_host_check.cpp"), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe), str(N), str(d), str(k)], check=True, capture_output=True, text=True).stdout
    lines = [ln for ln in out.splitlines() if ln.startswith("kind")]
    assert len(lines) == 2
    for ln in lines:
        m = re.match(r"kind (\d) ksteps (\d) worst_rel_r2 (\S+) at_training_point (\S+) layout (\w+)", ln)
        assert m, ln
        assert int(m.group(2)) == (2 if d + 1 <= 8 else 3)
        assert m.group(5) == "ok", ln
        # cancellation error of the product form: ~ d (range / 2 ls)^2 eps; coordinates here span +-3.5 / 0.3 length scales
        assert float(m.group(3)) < 1e-11, ln
        assert float(m.group(4)) < 1e-11, ln
