"""Race / hazard stress as a test: tools/stress_determinism.py repeats the training step of every kernel family at sizes that
fill the chip and requires the same bits every time (ELBO and forward-KL gradients).  Both hardware hazards found in
round 3 first showed up as run-to-run differences."""
import os
import subprocess
import sys

import pytest

from __graft_entry__ import ROOT


@pytest.mark.gpu
def test_every_kernel_family_reproduces_its_bits_at_full_occupancy():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_determinism.py"), "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 irreproducible cases" in r.stdout
