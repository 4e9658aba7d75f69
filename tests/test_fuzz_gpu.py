"""A short randomised parity sweep over every operator (tests/fuzz_ops.py: random ragged shapes, duplicates, lattice
points, HIP vs oracle, bit-exact / 1e-3 for the MLP).  The long form is run by hand: python tests/fuzz_ops.py 600 <seed>.
Round 1: a 150 s sweep (13 769 cases) found the nsample > 256 truncation of the grid ball query."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_fuzz_sweep_20s(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_ops.py"), "20", "12345"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = "\n".join(r.stdout.strip().splitlines()[-25:])
    assert r.returncode == 0, tail
    assert "0 failures" in tail
