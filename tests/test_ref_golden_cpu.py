"""CPU-suite half of the pin: oracle/sa_oracle.c must reproduce tests/golden/ref_gpu_pin.npz -- outputs of the
reference's own device code (tf_sampling_g.cu / tf_grouping_g.cu compiled unmodified for gfx950, scalar FMA
contraction = nvcc's default) recorded on an MI355X by tests/golden/make_golden_ref_gpu.py -- bit for bit."""
import os
import types

import numpy as np
import pytest

import ref_cases as RC

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_gpu_pin.npz")
SA = RC.sa_cases()
F4 = {k: v for k, v in RC.f4_cases().items() if not k.startswith(("three_", "k_interpolate"))}
FULL = RC.full_depth_cases()


@pytest.fixture(scope="module")
def gold():
    assert os.path.exists(GOLD), "tests/golden/ref_gpu_pin.npz missing (tests/golden/make_golden_ref_gpu.py writes it)"
    return np.load(GOLD)


@pytest.fixture(scope="module")
def cpu(oracle):
    return types.SimpleNamespace(**{k: v for k, v in vars(oracle).items() if callable(v)})


def _check(cpu, gold, name, case):
    assert [RC.sha1(a) for a in case[2]] == gold[name + "/in_sha1"].tolist(), "inputs of %s regenerated differently" % name
    outs = RC.run_numpy(cpu, case)
    for i, o in enumerate(outs):
        key = "%s/out%d" % (name, i)
        if key in gold.files:
            assert np.array_equal(o, gold[key]), "%s output %d differs from the reference device code" % (name, i)
        assert RC.sha1(o) == gold[name + "/out_sha1"][i], "%s output %d differs from the reference device code" % (name, i)


@pytest.mark.parametrize("name", sorted(SA))
def test_oracle_reproduces_reference_device_code(cpu, gold, name):
    _check(cpu, gold, name, SA[name])


@pytest.mark.parametrize("name", sorted(F4))
def test_oracle_rank4_reproduces_reference_device_code(cpu, gold, name):
    _check(cpu, gold, name, F4[name])


@pytest.mark.parametrize("name", sorted(k for k in FULL if k.startswith("ballD")))
def test_oracle_configs4_bands_reproduce_reference_device_code(cpu, gold, name):
    _check(cpu, gold, name, FULL[name])
