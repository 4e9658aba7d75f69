"""THE PIN: the reference's own device code -- tf_sampling_g.cu / tf_grouping_g.cu compiled
UNMODIFIED for gfx950 into oracle/_ref/libtf_ops_ref_fma.so (oracle/Makefile `ref_gpu`; test-only) -- against the
CPU oracle (oracle/sa_oracle.c) and against the HIP product path, three-way, bit for bit, on the case catalogue of
tests/ref_cases.py (configs[0]/[1]/[2]/[4] shapes, 10 % duplicates, lattice points, empty balls, points a few ulps
either side of a radius, idx == -1).

Also records which FMA policy the oracle's arithmetic decisions A/B (sa_oracle.c header) correspond to: the
scalar-contraction build (`fma`, the model of nvcc's default -fmad=true) must agree everywhere; the no-FMA build and
the gfx950 packed-math build of the very same sources must DISAGREE on the ulp-boundary cases, which shows those
cases do discriminate.
"""
import types

import numpy as np
import pytest

import ref_cases as RC
import ref_gpu
from conftest import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(gpu):
    if not ref_gpu.available("fma"):
        pytest.fail("oracle/_ref/libtf_ops_ref_fma.so missing: run `make -C oracle ref_gpu` in the build container "
                    "(it travels with the snapshot)")
    return ref_gpu.RefOps("fma")


@pytest.fixture(scope="module")
def hip(gpu):
    S, G, I = (pkg("utils.tf_ops.sampling.tf_sampling"), pkg("utils.tf_ops.grouping.tf_grouping"),
               pkg("utils.tf_ops.interpolation.tf_interpolate"))
    ns = types.SimpleNamespace()
    for mod in (I, G, S):
        for k, v in vars(mod).items():
            if callable(v) and not k.startswith("_"):
                setattr(ns, k, v)
    return ns


@pytest.fixture(scope="module")
def cpu(oracle):
    from oracle import interp_oracle
    ns = types.SimpleNamespace(**{k: v for k, v in vars(oracle).items() if callable(v)})
    ns.k_interpolate = interp_oracle.k_interpolate
    return ns


SA = RC.sa_cases()
FULL = RC.full_depth_cases()
F4 = RC.f4_cases()


def _same(a, b, what):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape and x.dtype == y.dtype, "%s output %d: %s %s vs %s %s" % (what, i, x.shape, x.dtype, y.shape, y.dtype)
        if not np.array_equal(x, y):
            bad = np.argwhere(x != y)
            raise AssertionError("%s output %d differs at %d of %d entries, first %s: %r vs %r"
                                 % (what, i, len(bad), x.size, bad[0].tolist(), x[tuple(bad[0])], y[tuple(bad[0])]))


@pytest.mark.parametrize("name", sorted(SA))
def test_sa_path_three_way(gpu, ref, hip, cpu, name):
    case = SA[name]
    r = RC.run_torch(ref, case, gpu)
    _same(RC.run_torch(hip, case, gpu), r, "HIP vs reference device code [%s]" % name)
    _same(RC.run_numpy(cpu, case), r, "oracle vs reference device code [%s]" % name)


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_depth_configs2_configs4(gpu, ref, hip, cpu, name):
    """configs[2] (16384 x 67 channels -> 4096 picks) and configs[4] (65536 points -> 4096 picks, the three layer-1
    bands) at FULL depth, one frame: HIP == reference device code == oracle."""
    case = FULL[name]
    r = RC.run_torch(ref, case, gpu)
    _same(RC.run_torch(hip, case, gpu), r, "HIP vs reference device code [%s]" % name)
    _same(RC.run_numpy(cpu, case), r, "oracle vs reference device code [%s]" % name)


# three_nn / three_interpolate: the oracle for these is pinned to the reference's CPU functions (no FMA on x86-64,
# oracle/interp_oracle.py); the reference's CUDA kernels contract differently, so they are compared separately below
_F4_EXACT = sorted(k for k in F4 if not k.startswith(("three_", "k_interpolate")))


@pytest.mark.parametrize("name", _F4_EXACT)
def test_rank4_ops_three_way(gpu, ref, hip, cpu, name):
    case = F4[name]
    r = RC.run_torch(ref, case, gpu)
    _same(RC.run_torch(hip, case, gpu), r, "HIP vs reference device code [%s]" % name)
    _same(RC.run_numpy(cpu, case), r, "oracle vs reference device code [%s]" % name)


@pytest.mark.parametrize("name", sorted(k for k in F4 if k.startswith(("three_", "k_interpolate"))))
def test_interpolation_against_reference_device_code(gpu, ref, hip, name):
    """The product follows the reference's CPU functions for these ops (that is what oracle/_ref pins, see
    tests/test_interpolate.py); its CUDA kernels may round the last bit differently (FMA contraction).  Indices must
    agree wherever the neighbours are not tied to the last bit; values to 1e-5 relative."""
    case = F4[name]
    r = RC.run_torch(ref, case, gpu)
    h = RC.run_torch(hip, case, gpu)
    for x, y in zip(h, r):
        if x.dtype == np.int32:
            assert (x != y).mean() < 1e-3
        else:
            assert np.allclose(x, y, rtol=1e-5, atol=1e-6)


def test_fma_policy_is_discriminated(gpu, cpu):
    """Decisions A/B: the oracle equals the scalar-contraction build on the ulp-boundary cases (asserted above); the
    other two builds of the same sources must differ from it there -- otherwise the cases would prove nothing."""
    if not (ref_gpu.available("nofma") and ref_gpu.available("hipdefault")):
        pytest.skip("alternative-arithmetic builds not present")
    nofma, hipdef = ref_gpu.RefOps("nofma"), ref_gpu.RefOps("hipdefault")
    names = [k for k in SA if "boundary" in k]
    diff = {"nofma": 0, "hipdefault": 0}
    for k in names:
        o = RC.run_numpy(cpu, SA[k])
        for tag, lib in (("nofma", nofma), ("hipdefault", hipdef)):
            r = RC.run_torch(lib, SA[k], gpu)
            diff[tag] += int(sum((a != b).sum() for a, b in zip(o, r)))
    assert diff["nofma"] > 0, "the no-FMA build agrees with the oracle on every boundary case: cases do not discriminate"
    assert diff["hipdefault"] > 0, "the gfx950 packed-math build agrees on every boundary case"
    # FPS, decision A: fused chain; the no-FMA build picks the other point on the crafted triple
    o = RC.run_numpy(cpu, SA["fps_fma_sensitive"])[0]
    assert not np.array_equal(RC.run_torch(nofma, SA["fps_fma_sensitive"], gpu)[0], o)
    assert np.array_equal(RC.run_torch(hipdef, SA["fps_fma_sensitive"], gpu)[0], o)   # FPS has no packed form
    print("FMA policy: oracle == scalar-contraction build; entries differing on boundary cases:", diff)
