"""lib/utils/tf_ops/interpolation operators: the numpy restatement against the reference's own CPU functions
(oracle/_ref, where built) and against golden vectors generated from them; the HIP kernels against both (-m gpu)."""
import numpy as np
import pytest

from conftest import pkg

import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interp_ref.npz")


@pytest.fixture(scope="module")
def interp():
    from oracle import interp_oracle
    from oracle import sa_oracle
    sa_oracle.lib()
    return interp_oracle


@pytest.mark.parametrize("name", ["rand", "few", "grid"])
def test_restatement_matches_reference_golden_vectors(interp, name):
    g = np.load(GOLD)
    d, i = interp.three_nn(g[name + "_xyz1"], g[name + "_xyz2"])
    assert np.array_equal(i, g[name + "_idx"]) and np.array_equal(d, g[name + "_dist"])
    o = interp.three_interpolate(g[name + "_pts"], g[name + "_idx"], g[name + "_w"])
    assert np.array_equal(o, g[name + "_interp"])


def test_restatement_matches_reference_cpu_functions_live(interp):
    if interp.ref_lib() is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference: `make -C oracle ref`)")
    rng = np.random.default_rng(11)
    for b, n, m, c in ((2, 513, 97, 5), (1, 64, 3, 16), (1, 200, 1000, 3)):
        x1 = rng.normal(0, 2, (b, n, 3)).astype(np.float32)
        x2 = np.round(rng.normal(0, 2, (b, m, 3)), 1).astype(np.float32)
        d, i = interp.three_nn(x1, x2)
        rd, ri = interp.ref_three_nn(x1, x2)
        assert np.array_equal(i, ri) and np.array_equal(d, rd)
        p = rng.normal(0, 1, (b, m, c)).astype(np.float32)
        w = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
        assert np.array_equal(interp.three_interpolate(p, i, w), interp.ref_three_interpolate(p, i, w))


def test_kats(interp):
    # three_nn: squared distances, ascending, ties keep index order; fewer than three known points: inf / 0
    x1 = np.array([[[0.0, 0.0, 0.0]]], np.float32)
    x2 = np.array([[[2.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0], [3.0, 0, 0], [1.0, 0, 0]]], np.float32)
    d, i = interp.three_nn(x1, x2)
    assert i.tolist() == [[[1, 2, 4]]] and d.tolist() == [[[1.0, 1.0, 1.0]]]
    d, i = interp.three_nn(x1, x2[:, :2])
    assert i.tolist() == [[[1, 0, 0]]] and d[0, 0, :2].tolist() == [1.0, 4.0] and np.isinf(d[0, 0, 2])
    # three_interpolate / k_interpolate: weighted sums
    p = np.array([[[1.0, 10.0], [2.0, 20.0], [4.0, 40.0]]], np.float32)
    idx = np.array([[[2, 0, 1]]], np.int32)
    w = np.array([[[0.5, 0.25, 0.25]]], np.float32)
    assert interp.three_interpolate(p, idx, w).tolist() == [[[2.75, 27.5]]]
    assert interp.k_interpolate(p, idx, w).tolist() == [[[2.75, 27.5]]]
    assert interp.k_interpolate(p, idx[:, :, :1], w[:, :, :1]).tolist() == [[[2.0, 20.0]]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rand", "few", "grid"])
def test_hip_matches_reference_golden_vectors(gpu, name):
    import torch
    I = pkg("utils.tf_ops.interpolation.tf_interpolate")
    g = np.load(GOLD)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    d, i = I.three_nn(t(g[name + "_xyz1"]), t(g[name + "_xyz2"]))
    assert np.array_equal(i.cpu().numpy(), g[name + "_idx"]) and np.array_equal(d.cpu().numpy(), g[name + "_dist"])
    o = I.three_interpolate(t(g[name + "_pts"]), t(g[name + "_idx"]), t(g[name + "_w"]))
    assert np.array_equal(o.cpu().numpy(), g[name + "_interp"])


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,c,k", [(2, 4096, 1024, 128, 5), (1, 16384, 4096, 64, 3), (3, 100, 5000, 7, 8), (1, 1, 1, 1, 1)])
def test_hip_matches_restatement(gpu, interp, b, n, m, c, k):
    import torch
    I = pkg("utils.tf_ops.interpolation.tf_interpolate")
    rng = np.random.default_rng(n + m + c)
    x1 = rng.uniform(-10, 10, (b, n, 3)).astype(np.float32)
    x2 = rng.uniform(-10, 10, (b, m, 3)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    d, i = I.three_nn(t(x1), t(x2))
    rd, ri = interp.three_nn(x1, x2)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)
    p = rng.normal(0, 1, (b, m, c)).astype(np.float32)
    w = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
    assert np.array_equal(I.three_interpolate(t(p), t(ri), t(w)).cpu().numpy(), interp.three_interpolate(p, ri, w))
    ik = rng.integers(0, m, (b, n, k)).astype(np.int32)
    wk = rng.uniform(0, 1, (b, n, k)).astype(np.float32)
    assert np.array_equal(I.k_interpolate(t(p), t(ik), t(wk)).cpu().numpy(), interp.k_interpolate(p, ik, wk))


@pytest.mark.gpu
def test_argument_errors(gpu):
    import torch
    I = pkg("utils.tf_ops.interpolation.tf_interpolate")
    with pytest.raises(ValueError):
        I.three_nn(torch.zeros(1, 4, 2, device=gpu), torch.zeros(1, 4, 3, device=gpu))
    with pytest.raises(ValueError):
        I.three_interpolate(torch.zeros(1, 4, 2, device=gpu), torch.zeros(1, 5, 2, dtype=torch.int32, device=gpu),
                            torch.zeros(1, 5, 2, device=gpu))
