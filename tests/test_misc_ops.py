"""SURVEY.md 8f rank 4, second part: query_ball_point_withidx, select_top_k / knn_point,
farthest_point_sample_with_preidx and the interpolation gradients.  CPU: known answers for the oracle and the
reference-generated golden vectors of three_interpolate_grad.  GPU: HIP kernels vs the oracle (bit-exact; the atomics
of the gradients are checked exactly on data whose partial sums are exact, and within 2e-5 otherwise)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg

GOLD = os.path.join(ROOT, "tests", "golden", "interp_ref.npz")


def _t(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


# ------------------------------------------------------------------------------------------------------- CPU
def test_oracle_selection_sort_and_knn_known_answers(oracle):
    d = np.array([[[5, 1, 4, 1, 3, 0.5]]], np.float32)
    oi, o = oracle.select_top_k(3, d)
    # step 0: min 0.5 @5 <-> pos 0 ; step 1: first 1 is already at pos 1 ; step 2: min of (4,1,3,5) = 1 @3 <-> pos 2
    assert o.tolist() == [[[0.5, 1, 1, 4, 3, 5]]] and oi.tolist() == [[[5, 1, 3, 2, 4, 0]]]
    oi, o = oracle.select_top_k(10, d)                                 # k > n: a full sort
    assert o.tolist() == [[[0.5, 1, 1, 3, 4, 5]]] and oi.tolist() == [[[5, 1, 3, 4, 2, 0]]]
    x1 = np.array([[[0, 0, 0], [1, 0, 0], [0, 2, 0], [3, 3, 3]]], np.float32)
    x2 = np.array([[[0.9, 0, 0]]], np.float32)
    val, idx = oracle.knn_point(2, x1, x2)
    assert idx.tolist() == [[[1, 0]]]
    assert val[0, 0, 0] == np.float32(np.float32(1 - np.float32(0.9)) ** 2) and val[0, 0, 1] == np.float32(0.9) ** 2


def test_oracle_withidx_and_preidx_known_answers(oracle):
    xyz1 = np.array([[[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [10, 0, 0]]], np.float32)
    xyz2 = np.array([[[1.4, 0, 0]]], np.float32)
    order = np.array([[[4, 3, 2, 1, 0]]], np.int32)
    idx, cnt = oracle.query_ball_point_withidx(1.5, 3, xyz1, xyz2, order)
    assert cnt.tolist() == [[3]] and idx.tolist() == [[[2, 1, 0]]]     # visiting order, not index order
    idx, cnt = oracle.query_ball_point_withidx(1.5, 4, xyz1, xyz2, order)
    assert cnt.tolist() == [[3]] and idx.tolist() == [[[2, 1, 0, 2]]]  # padded with the first hit
    idx, cnt = oracle.query_ball_point_withidx(0.1, 2, xyz1, xyz2, order)
    assert cnt.tolist() == [[0]] and idx.tolist() == [[[0, 0]]]
    # FPS from the chosen set {0}: farthest is 4 (d2 = 100); then min(d2 to 0, d2 to 4): point 3 -> min(9, 49) = 9
    # wins over point 2 -> 4; then points 1 and 2 tie at 1 (1: min(1, 81, 4), 2: min(4, 64, 1)) -> lower index
    assert oracle.farthest_point_sample_with_preidx(3, xyz1, np.array([[0]], np.int32)).tolist() == [[4, 3, 1]]
    # all points equally far (two duplicates of the preidx point, others at the same distance): the FIRST pick takes
    # the lowest index (serial scan), not the (k mod 1024) order
    p = np.zeros((1, 2050, 3), np.float32)
    p[0, 1:, 0] = 1.0
    p[0, 5, 0] = 0.0
    assert oracle.farthest_point_sample_with_preidx(2, p, np.array([[0, 5]], np.int32))[0, 0] == 1
    # without any preidx the field is 1e38 everywhere: first pick = index 0, then plain FPS
    q = np.random.default_rng(0).normal(0, 1, (1, 300, 3)).astype(np.float32)
    assert np.array_equal(oracle.farthest_point_sample_with_preidx(20, q, np.zeros((1, 0), np.int32)),
                          oracle.farthest_point_sample(20, q))


@pytest.mark.parametrize("name", ["rand", "few", "grid"])
def test_oracle_interpolate_grad_matches_reference_golden_vectors(oracle, name):
    """threeinterpolate_grad_cpu of the reference (tf_interpolate.cpp:158-180) generated these (make_golden_interp.py)"""
    g = np.load(GOLD)
    got = oracle.k_interpolate_grad(g[name + "_pts"], g[name + "_idx"], g[name + "_w"], g[name + "_gout"])
    assert np.array_equal(got, g[name + "_gpts"])


# ------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,ns,r", [(2, 4096, 256, 32, 0.8), (1, 300, 17, 8, 0.5), (2, 1000, 64, 100, 3.0)])
def test_query_ball_point_withidx_matches_oracle(gpu, oracle, b, n, m, ns, r):
    G, syn = pkg("utils.tf_ops.grouping.tf_grouping"), pkg("synthetic")
    rng = np.random.default_rng(n + ns)
    xyz1 = syn.kitti_like_batch(b, n=n)[:, :, :3].copy()
    xyz2 = xyz1[:, :m].copy()
    order = np.stack([np.stack([rng.permutation(n) for _ in range(m)]) for _ in range(b)]).astype(np.int32)
    idx, cnt = G.query_ball_point_withidx(r, ns, _t(xyz1, gpu), _t(xyz2, gpu), _t(order, gpu))
    ridx, rcnt = oracle.query_ball_point_withidx(r, ns, xyz1, xyz2, order)
    assert np.array_equal(cnt.cpu().numpy(), rcnt) and np.array_equal(idx.cpu().numpy(), ridx)
    # with the identity order it is query_ball_point
    ident = np.broadcast_to(np.arange(n, dtype=np.int32), (b, m, n)).copy()
    i2, c2 = G.query_ball_point_withidx(r, ns, _t(xyz1, gpu), _t(xyz2, gpu), _t(ident, gpu))
    i3, c3 = G.query_ball_point(r, ns, _t(xyz1, gpu), _t(xyz2, gpu))
    assert torch.equal(i2, i3) and torch.equal(c2, c3)


@pytest.mark.gpu
@pytest.mark.parametrize("b,m,n,k", [(2, 50, 1000, 16), (1, 3, 16384, 8), (1, 7, 65, 65), (2, 4, 10, 30), (1, 5, 300, 1)])
def test_select_top_k_matches_oracle(gpu, oracle, b, m, n, k):
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(n + k)
    d = rng.integers(0, 50, (b, m, n)).astype(np.float32) * 0.25        # many exact ties
    d[0, 0, : min(n, 5)] = np.inf
    oi, o = G.select_top_k(k, _t(d, gpu))
    roi, ro = oracle.select_top_k(k, d)
    assert np.array_equal(o.cpu().numpy(), ro) and np.array_equal(oi.cpu().numpy(), roi)
    kk = min(k, n)
    assert (ro[:, :, 1:kk] >= ro[:, :, :kk - 1]).all()                  # the head is sorted
    assert np.array_equal(np.sort(roi, -1), np.broadcast_to(np.arange(n), roi.shape))   # a permutation


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,c,k", [(2, 2048, 128, 3, 16), (1, 500, 33, 7, 5)])
def test_knn_point_matches_oracle(gpu, oracle, b, n, m, c, k):
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(n + c)
    x1 = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    x2 = rng.normal(0, 1, (b, m, c)).astype(np.float32)
    x2[:, :4] = x1[:, :4]                                               # zero distances
    val, idx = G.knn_point(k, _t(x1, gpu), _t(x2, gpu))
    rval, ridx = oracle.knn_point(k, x1, x2)
    assert tuple(val.shape) == (b, m, k) and np.array_equal(val.cpu().numpy(), rval) and np.array_equal(idx.cpu().numpy(), ridx)
    assert (ridx[:, :4, 0] == np.arange(4)).all() and (rval[:, :4, 0] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,c,m,m1,dup", [(2, 4096, 3, 256, 64, 400), (1, 3000, 67, 40, 10, 0), (2, 700, 3, 50, 0, 0),
                                            (1, 20000, 3, 30, 5, 2000)])
def test_fps_with_preidx_matches_oracle(gpu, oracle, b, n, c, m, m1, dup):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    rng = np.random.default_rng(n + m1)
    p = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    if dup:
        for i in range(b):
            p[i, rng.integers(0, n, dup)] = p[i, rng.integers(0, n, dup)]
    pre = rng.integers(0, n, (b, m1)).astype(np.int32)
    got = S.farthest_point_sample_with_preidx(m, _t(p, gpu), _t(pre, gpu)).cpu().numpy()
    assert got.shape == (b, m) and np.array_equal(got, oracle.farthest_point_sample_with_preidx(m, p, pre))


@pytest.mark.gpu
def test_fps_with_preidx_first_pick_tie_rule(gpu, oracle):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    p = np.zeros((1, 2050, 3), np.float32)
    p[0, 1:, 0] = 1.0
    p[0, 5, 0] = 0.0
    pre = np.array([[0, 5]], np.int32)
    got = S.farthest_point_sample_with_preidx(6, _t(p, gpu), _t(pre, gpu)).cpu().numpy()
    assert got[0, 0] == 1 and np.array_equal(got, oracle.farthest_point_sample_with_preidx(6, p, pre))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rand", "few", "grid"])
def test_three_interpolate_grad_against_reference_golden_vectors(gpu, name):
    I = pkg("utils.tf_ops.interpolation.tf_interpolate")
    g = np.load(GOLD)
    got = I.three_interpolate_grad(_t(g[name + "_pts"], gpu), _t(g[name + "_idx"], gpu), _t(g[name + "_w"], gpu),
                                   _t(g[name + "_gout"], gpu)).cpu().numpy()
    np.testing.assert_allclose(got, g[name + "_gpts"], rtol=0, atol=3e-5)    # atomics: summation order differs


@pytest.mark.gpu
def test_interpolate_grads_exact_on_dyadic_data_and_adjoint(gpu, oracle):
    I = pkg("utils.tf_ops.interpolation.tf_interpolate")
    rng = np.random.default_rng(4)
    b, m, n, c, k = 2, 40, 900, 11, 5
    pts = rng.integers(-4, 5, (b, m, c)).astype(np.float32)
    idx = rng.integers(0, m, (b, n, k)).astype(np.int32)
    w = rng.integers(0, 8, (b, n, k)).astype(np.float32) * 0.125         # dyadic: every product and sum is exact
    gout = rng.integers(-8, 9, (b, n, c)).astype(np.float32)
    got = I.k_interpolate_grad(_t(pts, gpu), _t(idx, gpu), _t(w, gpu), _t(gout, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.k_interpolate_grad(pts, idx, w, gout))
    got3 = I.three_interpolate_grad(_t(pts, gpu), _t(idx[:, :, :3], gpu), _t(w[:, :, :3], gpu), _t(gout, gpu)).cpu().numpy()
    assert np.array_equal(got3, oracle.k_interpolate_grad(pts, idx[:, :, :3], w[:, :, :3], gout))
    fwd = I.k_interpolate(_t(pts, gpu), _t(idx, gpu), _t(w, gpu)).double()
    assert float((fwd * _t(gout, gpu).double()).sum()) == float((_t(pts, gpu).double() * _t(got, gpu).double()).sum())
    with pytest.raises(ValueError, match="grad_out"):
        I.three_interpolate_grad(_t(pts, gpu), _t(idx[:, :, :3], gpu), _t(w[:, :, :3], gpu), _t(gout[:, :-1], gpu))


@pytest.mark.gpu
def test_copy_blocks_matches_slicing(gpu):
    # sa_copy_blocks: up to four strided block copies in one launch == torch slicing + contiguous()
    N = pkg("utils._native")
    rng = np.random.default_rng(5)
    pc = _t(rng.normal(0, 1, (3, 700, 4)).astype(np.float32), gpu)
    big = _t(rng.normal(0, 1, (3, 512, 37)).astype(np.float32), gpu)
    xyz = torch.full((3, 700, 3), -1.0, device=gpu)
    feat = torch.full((3, 700, 1), -1.0, device=gpu)
    pre = torch.full((3, 200, 37), -1.0, device=gpu)
    tail = torch.full((3, 100, 5), -1.0, device=gpu)
    N.copy_blocks([(pc[:, :, 0:3], xyz, 3, 700, 3), (pc[:, :, 3:], feat, 3, 700, 1),
                   (big[:, 56:256], pre, 3, 200, 37), (big[:, 412:, 30:35], tail, 3, 100, 5)])
    torch.cuda.synchronize()
    assert torch.equal(xyz, pc[:, :, 0:3]) and torch.equal(feat, pc[:, :, 3:])
    assert torch.equal(pre, big[:, 56:256]) and torch.equal(tail, big[:, 412:, 30:35])
    with pytest.raises(ValueError):
        import ctypes
        N.check(N.lib().sa_copy_blocks(5, (ctypes.c_long * 45)(), N.current_stream()), "copy_blocks")
