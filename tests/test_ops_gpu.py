"""Parity of every HIP operator against the CPU oracle on identical seeded inputs (run with -m gpu).
Bar (BASELINE.json): bit-exact FPS indices and ball-query neighbour sets/counts, bit-exact copies and
distance matrices, grouped-MLP outputs within 1e-3 relative (max|d| / max|ref|) of the fp32 oracle."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
MLP_TOL = 1e-3


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _cloud(rng, b, n, scale=10.0, dup=0):
    p = rng.uniform(-scale, scale, (b, n, 3)).astype(np.float32)
    if dup:
        src = rng.integers(0, n - dup, (b, dup))
        for i in range(b):
            p[i, n - dup:] = p[i, src[i]]
    return p


# ----------------------------------------------------------------------------------- FPS
@pytest.mark.parametrize("b,n,m,dup", [(1, 5, 3, 0), (2, 64, 64, 0), (3, 1000, 100, 0), (2, 1024, 256, 100),
                                       (2, 2048, 512, 400), (2, 4096, 512, 0), (1, 5000, 300, 500),
                                       (2, 16384, 1024, 1600), (1, 777, 777, 0)])
def test_fps_matches_oracle(gpu, oracle, b, n, m, dup):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    rng = np.random.default_rng(n * 7 + m)
    p = _cloud(rng, b, n, dup=dup)
    got = S.farthest_point_sample(m, _t(p, gpu)).cpu().numpy()
    ref = oracle.farthest_point_sample(m, p)
    assert got.dtype == np.int32 and got.shape == (b, m)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("n_all,start,end,m", [(1024, 0, 512, 256), (1024, 512, 1024, 256), (5000, 100, 4196, 300),
                                               (16384, 0, 16384, 512), (20000, 3000, 19384, 200)])
def test_fps_in_place_range_with_fused_centres(gpu, oracle, n_all, start, end, m):
    # sa_fps_ex2: a range slice xyz[:, start:end] of a larger tensor sampled in place (frame stride), the picked points
    # written by the sampler itself, into a column window of a wider centre tensor -- == slice copy + sampler + gather
    N = pkg("utils._native")
    rng = np.random.default_rng(n_all + start + m)
    b, n = 2, end - start
    p = _cloud(rng, b, n_all, dup=n_all // 10)
    t = _t(p, gpu)
    out = torch.full((b, m + 7), -5, dtype=torch.int32, device=gpu)
    ctr = torch.full((b, m + 9, 3), -7.0, dtype=torch.float32, device=gpu)
    st = N.lib().sa_fps_ex2(b, n, 3, m, t.data_ptr() + 12 * start, 3 * n_all, None, out.data_ptr() + 4 * 3, m + 7, start,
                            ctr.data_ptr() + 12 * 4, 3 * (m + 9), N.current_stream())
    assert st == 0
    torch.cuda.synchronize()
    ref = oracle.farthest_point_sample(m, np.ascontiguousarray(p[:, start:end]))
    got = out.cpu().numpy()
    assert np.array_equal(got[:, 3:3 + m], ref + start) and (got[:, :3] == -5).all() and (got[:, 3 + m:] == -5).all()
    c = ctr.cpu().numpy()
    assert np.array_equal(c[:, 4:4 + m], np.take_along_axis(p[:, start:end], ref[..., None].astype(np.int64), 1))
    assert (c[:, :4] == -7).all() and (c[:, 4 + m:] == -7).all()


def test_ffps_in_place_range_with_fused_centres(gpu, oracle):
    # sa_calc_square_dist_self_ws on a range slice read in place + sa_fps_with_distance_ex2 writing the picked points
    N = pkg("utils._native")
    rng = np.random.default_rng(99)
    b, n_all, start, end, c1, m = 2, 1024, 0, 512, 128, 256
    n = end - start
    xyz = _cloud(rng, b, n_all)
    feat = rng.normal(0, 1, (b, n_all, c1)).astype(np.float32)
    tx, tf = _t(xyz, gpu), _t(feat, gpu)
    lib = N.lib()
    dist = torch.empty((b, n, n), dtype=torch.float32, device=gpu)
    ws = torch.empty((lib.sa_calc_square_dist_ws_bytes(b, n, n, 3 + c1, 1) + 3) // 4, dtype=torch.float32, device=gpu)
    st = lib.sa_calc_square_dist_self_ws(b, n, 3, c1, tx.data_ptr() + 12 * start, n_all, tf.data_ptr() + 4 * c1 * start,
                                         n_all, dist.data_ptr(), ws.data_ptr(), N.current_stream())
    assert st == 0
    f = np.concatenate([xyz[:, start:end], feat[:, start:end]], -1)
    ref_d = oracle.calc_square_dist(f, f)
    assert np.array_equal(dist.cpu().numpy(), ref_d)
    out = torch.empty((b, m), dtype=torch.int32, device=gpu)
    ctr = torch.empty((b, m, 3), dtype=torch.float32, device=gpu)
    st = lib.sa_fps_with_distance_ex2(b, n, m, dist.data_ptr(), None, out.data_ptr(), m, start, tx.data_ptr() + 12 * start,
                                      3 * n_all, ctr.data_ptr(), 3 * m, N.current_stream())
    assert st == 0
    torch.cuda.synchronize()
    ref = oracle.farthest_point_sample_with_distance(m, ref_d)
    assert np.array_equal(out.cpu().numpy(), ref + start)
    assert np.array_equal(ctr.cpu().numpy(), np.take_along_axis(xyz[:, start:end], ref[..., None].astype(np.int64), 1))


@pytest.mark.parametrize("nf,nd,mf,md", [(512, 512, 256, 256), (4096, 4096, 512, 512), (2000, 1500, 300, 200)])
def test_fps_dual_launch_equals_the_two_samplers(gpu, oracle, nf, nd, mf, md):
    # sa_fps_dual_ex: the matrix sampler (range [0, nf)) and the coordinate sampler (range [n_all - nd, n_all)) of one layer
    # in ONE launch, indices + centres of both into one tensor each == oracle F-FPS and D-FPS on the two ranges
    N = pkg("utils._native")
    rng = np.random.default_rng(nf + nd + mf)
    b, c1 = 3, 24
    n_all = max(nf, nd) + 77
    xyz = _cloud(rng, b, n_all, dup=50)
    feat = rng.normal(0, 1, (b, n_all, c1)).astype(np.float32)
    tx = _t(xyz, gpu)
    f = np.concatenate([xyz[:, :nf], feat[:, :nf]], -1)
    ref_d = oracle.calc_square_dist(f, f)
    dist = _t(ref_d, gpu)
    ds = n_all - nd
    idx = torch.full((b, mf + md), -1, dtype=torch.int32, device=gpu)
    ctr = torch.full((b, mf + md, 3), -9.0, dtype=torch.float32, device=gpu)
    st = N.lib().sa_fps_dual_ex(b, nf, mf, dist.data_ptr(), idx.data_ptr(), mf + md, 0, tx.data_ptr(), 3 * n_all, ctr.data_ptr(),
                                3 * (mf + md), nd, md, tx.data_ptr() + 12 * ds, 3 * n_all, idx.data_ptr() + 4 * mf, mf + md, ds,
                                ctr.data_ptr() + 12 * mf, 3 * (mf + md), N.current_stream())
    assert st == 0
    torch.cuda.synchronize()
    rf = oracle.farthest_point_sample_with_distance(mf, ref_d)
    rd = oracle.farthest_point_sample(md, np.ascontiguousarray(xyz[:, ds:])) + ds
    ref = np.concatenate([rf, rd], 1)
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(ctr.cpu().numpy(), np.take_along_axis(xyz, ref[..., None].astype(np.int64), 1))


def test_fps_all_points_identical(gpu, oracle):
    # every distance is 0: ties everywhere, the (k mod 1024, k) rule decides every pick
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    p = np.ones((2, 3000, 3), np.float32)
    got = S.farthest_point_sample(50, _t(p, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample(50, p))
    assert (got == 0).all()


def test_fps_tiebreak_and_fma_kats(gpu):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    n = 2048
    p = np.zeros((1, n, 3), np.float32)
    p[0, 7, 0] = 3.0
    p[0, 1030, 0] = -3.0
    assert S.farthest_point_sample(2, _t(p, gpu)).cpu().tolist() == [[0, 1030]]
    o = [-0.6892402172088623, -0.11729754507541656, -0.16030718386173248]
    a = [0.6565782427787781, 0.3208279013633728, -0.18391664326190948]
    bb = [-0.25111478567123413, -0.14090700447559357, 1.1855113506317139]
    p = np.array([[o, a, bb]], np.float32)
    assert S.farthest_point_sample(2, _t(p, gpu)).cpu().tolist() == [[0, 2]]   # fused chain, see KAT


@pytest.mark.parametrize("b,n,c,m", [(2, 300, 5, 40), (1, 2048, 67, 128), (2, 512, 131, 64), (1, 20000, 3, 64),
                                     (2, 16384, 67, 192)])   # last: BASELINE.json configs[2] (F-FPS isolated, 3+64 channels)
def test_fps_generic_channels_and_large_n(gpu, oracle, b, n, c, m):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    rng = np.random.default_rng(c + n)
    p = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    got = S.farthest_point_sample(m, _t(p, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample(m, p))


@pytest.mark.parametrize("b,n,c,m,dup", [(2, 5000, 67, 300, 1500), (2, 40000, 3, 200, 8000), (1, 65536, 3, 96, 0),
                                         (20, 8200, 67, 24, 100), (3, 1500, 67, 64, 700)])
def test_fps_cooperative_kernel_matches_oracle(gpu, oracle, b, n, c, m, dup):
    # several workgroups per frame (fps_coop.hip): duplicated rows put equal maxima into DIFFERENT workgroups, so
    # the explicit (k mod 1024, k div 1024) key decides; b = 20 needs two cooperative launches at 16 WGs/frame
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    rng = np.random.default_rng(n + c + m)
    p = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    if dup:
        for i in range(b):
            p[i, rng.integers(0, n, dup)] = p[i, rng.integers(0, n, dup)]
    got = S.farthest_point_sample(m, _t(p, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample(m, p))


def test_fps_cooperative_kernel_all_identical_and_two_values(gpu, oracle):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    p = np.ones((2, 20000, 3), np.float32)
    assert (S.farthest_point_sample(40, _t(p, gpu)).cpu().numpy() == 0).all()
    # two distinct locations only: every pick after the second is a pure tie-break over thousands of candidates
    q = np.zeros((1, 9000, 67), np.float32)
    q[0, 1::2] = 1.0
    got = S.farthest_point_sample(12, _t(q, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample(12, q))


def test_single_band_api_grid_and_scan_kernels_agree(gpu, oracle, monkeypatch):
    """query_ball_point(_dilated) goes through the grid kernel from n = 512 up; the plain scan kernels (forced here
    by raising the threshold) must give the same bits, and both equal the oracle"""
    G, syn = pkg("utils.tf_ops.grouping.tf_grouping"), pkg("synthetic")
    xyz = syn.kitti_like_batch(2, n=6000)[:, :, :3].copy()
    ctr = xyz[:, ::11].copy()
    a = G.query_ball_point_dilated(0.4, 0.8, 48, _t(xyz, gpu), _t(ctr, gpu))
    c = G.query_ball_point(6.0, 300, _t(xyz, gpu), _t(ctr, gpu))          # nsample beyond the grid kernel's LDS rows,
                                                                           # and balls that really hold > 256 points
    monkeypatch.setattr(G, "GRID_BALL_QUERY_MIN_N", 1 << 30)
    a2 = G.query_ball_point_dilated(0.4, 0.8, 48, _t(xyz, gpu), _t(ctr, gpu))
    c2 = G.query_ball_point(6.0, 300, _t(xyz, gpu), _t(ctr, gpu))
    for x, y in zip(a + c, a2 + c2):
        assert torch.equal(x, y)
    ri, rc = oracle.query_ball_point_dilated(0.4, 0.8, 48, xyz, ctr)
    assert np.array_equal(a[0].cpu().numpy(), ri) and np.array_equal(a[1].cpu().numpy(), rc)
    ri, rc = oracle.query_ball_point(6.0, 300, xyz, ctr)
    assert np.array_equal(c[0].cpu().numpy(), ri) and np.array_equal(c[1].cpu().numpy(), rc)
    assert rc.max() == 300
    # nsample = 256 (the capacity of the grid kernel's lists) with fuller balls: the ordered fallback scan
    d = G.query_ball_point(6.0, 256, _t(xyz, gpu), _t(ctr, gpu))
    ri, rc = oracle.query_ball_point(6.0, 256, xyz, ctr)
    assert np.array_equal(d[0].cpu().numpy(), ri) and np.array_equal(d[1].cpu().numpy(), rc)


@pytest.mark.parametrize("b,n,m,c0,c1,sym", [(2, 512, 512, 3, 64, True), (1, 300, 300, 3, 128, True), (2, 1000, 1000, 3, 32, True),
                                             (2, 384, 640, 3, 64, False), (1, 129, 77, 3, 4, False), (1, 128, 128, 5, 0, True),
                                             (1, 700, 700, 3, 64, True)])
def test_calc_square_dist_packed_form(gpu, oracle, b, n, m, c0, c1, sym):
    """sa_calc_square_dist_split_ws (operands packed once into the LDS image, plain-copy staging) against the oracle,
    bit for bit, incl. ragged sizes, several K stages, one-piece rows and distinct operands"""
    N = pkg("utils._native")
    rng = np.random.default_rng(n + m + c1)
    a0 = rng.normal(0, 1, (b, n, c0)).astype(np.float32)
    a1 = rng.normal(0, 1, (b, n, c1)).astype(np.float32)
    if sym:
        b0, b1 = a0, a1
    else:
        b0 = rng.normal(0, 1, (b, m, c0)).astype(np.float32)
        b1 = rng.normal(0, 1, (b, m, c1)).astype(np.float32)
    ta0, ta1 = _t(a0, gpu), (_t(a1, gpu) if c1 else None)
    tb0, tb1 = (ta0, ta1) if sym else (_t(b0, gpu), (_t(b1, gpu) if c1 else None))
    lib = N.lib()
    ws = torch.empty((lib.sa_calc_square_dist_ws_bytes(b, n, m, c0 + c1, 1 if sym else 0) + 3) // 4, dtype=torch.float32, device=gpu)
    out = torch.full((b, n, m), -7.0, dtype=torch.float32, device=gpu)
    st = lib.sa_calc_square_dist_split_ws(b, n, m, c0, c1, ta0.data_ptr(), ta1.data_ptr() if c1 else None, tb0.data_ptr(),
                                          tb1.data_ptr() if c1 else None, out.data_ptr(), ws.data_ptr(), N.current_stream())
    assert st == 0
    torch.cuda.synchronize()
    ref = oracle.calc_square_dist(np.concatenate([a0, a1], -1), np.concatenate([b0, b1], -1))
    got = out.cpu().numpy()
    assert np.array_equal(got, ref)
    if sym:
        assert np.array_equal(got, got.transpose(0, 2, 1))


def test_fps_forced_generic_kernel_equals_register_kernel(gpu, oracle):
    N = pkg("utils._native")
    rng = np.random.default_rng(5)
    p = _cloud(rng, 2, 3000, dup=300)
    t = _t(p, gpu)
    out = torch.empty((2, 200), dtype=torch.int32, device=gpu)
    temp = torch.empty((2, 3000), dtype=torch.float32, device=gpu)
    st = N.lib().sa_fps_generic(2, 3000, 3, 200, t.data_ptr(), temp.data_ptr(), out.data_ptr(), 0,
                                N.current_stream())
    assert st == 0
    assert np.array_equal(out.cpu().numpy(), oracle.farthest_point_sample(200, p))


@pytest.mark.parametrize("b,n,m", [(1, 4, 4), (2, 100, 30), (2, 512, 256), (1, 1500, 200), (2, 4096, 512)])
def test_fps_with_distance(gpu, oracle, b, n, m):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    rng = np.random.default_rng(n + m)
    f = rng.normal(0, 1, (b, n, 9)).astype(np.float32)
    d = oracle.calc_square_dist(f, f)
    got = S.farthest_point_sample_with_distance(m, _t(d, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample_with_distance(m, d))


@pytest.mark.parametrize("kind,b,n,m", [("fps_ordered", 4, 4096, 512), ("random", 4, 4096, 301), ("random", 7, 2900, 2), ("ties", 13, 2048, 64),
                                        ("fps_ordered", 6, 3000, 1), ("nonmetric", 14, 2000, 257), ("fps_ordered", 2, 8192, 130),
                                        ("random", 13, 2048, 2100)])
def test_fps_with_distance_rows_requested_ahead(gpu, oracle, kind, b, n, m):
    """Matrices beyond 192 MB per call are sampled with the runner-up's row requested a pick ahead (csrc/fps.hip, round 5).
    The picks must not depend on it: points in FPS order (the layer-2 case: nearly every prediction right), random features
    (nearly every prediction wrong: the direct path behind a stale request), constant rows (every arg-max a tie), arbitrary
    non-symmetric rows with negative entries, frame sizes that are no multiple of the workgroup, m = 1 / 2 / odd, and more
    picks than points (the surplus repeats; that shape keeps the loop without the LDS pick list)."""
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    assert b * n * n * 4 > 192 << 20
    rng = np.random.default_rng(n + m + b)
    if kind == "fps_ordered":
        p = _cloud(rng, 1, n, scale=30.0)
        order = oracle.farthest_point_sample(n, p)[0]
        base = p[:, order]                                              # every prefix is the FPS sample of the cloud
        f = np.concatenate([base + rng.normal(0, 1e-3, (b, n, 3)).astype(np.float32), rng.normal(0, 0.05, (b, n, 5)).astype(np.float32)], -1)
        d = oracle.calc_square_dist(f.astype(np.float32), f.astype(np.float32))
    elif kind == "random":
        f = rng.normal(0, 1, (b, n, 9)).astype(np.float32)
        d = oracle.calc_square_dist(f, f)
    elif kind == "ties":
        d = np.full((b, n, n), 2.5, np.float32)
        d[:, np.arange(n), np.arange(n)] = 0.0
    else:
        d = rng.normal(-0.5, 1.0, (b, n, n)).astype(np.float32)
    got = S.farthest_point_sample_with_distance(m, _t(d, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample_with_distance(m, d))


def test_fps_with_distance_negative_rows(gpu, oracle):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    d = np.full((1, 3, 3), -2.0, np.float32)
    assert S.farthest_point_sample_with_distance(3, _t(d, gpu)).cpu().tolist() == [[0, 0, 0]]
    rng = np.random.default_rng(0)
    d = rng.normal(-0.5, 1.0, (2, 700, 700)).astype(np.float32)
    got = S.farthest_point_sample_with_distance(100, _t(d, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample_with_distance(100, d))


# ----------------------------------------------------------------------------------- F-FPS without the matrix
def _run_ffps_fly(gpu, xyz, feat, m, start=0, end=None):
    N = pkg("utils._native")
    b, n_all, _ = xyz.shape
    end = n_all if end is None else end
    n = end - start
    tx, tf = _t(xyz, gpu), _t(feat, gpu)
    out = torch.full((b, m + 3), -7, dtype=torch.int32, device=gpu)
    ctr = torch.zeros((b, m + 3, 3), dtype=torch.float32, device=gpu)
    ws = torch.empty((int(N.lib().sa_ffps_fly_ws_bytes(b, n)) + 7) // 8, dtype=torch.int64, device=gpu)
    st = N.lib().sa_ffps_fly_ex(b, n, feat.shape[2], m, tx.data_ptr() + 12 * start, 3 * n_all, tf.data_ptr() + 4 * feat.shape[2] * start,
                                feat.shape[2] * n_all, ws.data_ptr(), out.data_ptr() + 4 * 2, m + 3, start, ctr.data_ptr() + 12 * 2,
                                3 * (m + 3), N.current_stream())
    return st, out.cpu().numpy(), ctr.cpu().numpy()


@pytest.mark.parametrize("b,n,m,dup", [(3, 4096, 512, 0), (2, 4096, 300, 600), (2, 2048, 256, 100), (3, 1024, 128, 0),
                                       (70, 4096, 64, 0)])          # 70 frames x 4 workgroups: two launches inside the call
def test_ffps_fly_equals_the_matrix_sampler(gpu, oracle, b, n, m, dup):
    # csrc/ffps_fly.hip: farthest_point_sample_with_distance(m, calc_square_dist(concat(xyz, feat))) with every row
    # computed on the fly -- the SAME picks as the oracle's matrix + sampler, ties (duplicated rows) included
    rng = np.random.default_rng(b * 7 + n + m)
    xyz = _cloud(rng, b, n, scale=6.0)
    feat = rng.normal(0, 0.7, (b, n, 64)).astype(np.float32)
    if dup:
        src = rng.integers(0, n - dup, dup)
        xyz[:, n - dup:] = xyz[:, src]
        feat[:, n - dup:] = feat[:, src]
    st, out, ctr = _run_ffps_fly(gpu, xyz, feat, m)
    assert st == 0
    cat = np.concatenate([xyz, feat], -1)
    nref = min(b, 6)                                               # the oracle's matrix is n x n per frame
    ref = oracle.farthest_point_sample_with_distance(m, oracle.calc_square_dist(cat[:nref], cat[:nref]))
    assert np.array_equal(out[:nref, 2:2 + m], ref)
    assert (out[:, :2] == -7).all() and (out[:, 2 + m:] == -7).all()      # nothing outside its columns
    for f in range(nref):
        assert np.array_equal(ctr[f, 2:2 + m], xyz[f][ref[f]])
    if b > nref:                                                   # the frames of the second launch: against the HIP matrix path
        S, M = pkg("utils.tf_ops.sampling.tf_sampling"), pkg("utils.model_util")
        tc = _t(cat[nref:], gpu)
        ref2 = S.farthest_point_sample_with_distance(m, M.calc_square_dist(tc, tc, norm=False)).cpu().numpy()
        assert np.array_equal(out[nref:, 2:2 + m], ref2)


def test_ffps_fly_on_a_range_and_unsupported_shapes(gpu, oracle):
    rng = np.random.default_rng(5)
    xyz = _cloud(rng, 2, 3000, scale=4.0)
    feat = rng.normal(0, 1, (2, 3000, 64)).astype(np.float32)
    st, out, ctr = _run_ffps_fly(gpu, xyz, feat, 100, start=500, end=2548)          # rows 500 .. 2547 of every frame, in place
    assert st == 0
    cat = np.concatenate([xyz, feat], -1)[:, 500:2548]
    ref = oracle.farthest_point_sample_with_distance(100, oracle.calc_square_dist(cat, cat)) + 500
    assert np.array_equal(out[:, 2:102], ref)
    assert np.array_equal(ctr[0, 2:102], xyz[0][ref[0]])
    assert _run_ffps_fly(gpu, xyz, feat, 10, start=0, end=3000)[0] == -3                                  # n not 1024 / 2048 / 4096
    assert _run_ffps_fly(gpu, xyz[:, :1024], rng.normal(0, 1, (2, 1024, 32)).astype(np.float32), 10)[0] == -3   # c1 != 64


def test_sample_layer_with_and_without_the_matrix_gives_the_same_layer(gpu):
    # the 'FS' layer of 3dssd.yaml row 2 through sample_layer both ways: fps_idx [F-FPS | D-FPS] and the centres
    lu = pkg("utils.layers_util")
    rng = np.random.default_rng(9)
    xyz = _t(_cloud(rng, 5, 4096, scale=20.0), gpu)
    feat = _t(rng.normal(0, 0.5, (5, 4096, 64)).astype(np.float32), gpu)
    a = lu.sample_layer(xyz, feat, [-1], ["FS"], [512], None, None, [0.4], side_mode=5, ffps_fly=False)
    b = lu.sample_layer(xyz, feat, [-1], ["FS"], [512], None, None, [0.4], side_mode=5, ffps_fly=True)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[0].shape == (5, 1024)


# ----------------------------------------------------------------------------------- gathers
@pytest.mark.parametrize("c", [1, 3, 4, 64, 67, 256])
def test_gather_and_group_point(gpu, oracle, c):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(c)
    b, n, m, ns = 2, 500, 37, 16
    pts = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m)).astype(np.int32)
    assert np.array_equal(S.gather_point(_t(pts, gpu), _t(idx, gpu)).cpu().numpy(), oracle.gather_point(pts, idx))
    gidx = rng.integers(-1, n, (b, m, ns)).astype(np.int32)      # includes -1 -> zero rows
    got = G.group_point(_t(pts, gpu), _t(gidx, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.group_point(pts, gidx))


@pytest.mark.parametrize("b,n,c,m,ns", [(3, 700, 3, 37, 16),      # 4-rows-per-lane path, rows per frame % 64 != 0
                                         (2, 900, 3, 33, 7),       # rows per frame % 4 != 0 -> generic path
                                         (2, 4096, 64, 1024, 32),  # layer-2 shape: 64-row blocks, one frame per block
                                         (3, 500, 64, 21, 5),      # blocks straddle frames, ragged tail
                                         (2, 512, 128, 100, 32), (2, 512, 256, 60, 16), (5, 300, 4, 19, 9),
                                         (2, 300, 8, 50, 3), (1, 100, 16, 1, 1), (2, 640, 32, 64, 64), (2, 300, 12, 31, 4),
                                         (3, 800, 1, 40, 32), (2, 800, 1, 33, 7)])   # c = 1: packed / generic
def test_group_and_gather_point_shapes_of_the_fast_paths(gpu, oracle, b, n, c, m, ns):
    # round 4: gather.hip's 64-row-block kernels (c = 4..256, a power of two), the packed c = 3 kernel and the generic
    # fallback -- bit-exact copies, -1 rows zero (group_point), frame boundaries inside a block, ragged ends
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(b * 1000 + c * 10 + ns)
    pts = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    gidx = rng.integers(-1, n, (b, m, ns)).astype(np.int32)
    gidx[:, 0, :] = -1                                            # a whole ball of zero rows
    got = G.group_point(_t(pts, gpu), _t(gidx, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.group_point(pts, gidx))
    idx = rng.integers(0, n, (b, m * ns)).astype(np.int32)
    assert np.array_equal(S.gather_point(_t(pts, gpu), _t(idx, gpu)).cpu().numpy(), oracle.gather_point(pts, idx))


# ----------------------------------------------------------------------------------- ball query
def _check_ball(got_idx, got_cnt, ref_idx, ref_cnt):
    assert np.array_equal(got_cnt, ref_cnt)
    assert np.array_equal(got_idx, ref_idx)          # includes padding and zero-filled empty rows


@pytest.mark.parametrize("b,n,m,r,ns", [(1, 6, 1, 0.5, 5), (2, 1000, 100, 1.5, 32), (2, 4096, 512, 2.0, 64),
                                        (1, 3000, 77, 0.3, 16), (2, 513, 9, 50.0, 8), (1, 2000, 50, 3.0, 100)])
def test_query_ball_point(gpu, oracle, b, n, m, r, ns):
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(n + ns)
    xyz1 = _cloud(rng, b, n, scale=8.0, dup=n // 10)
    xyz2 = np.concatenate([xyz1[:, :m // 2], _cloud(rng, b, m - m // 2, scale=12.0)], 1)   # some empty balls
    idx, cnt = G.query_ball_point(r, ns, _t(xyz1, gpu), _t(xyz2, gpu))
    ridx, rcnt = oracle.query_ball_point(r, ns, xyz1, xyz2)
    _check_ball(idx.cpu().numpy(), cnt.cpu().numpy(), ridx, rcnt)


@pytest.mark.parametrize("rmin,rmax,ns", [(0.0, 0.8, 32), (0.8, 1.6, 32), (1.6, 3.2, 64), (0.5, 0.5000001, 4)])
def test_query_ball_point_dilated(gpu, oracle, rmin, rmax, ns):
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(int(rmax * 100))
    b, n, m = 2, 3000, 300
    xyz1 = _cloud(rng, b, n, scale=6.0, dup=300)
    xyz2 = np.concatenate([xyz1[:, :200], _cloud(rng, b, 100, scale=9.0)], 1)
    idx, cnt = G.query_ball_point_dilated(rmin, rmax, ns, _t(xyz1, gpu), _t(xyz2, gpu))
    ridx, rcnt = oracle.query_ball_point_dilated(rmin, rmax, ns, xyz1, xyz2)
    _check_ball(idx.cpu().numpy(), cnt.cpu().numpy(), ridx, rcnt)


def test_ball_query_boundary_kats(gpu):
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    def line(xs):
        p = np.zeros((1, len(xs), 3), np.float32)
        p[0, :, 0] = xs
        return p
    xyz1, xyz2 = line([0, .25, .5, .75, 1, 1.25]), line([0.5])
    idx, cnt = G.query_ball_point(0.5, 5, _t(xyz1, gpu), _t(xyz2, gpu))
    assert cnt.cpu().tolist() == [[3]] and idx.cpu().tolist() == [[[1, 2, 3, 1, 1]]]
    r32 = np.float32(0.2)
    xyz1 = line([r32, np.nextafter(r32, np.float32(0))])
    idx, cnt = G.query_ball_point(0.2, 2, _t(xyz1, gpu), _t(line([0]), gpu))
    assert cnt.cpu().tolist() == [[1]] and idx.cpu().tolist() == [[[1, 1]]]
    xyz1 = line([0, .25, .5, .75, 1, 1.25, .5])
    idx, cnt = G.query_ball_point_dilated(0.5, 0.8, 8, _t(xyz1, gpu), _t(line([0.5]), gpu))
    assert cnt.cpu().tolist() == [[5]] and idx.cpu()[0, 0].tolist() == [0, 2, 4, 5, 6, 0, 0, 0]


def test_ball_query_multi_band_equals_single_band(gpu, oracle):
    import ctypes
    N = pkg("utils._native")
    rng = np.random.default_rng(11)
    b, n, m = 2, 5000, 640
    xyz1 = _cloud(rng, b, n, scale=5.0, dup=500)
    xyz2 = xyz1[:, rng.permutation(n)[:m]].copy()
    radii, nss = [0.4, 0.8, 1.6], [32, 32, 64]
    t1, t2 = _t(xyz1, gpu), _t(xyz2, gpu)
    idx = [torch.empty((b, m, s), dtype=torch.int32, device=gpu) for s in nss]
    cnt = [torch.empty((b, m), dtype=torch.int32, device=gpu) for _ in nss]
    st = N.lib().sa_query_ball_point_multi(
        b, n, m, 3, (ctypes.c_float * 3)(0.0, 0.4, 0.8), (ctypes.c_float * 3)(*radii), (ctypes.c_int * 3)(*nss), 1,
        t1.data_ptr(), t2.data_ptr(), (ctypes.c_void_p * 3)(*[t.data_ptr() for t in idx]),
        (ctypes.c_void_p * 3)(*[t.data_ptr() for t in cnt]), N.current_stream())
    assert st == 0
    for i in range(3):
        ridx, rcnt = oracle.query_ball_point_dilated(0.0 if i == 0 else radii[i - 1], radii[i], nss[i], xyz1, xyz2)
        _check_ball(idx[i].cpu().numpy(), cnt[i].cpu().numpy(), ridx, rcnt)


@pytest.mark.parametrize("n,m,dil", [(1, 3, 1), (63, 5, 0), (64, 9, 1), (65, 130, 1), (512, 256, 0), (513, 33, 1), (1024, 512, 1),
                                     (1025, 77, 0), (2048, 101, 1)])
def test_ball_query_one_query_per_wave_form(gpu, oracle, n, m, dil):
    # n <= 2048 takes ball_query_small_kernel (the layer3 / layer4 shapes of the backbone): every register-step count,
    # ragged last step, bands of 1..64 samples that fill early or never, duplicated points, empty balls
    import ctypes
    N = pkg("utils._native")
    rng = np.random.default_rng(n * 7 + m)
    b = 3
    xyz1 = _cloud(rng, b, n, scale=4.0, dup=n // 8)
    near = xyz1[:, rng.integers(0, n, m - m // 3)] + rng.normal(0, 0.05, (b, m - m // 3, 3)).astype(np.float32)
    near[:, ::5] = xyz1[:, rng.integers(0, n, (m - m // 3 + 4) // 5)]          # centres that ARE points: d2 == 0
    xyz2 = np.concatenate([near, _cloud(rng, b, m // 3, scale=40.0)], 1).astype(np.float32)   # far ones: empty balls
    radii, nss = [0.3, 0.9, 2.5, 40.0], [1, 17, 64, 32]
    lo = [0.0] + radii[:-1]
    t1, t2 = _t(xyz1, gpu), _t(xyz2, gpu)
    idx = [torch.full((b, m, s_), -1, dtype=torch.int32, device=gpu) for s_ in nss]
    cnt = [torch.full((b, m), -1, dtype=torch.int32, device=gpu) for _ in nss]
    st = N.lib().sa_query_ball_point_multi(
        b, n, m, 4, (ctypes.c_float * 4)(*lo), (ctypes.c_float * 4)(*radii), (ctypes.c_int * 4)(*nss), dil,
        t1.data_ptr(), t2.data_ptr(), (ctypes.c_void_p * 4)(*[t.data_ptr() for t in idx]),
        (ctypes.c_void_p * 4)(*[t.data_ptr() for t in cnt]), N.current_stream())
    assert st == 0
    for i in range(4):
        if dil:
            ridx, rcnt = oracle.query_ball_point_dilated(lo[i], radii[i], nss[i], xyz1, xyz2)
        else:
            ridx, rcnt = oracle.query_ball_point(radii[i], nss[i], xyz1, xyz2)
        _check_ball(idx[i].cpu().numpy(), cnt[i].cpu().numpy(), ridx, rcnt)
    assert (cnt[0].cpu().numpy() == 0).any() and (cnt[3].cpu().numpy() == min(32, n)).any()


# ----------------------------------------------------------------------------------- sqdist
@pytest.mark.parametrize("n,m,c", [(5, 7, 3), (64, 64, 67), (300, 200, 131), (512, 512, 16)])
def test_calc_square_dist_bit_exact(gpu, oracle, n, m, c):
    M = pkg("utils.model_util")
    rng = np.random.default_rng(n + c)
    a = rng.normal(0, 2, (2, n, c)).astype(np.float32)
    bb = rng.normal(0, 2, (2, m, c)).astype(np.float32)
    got = M.calc_square_dist(_t(a, gpu), _t(bb, gpu), norm=False).cpu().numpy()
    assert np.array_equal(got, oracle.calc_square_dist(a, bb))


# ----------------------------------------------------------------------------------- fused MLP
def _rand_layers(rng, dims):
    ws = [rng.normal(0, 1.0 / np.sqrt(dims[i]), (dims[i], dims[i + 1])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [rng.normal(0, 0.1, dims[i + 1]).astype(np.float32) for i in range(len(dims) - 1)]
    return ws, bs


def _pad_like_ball_query(idx, cnt):
    """The contract of sa_group_mlp_max's default (compact) mode is the ball query's output format: a row of idx
    holds cnt distinct hits and is padded with its FIRST hit up to nsample (tf_grouping_g.cu:245-248)."""
    idx = idx.copy()
    ns = idx.shape[2]
    pad = np.arange(ns)[None, None, :] >= np.maximum(cnt, 1)[:, :, None]
    return np.where(pad, idx[:, :, :1], idx)


def _run_group_mlp(gpu, xyz, feat, new_xyz, idx, cnt, ws, bs, contiguous=True, flags=0, precision=None, overflow_ok=False,
                   chain=False):
    """One sa_group_mlp_max call through the C ABI.  The fp16 range flag of the call must stay 0 unless overflow_ok (its
    value is left in _run_group_mlp.overflow)."""
    import ctypes
    N = pkg("utils._native")
    Wt = pkg("utils.weights")
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    c = 0 if feat is None else feat.shape[2]
    # one buffer per scale (the library's own host side does the same); contiguous=False: separately allocated
    # layers, which must still work (they take the non-streamed kernels)
    layers = Wt.pack_scale(ws, bs, gpu, precision) if contiguous else [Wt.PackedLayer(w, bb, gpu) for w, bb in zip(ws, bs)]
    flags |= Wt.scale_flags(layers)
    nl = len(layers)
    out = torch.full((b, m, layers[-1].N + 5), -7.0, dtype=torch.float32, device=gpu)   # strided output
    dims = (ctypes.c_int * (nl + 1))(*([c + 3] + [l.N for l in layers]))
    tx, tn, ti, tc = _t(xyz, gpu), _t(new_xyz, gpu), _t(idx, gpu), _t(cnt, gpu)
    tf = _t(feat, gpu) if feat is not None else None
    # chain=True: scratch for the GEMM chain of csrc/mlp_gemm.hip as well (taken by the eligible fp16 scales)
    plan, plan_bytes = N.mlp_plan_ws(b, m, ns, gpu, c, [c + 3] + [l.N for l in layers]) if chain else N.mlp_plan_ws(b, m, ns, gpu)
    ovf = torch.zeros(1, dtype=torch.int32, device=gpu)
    ovf_ptr = ovf.data_ptr()
    st = N.lib().sa_group_mlp_max(b, n, m, ns, c, tx.data_ptr(), tf.data_ptr() if tf is not None else None,
                                  tn.data_ptr(), ti.data_ptr(), tc.data_ptr(), nl, dims,
                                  (ctypes.c_void_p * nl)(*[l.w.data_ptr() for l in layers]),
                                  (ctypes.c_void_p * nl)(*[l.bias.data_ptr() for l in layers]),
                                  out.data_ptr(), layers[-1].N + 5, 2,
                                  plan.data_ptr(), plan_bytes, flags, ovf_ptr, N.current_stream())
    assert st == 0
    torch.cuda.synchronize()
    _run_group_mlp.overflow = int(ovf.item())
    assert overflow_ok or _run_group_mlp.overflow == 0, "fp16 range flag raised on in-range data"
    o = out.cpu().numpy()
    assert (o[:, :, :2] == -7.0).all() and (o[:, :, 2 + layers[-1].N:] == -7.0).all()   # untouched margins
    return o[:, :, 2:2 + layers[-1].N]


@pytest.mark.parametrize("c,ns,dims", [(1, 32, [16, 16, 32]), (1, 64, [32, 32, 64]), (64, 32, [64, 64, 128]),
                                       (64, 64, [64, 96, 128]), (128, 32, [128, 192, 256]),
                                       (256, 16, [256, 256, 512]), (256, 32, [256, 512, 1024]),
                                       (5, 8, [24]), (3, 20, [40, 72]), (0, 48, [16, 32, 48])])
def test_group_mlp_max(gpu, oracle, c, ns, dims):
    rng = np.random.default_rng(c * 100 + ns)
    b, n, m = 2, 600, 45
    xyz = _cloud(rng, b, n, scale=4.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32) if c > 0 else None
    new_xyz = xyz[:, :m] + rng.normal(0, 0.1, (b, m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    cnt[:, ::7] = 0                                           # empty balls -> zero output
    ws, bs = _rand_layers(rng, [c + 3] + dims)
    # arbitrary idx rows: every one of the nsample rows counts (dense plan, flags = 1)
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, idx, cnt, ws, bs, flags=1)
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, idx, cnt, ws, bs)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < MLP_TOL, "relative error %g" % err
    assert (got[cnt == 0] == 0).all()
    # ball-query-format idx rows (padded with the first hit): only the distinct rows are evaluated (default) and
    # the result is BIT-IDENTICAL to evaluating all of them
    pidx = _pad_like_ball_query(idx, cnt)
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, pidx, cnt, ws, bs)
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < MLP_TOL, "relative error %g (compact plan)" % err
    assert (got[cnt == 0] == 0).all()
    assert np.array_equal(got, _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, flags=1))


@pytest.mark.parametrize("c,ns,dims,m", [(1, 8, [16, 16, 32], 45), (1, 16, [32, 32, 64], 45), (1, 20, [16, 16, 32], 33),
                                         (8, 32, [12, 16, 20], 45), (64, 16, [64, 64, 128], 45),
                                         (64, 48, [64, 96, 128], 29), (64, 96, [64, 64, 128], 7),
                                         (64, 32, [40, 50, 100], 300), (1, 64, [32, 32, 64], 1100),
                                         (128, 32, [128, 128, 256], 45), (128, 16, [128, 256, 256], 45),
                                         (128, 64, [128, 192, 256], 20), (128, 8, [100, 130, 250], 33),
                                         (128, 32, [128, 128, 256], 700), (256, 16, [256, 256, 512], 45),
                                         (256, 32, [256, 512, 1024], 45), (256, 16, [256, 256, 512], 700),
                                         (1, 8, [16, 16, 32], 5000), (64, 40, [64, 64, 128], 2100)])   # 10 000 / 4 200 balls: several 4096-ball chunks of the row plan
def test_group_mlp_max_rowwave_shapes(gpu, oracle, c, ns, dims, m):
    # the LDS-resident-weight / register-resident-activation kernels (mlp_rowwave.hip; C = 128 rows take the
    # streamed-weight variant, several passes per workgroup at m = 700): every pooling layout
    # (4, 2, 1 balls per 32-row tile, 2 and 3 tiles per ball), padded channel counts, ragged ball counts and
    # enough balls for several tiles per wave (software-pipelined gathers)
    rng = np.random.default_rng(c * 1000 + ns + m)
    b, n = 2, 600
    xyz = _cloud(rng, b, n, scale=4.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = xyz[:, rng.integers(0, n, m)] + rng.normal(0, 0.1, (b, m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    cnt[:, ::5] = 0
    ws, bs = _rand_layers(rng, [c + 3] + dims)
    # arbitrary idx rows: every one of the nsample rows counts (dense plan, flags = 1)
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, idx, cnt, ws, bs, flags=1)
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, idx, cnt, ws, bs)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < MLP_TOL, "relative error %g" % err
    assert (got[cnt == 0] == 0).all()
    # ball-query-format idx rows (padded with the first hit): only the distinct rows are evaluated (default) and
    # the result is BIT-IDENTICAL to evaluating all of them
    pidx = _pad_like_ball_query(idx, cnt)
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, pidx, cnt, ws, bs)
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < MLP_TOL, "relative error %g (compact plan)" % err
    assert (got[cnt == 0] == 0).all()
    assert np.array_equal(got, _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, flags=1))


@pytest.mark.parametrize("c,ns,dims,m", [(256, 16, [256, 256, 512], 300), (256, 32, [256, 512, 1024], 300),
                                         (128, 32, [128, 128, 256], 300), (128, 32, [128, 192, 256], 700),
                                         (128, 32, [128, 256, 256], 300),
                                         (64, 32, [64, 64, 128], 45), (5, 8, [24], 45), (0, 48, [16, 32, 48], 45)])
def test_group_mlp_max_operand_precisions(gpu, oracle, c, ns, dims, m):
    # both operand precisions of the fused kernels (csrc/mlp.hip "Operand precision") on the same inputs: split bf16
    # (three passes) stays ~1e-5 of the fp32 oracle; fp16 (one pass) stays inside the 1e-3 bar on the wide (layer3,
    # layer4) shapes it is selected for by utils/weights.scale_precision, and is still correct code on any other shape (the
    # looser bound there is why the rule does not select it)
    rng = np.random.default_rng(c * 7 + ns + m)
    b, n = 2, 600
    xyz = _cloud(rng, b, n, scale=4.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32) if c > 0 else None
    new_xyz = xyz[:, rng.integers(0, n, m)] + rng.normal(0, 0.1, (b, m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    pidx = _pad_like_ball_query(idx, cnt)
    ws, bs = _rand_layers(rng, [c + 3] + dims)
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, pidx, cnt, ws, bs)
    Wt = pkg("utils.weights")
    wide = all(w.shape[0] >= Wt.FP16_MIN_K for w in ws)
    assert Wt.scale_precision(ws) == ("fp16" if wide else "bf16x3")
    for precision, bar in (("bf16x3", 5e-5), ("fp16", MLP_TOL if wide else 3e-3)):
        got = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision=precision)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print("precision %s dims %s: %.2e of the fp32 oracle" % (precision, [c + 3] + dims, err))
        assert err < bar, "%s: relative error %g" % (precision, err)
        assert (got[cnt == 0] == 0).all()
        assert np.array_equal(got, _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, flags=1, precision=precision))


@pytest.mark.parametrize("c,ns,dims,m,b", [(256, 16, [256, 256, 512], 300, 2), (256, 32, [256, 512, 1024], 300, 2),
                                           (256, 32, [256, 512, 1024], 257, 3), (128, 32, [128, 128, 256], 700, 2),
                                           (128, 64, [128, 192, 256], 150, 1), (64, 48, [160, 128, 288], 211, 2),
                                           (256, 40, [256, 384, 544], 97, 2), (8, 16, [256, 128, 512], 301, 1)])
def test_group_mlp_gemm_chain(gpu, oracle, c, ns, dims, m, b):
    # the wide scales as three large-tile GEMM launches over packed fp16 intermediates (csrc/mlp_gemm.hip; opt-in): within the
    # fp16 bar of the fp32 oracle, and BIT-IDENTICAL to the one-launch fused kernels of the same precision (same
    # operands, same k order, fp32 accumulation in the matrix cores); ragged tile counts (rows not a multiple of the
    # 128 / 256-row workgroup tiles, channel tiles not a multiple of 8), empty balls, balls of more than 32 rows (split
    # across tiles: atomic max), dense and compact plans
    rng = np.random.default_rng(c * 31 + ns + m)
    n = 600
    xyz = _cloud(rng, b, n, scale=4.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = xyz[:, rng.integers(0, n, m)] + rng.normal(0, 0.1, (b, m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    cnt[:, ::6] = 0
    pidx = _pad_like_ball_query(idx, cnt)
    ws, bs = _rand_layers(rng, [c + 3] + dims)
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, pidx, cnt, ws, bs)
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="fp16", chain=True, flags=16)   # bit 4: the chain
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("gemm chain dims %s: %.2e of the fp32 oracle" % ([c + 3] + dims, err))
    assert err < MLP_TOL, "relative error %g" % err
    assert (got[cnt == 0] == 0).all()
    fused = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="fp16", chain=True)            # default: fused kernels
    assert np.array_equal(got, fused)
    # flags bit 3: the 64-row / streamed kernels instead of the 96-row kernel of csrc/mlp_wide128.hip (which the default
    # call takes for the layer4 shapes): bit-identical as well
    assert np.array_equal(fused, _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="fp16", flags=8))
    # ... and bit 5 forces the 96-row kernel for every shape it supports (first hidden width 256)
    assert np.array_equal(fused, _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="fp16", flags=32))
    # split bf16 never takes the chain (fp16 only): same call, other precision, still correct
    got3 = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="bf16x3", chain=True, flags=16)
    assert np.abs(got3 - ref).max() / np.abs(ref).max() < 5e-5


@pytest.mark.parametrize("c,ns,dims,m", [(128, 32, [128, 128, 256], 300),      # streamed row-wave kernel (layer3 shape)
                                         (256, 16, [256, 256, 512], 300),      # streamed, 4 waves (layer4 scale 0)
                                         (256, 32, [256, 512, 1024], 300),     # 64-row kernel (layer4 scale 1)
                                         (200, 24, [160, 136], 90)])           # generic kernel
@pytest.mark.parametrize("where", ["input", "hidden"])
def test_group_mlp_fp16_overflow_is_flagged_and_bf16x3_is_unaffected(gpu, oracle, c, ns, dims, m, where):
    # VERDICT r2 weak #5 / ADVICE r2: nothing guarded ACTIVATIONS against the fp16 range.  An input feature (where =
    # "input") or a hidden activation (where = "hidden": in-range inputs, weights that amplify them) above 65504 must
    # raise the call's overflow word in the fp16 form; the split-bf16 form of the same call has no range limit, raises
    # nothing and stays within its bar of the fp32 oracle.
    rng = np.random.default_rng(c + ns + m)
    b, n = 2, 600
    xyz = _cloud(rng, b, n, scale=4.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = xyz[:, rng.integers(0, n, m)] + rng.normal(0, 0.1, (b, m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(1, ns + 1, (b, m)).astype(np.int32)
    pidx = _pad_like_ball_query(idx, cnt)
    ws, bs = _rand_layers(rng, [c + 3] + dims)
    if where == "input":
        feat[1, pidx[1, 7, 0], 5] = 7.0e4                      # one feature value of one gathered point
    else:
        ws[0] = ws[0].copy()
        ws[0][:, 3] = np.abs(ws[0][:, 3]) + 40.0               # channel 3 of layer 0 sums ~c x 40 x |x|: far above 65504?
        feat = np.abs(feat) * 20.0                             # |x| ~ 16 -> ~c * 40 * 16 >= 8e4 for c >= 128
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, pidx, cnt, ws, bs)
    assert np.isfinite(ref).all()
    got16 = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="fp16", overflow_ok=True)
    assert _run_group_mlp.overflow == 1, "fp16 form: out-of-range %s not flagged" % where
    del got16                                                  # unspecified by contract
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="bf16x3")
    assert _run_group_mlp.overflow == 0
    assert np.abs(got - ref).max() / np.abs(ref).max() < 5e-5


@pytest.mark.parametrize("c,ns,dims,m", [(128, 32, [128, 128, 256], 300), (256, 16, [256, 256, 512], 300),
                                         (256, 32, [256, 512, 1024], 300), (200, 24, [160, 136], 90)])
@pytest.mark.parametrize("bad", [float("nan"), -float("nan"), float("inf"), -float("inf")])
def test_group_mlp_fp16_nan_and_inf_inputs_are_flagged(gpu, c, ns, dims, m, bad):
    # the range guard reduced with v_pk_max_f16 (maxNum: a NaN half next to a finite one is dropped) and the
    # packed ReLU turns a NaN activation into 0 -- a NaN input became zeros without raising the flag, where the
    # reference would propagate it.  The input guard now compares magnitude bits as integers: NaN and inf of either
    # sign, in a single gathered feature value, raise the word.
    rng = np.random.default_rng(c + ns)
    b, n = 2, 600
    xyz = _cloud(rng, b, n, scale=4.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = xyz[:, rng.integers(0, n, m)] + rng.normal(0, 0.1, (b, m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(1, ns + 1, (b, m)).astype(np.int32)
    pidx = _pad_like_ball_query(idx, cnt)
    ws, bs = _rand_layers(rng, [c + 3] + dims)
    _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="fp16", overflow_ok=True)
    assert _run_group_mlp.overflow == 0
    feat[1, pidx[1, 11, 0], 6] = bad
    _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="fp16", overflow_ok=True)
    assert _run_group_mlp.overflow == 1, "fp16 form: %r input not flagged" % bad


def test_backbone_raises_on_fp16_overflow_and_runs_in_bf16x3(gpu):
    # the same guard one level up: a feature scale that pushes layer3's inputs out of the fp16 range makes
    # SABackbone.raise_if_overflow() raise under the default per-scale precision rule, and precision="bf16x3" runs clean
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    params = dict(syn.random_backbone_params(arch))
    params["layer2/ensemble/bn/gamma"] = params["layer2/ensemble/bn/gamma"] * 1.0e5      # layer3's input features x 1e5
    pts = torch.from_numpy(syn.kitti_like_batch(1, n=16384, first_frame=40)).to(gpu)
    B = pkg("backbone")
    net = B.SABackbone(arch, params, gpu)
    net(pts)
    with pytest.raises(FloatingPointError):
        net.raise_if_overflow()
    net.raise_if_overflow()                                    # the flag was cleared by the raise
    net2 = B.SABackbone(arch, params, gpu, precision="bf16x3")
    xl, fl, _ = net2(pts)
    net2.raise_if_overflow()
    assert torch.isfinite(fl[-1]).all()


@pytest.mark.parametrize("c,nss,dimss", [
    (1, [32, 32, 64], [[16, 16, 32], [16, 16, 32], [32, 32, 64]]),                     # layer1: LDS-resident weights, split bf16
    (64, [32, 32, 64], [[64, 64, 128], [64, 64, 128], [64, 96, 128]]),                 # layer2
    (128, [32, 32, 32], [[128, 128, 256], [128, 192, 256], [128, 256, 256]]),          # layer3: streamed, fp16
    (256, [16, 32], [[256, 256, 512], [256, 512, 1024]]),                              # layer4: two scales -> the per-scale loop
])
def test_group_mlp_max_layer_equals_scale_by_scale(gpu, oracle, c, nss, dimss):
    # sa_group_mlp_max_layer (the three scales of a layer in ONE launch where the shapes are the reference configuration's)
    # == sa_group_mlp_plan + one sa_group_mlp_max per scale, bit for bit, and within the bar of the oracle
    import ctypes
    N, Wt = pkg("utils._native"), pkg("utils.weights")
    rng = np.random.default_rng(c + len(nss))
    b, n, m = 8, 700, 96
    k = len(nss)
    xyz = _cloud(rng, b, n, scale=4.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = xyz[:, rng.integers(0, n, m)] + rng.normal(0, 0.1, (b, m, 3)).astype(np.float32)
    tx, tf, tn = _t(xyz, gpu), _t(feat, gpu), _t(new_xyz, gpu)
    idxs, cnts, layers, refs = [], [], [], []
    for ns, dims in zip(nss, dimss):
        idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
        cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
        idx = _pad_like_ball_query(idx, cnt)
        ws, bs = _rand_layers(rng, [c + 3] + dims)
        idxs.append(_t(idx, gpu)); cnts.append(_t(cnt, gpu)); layers.append(Wt.pack_scale(ws, bs, gpu))
        refs.append(oracle.group_mlp_max(xyz, feat, new_xyz, idx, cnt, ws, bs))
    ctot = sum(d[-1] for d in dimss)
    offs = [sum(d[-1] for d in dimss[:i]) for i in range(k)]
    lib = N.lib()

    def run(fused):
        out = torch.full((b, m, ctot), -3.0, dtype=torch.float32, device=gpu)
        plans = [N.mlp_plan_ws(b, m, ns, gpu) for ns in nss]
        nsa = (ctypes.c_int * k)(*nss)
        st = lib.sa_group_mlp_plan(b, m, k, nsa, (ctypes.c_void_p * k)(*[t.data_ptr() for t in cnts]),
                                   (ctypes.c_void_p * k)(*[p[0].data_ptr() for p in plans]), out.data_ptr(), ctot,
                                   (ctypes.c_int * k)(*offs), (ctypes.c_int * k)(*[d[-1] for d in dimss]), 0, N.current_stream())
        assert st == 0
        flags = [2 | Wt.scale_flags(ls) for ls in layers]
        if fused:
            st = lib.sa_group_mlp_max_layer(
                k, b, n, m, nsa, c, tx.data_ptr(), tf.data_ptr(), tn.data_ptr(),
                (ctypes.c_void_p * k)(*[t.data_ptr() for t in idxs]), (ctypes.c_void_p * k)(*[t.data_ptr() for t in cnts]), 3,
                (ctypes.c_int * (4 * k))(*[v for d in dimss for v in [c + 3] + d]),
                (ctypes.c_void_p * (3 * k))(*[l.w.data_ptr() for ls in layers for l in ls]),
                (ctypes.c_void_p * (3 * k))(*[l.bias.data_ptr() for ls in layers for l in ls]), out.data_ptr(), ctot,
                (ctypes.c_int * k)(*offs), (ctypes.c_void_p * k)(*[p[0].data_ptr() for p in plans]),
                (ctypes.c_ulong * k)(*[p[1] for p in plans]), (ctypes.c_int * k)(*flags), None, N.current_stream())
            assert st == 0
        else:
            for i in range(k):
                st = lib.sa_group_mlp_max(b, n, m, nss[i], c, tx.data_ptr(), tf.data_ptr(), tn.data_ptr(), idxs[i].data_ptr(),
                                          cnts[i].data_ptr(), 3, (ctypes.c_int * 4)(*([c + 3] + dimss[i])),
                                          (ctypes.c_void_p * 3)(*[l.w.data_ptr() for l in layers[i]]),
                                          (ctypes.c_void_p * 3)(*[l.bias.data_ptr() for l in layers[i]]), out.data_ptr(), ctot,
                                          offs[i], plans[i][0].data_ptr(), plans[i][1], flags[i], None, N.current_stream())
                assert st == 0
        torch.cuda.synchronize()
        return out.cpu().numpy()

    a, bb = run(True), run(False)
    assert np.array_equal(a, bb)
    for i in range(k):
        got = a[:, :, offs[i]:offs[i] + dimss[i][-1]]
        assert np.abs(got - refs[i]).max() / np.abs(refs[i]).max() < MLP_TOL


def test_group_mlp_max_separately_allocated_layers(gpu, oracle):
    # layer-3 shape with the three layers in three device buffers: no streamed path, same result
    rng = np.random.default_rng(77)
    b, n, m, ns, c = 2, 600, 45, 32, 128
    xyz = _cloud(rng, b, n, scale=4.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = xyz[:, :m].copy()
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(1, ns + 1, (b, m)).astype(np.int32)
    ws, bs = _rand_layers(rng, [c + 3, 128, 128, 256])
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, idx, cnt, ws, bs, contiguous=False, flags=1)
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, idx, cnt, ws, bs)
    assert np.abs(got - ref).max() / np.abs(ref).max() < MLP_TOL
    pidx = _pad_like_ball_query(idx, cnt)
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, contiguous=False)
    ref = oracle.group_mlp_max(xyz, feat, new_xyz, pidx, cnt, ws, bs)
    assert np.abs(got - ref).max() / np.abs(ref).max() < MLP_TOL


def test_group_mlp_max_identity_layout_probe(gpu):
    # asymmetric probe of the MFMA operand/accumulator mapping: one layer whose weight matrix selects
    # and scales single input channels, so any row/column/transposition mix-up shows up exactly
    rng = np.random.default_rng(3)
    b, n, m, ns, c = 1, 64, 4, 32, 13
    xyz = np.zeros((b, n, 3), np.float32)
    feat = rng.integers(1, 50, (b, n, c)).astype(np.float32)
    new_xyz = np.zeros((b, m, 3), np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = np.full((b, m), ns, np.int32)
    W = np.zeros((c + 3, 40), np.float32)
    for o in range(40):
        W[(o * 5 + 2) % c, o] = float(o + 1)
    got = _run_group_mlp(gpu, xyz, feat, new_xyz, idx, cnt, [W], [np.zeros(40, np.float32)])
    g = np.take_along_axis(feat[:, None].repeat(m, 1), idx[..., None].repeat(c, -1), 2)   # [b,m,ns,c]
    ref = np.stack([g[..., (o * 5 + 2) % c].max(-1) * (o + 1) for o in range(40)], -1)
    assert np.array_equal(got, ref)        # small integers: exact in split-bf16


@pytest.mark.parametrize("rows,K,N,relu", [(100, 128, 64, True), (77, 384, 128, True), (300, 1536, 512, True),
                                           (64, 256, 128, True), (50, 128, 3, False), (33, 7, 5, False)])
def test_dense(gpu, oracle, rows, K, N, relu):
    N_ = pkg("utils._native")
    Wt = pkg("utils.weights")
    rng = np.random.default_rng(K + N)
    x = rng.normal(0, 1, (rows, K)).astype(np.float32)
    w = rng.normal(0, 1 / np.sqrt(K), (K, N)).astype(np.float32)
    bias = rng.normal(0, 0.2, N).astype(np.float32)
    L = Wt.PackedLayer(w, bias, gpu)
    tx = _t(x, gpu)
    y = torch.empty((rows, N), dtype=torch.float32, device=gpu)
    st = N_.lib().sa_dense(rows, K, N, tx.data_ptr(), L.w.data_ptr(), L.bias.data_ptr(), int(relu), y.data_ptr(),
                           N_.current_stream())
    assert st == 0
    ref = oracle.dense(x, w, bias, relu)
    err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < MLP_TOL, "relative error %g" % err


@pytest.mark.parametrize("rows,K,H", [(2048, 256, 128), (70, 100, 64), (33, 16, 40), (1, 300, 128)])
def test_vote_tail_one_launch_equals_the_three_launches(gpu, oracle, rows, K, H):
    # sa_vote_tail == sa_dense (ReLU) -> sa_dense (H -> 3) -> sa_vote_translate, bit for bit, and within the MLP bar of
    # the oracle's vote layer
    N_ = pkg("utils._native")
    Wt = pkg("utils.weights")
    rng = np.random.default_rng(rows + K + H)
    x = rng.normal(0, 1, (rows, K)).astype(np.float32)
    xyz = rng.uniform(-30, 30, (rows, 3)).astype(np.float32)
    w1 = rng.normal(0, 1 / np.sqrt(K), (K, H)).astype(np.float32)
    b1 = rng.normal(0, 0.2, H).astype(np.float32)
    w2 = rng.normal(0, 4 / np.sqrt(H), (H, 3)).astype(np.float32)          # offsets beyond the clip range on some rows
    b2 = rng.normal(0, 0.5, 3).astype(np.float32)
    L1, L2 = Wt.PackedLayer(w1, b1, gpu), Wt.PackedLayer(w2, b2, gpu)
    tx, tp = _t(x, gpu), _t(xyz, gpu)
    lo = (-3.0, -2.0, -3.0)
    lib, st = N_.lib(), N_.current_stream()
    f32 = lambda *sh: torch.full(sh, -77.0, dtype=torch.float32, device=gpu)
    h_a, o_a, out_a, h_b, o_b, out_b = f32(rows, H), f32(rows, 3), f32(rows, 3), f32(rows, H), f32(rows, 3), f32(rows, 3)
    assert lib.sa_dense(rows, K, H, tx.data_ptr(), L1.w.data_ptr(), L1.bias.data_ptr(), 1, h_a.data_ptr(), st) == 0
    assert lib.sa_dense(rows, H, 3, h_a.data_ptr(), L2.w.data_ptr(), L2.bias.data_ptr(), 0, o_a.data_ptr(), st) == 0
    assert lib.sa_vote_translate(rows, tp.data_ptr(), o_a.data_ptr(), *lo, out_a.data_ptr(), st) == 0
    assert lib.sa_vote_tail(rows, K, H, tx.data_ptr(), L1.w.data_ptr(), L1.bias.data_ptr(), L2.w.data_ptr(), L2.bias.data_ptr(),
                            h_b.data_ptr(), o_b.data_ptr(), tp.data_ptr(), *lo, out_b.data_ptr(), st) == 0
    torch.cuda.synchronize()
    for a, b_ in ((h_a, h_b), (o_a, o_b), (out_a, out_b)):
        assert np.array_equal(a.cpu().numpy().view(np.uint32), b_.cpu().numpy().view(np.uint32))
    ref_h = oracle.dense(x, w1, b1, True)
    ref_o = oracle.dense(ref_h, w2, b2, False)
    assert np.abs(h_b.cpu().numpy() - ref_h).max() / np.abs(ref_h).max() < MLP_TOL
    assert np.abs(o_b.cpu().numpy() - ref_o).max() / np.abs(ref_o).max() < MLP_TOL
    o = o_b.cpu().numpy()
    assert np.array_equal(out_b.cpu().numpy(), xyz + np.minimum(np.maximum(o, np.float32(lo)), -np.float32(lo)))
    assert (np.abs(o) > 3.0).any()
    assert lib.sa_vote_tail(rows, K, 160, tx.data_ptr(), L1.w.data_ptr(), L1.bias.data_ptr(), L2.w.data_ptr(), L2.bias.data_ptr(),
                            h_b.data_ptr(), o_b.data_ptr(), tp.data_ptr(), *lo, out_b.data_ptr(), st) == -3


# ----------------------------------------------------------------------------------- API behaviour
def test_argument_errors_match_reference_conditions(gpu):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    x = torch.zeros((1, 8, 3), device=gpu)
    with pytest.raises(ValueError, match="positive npoint"):
        S.farthest_point_sample(0, x)
    with pytest.raises(ValueError, match="inp shape"):
        S.farthest_point_sample(2, x[0])
    with pytest.raises(ValueError, match="num_points,num_points"):
        S.farthest_point_sample_with_distance(2, torch.zeros((1, 4, 5), device=gpu))
    with pytest.raises(ValueError, match="positive radius"):
        G.query_ball_point(0.0, 4, x, x)
    with pytest.raises(ValueError, match="positive nsample"):
        G.query_ball_point(1.0, 0, x, x)
    with pytest.raises(ValueError, match="xyz1 shape"):
        G.query_ball_point(1.0, 4, torch.zeros((1, 8, 4), device=gpu), x)
    with pytest.raises(ValueError, match="min_radius"):
        G.query_ball_point_dilated(-1.0, 1.0, 4, x, x)
    with pytest.raises(ValueError, match="idx shape"):
        G.group_point(x, torch.zeros((2, 3, 4), dtype=torch.int32, device=gpu))


@pytest.mark.parametrize("b,n,m,dup,scale", [(2, 16384, 4096, 0, 30.0), (1, 16384, 700, 3000, 5.0), (2, 9000, 512, 100, 50.0),
                                             (1, 4100, 4100, 0, 1.0), (1, 16384, 64, 16000, 2.0)])
def test_fps_bucket_kernel_matches_oracle(gpu, oracle, b, n, m, dup, scale):
    # the culled kernel directly (also reached through farthest_point_sample for n > 4096): clustered,
    # heavily duplicated and fully-sampled clouds
    N = pkg("utils._native")
    rng = np.random.default_rng(n + m + dup)
    p = _cloud(rng, b, n, scale=scale, dup=dup)
    p[:, :, 1] *= 0.05                                    # flat in y like a LiDAR frame
    t = _t(p, gpu)
    out = torch.empty((b, m), dtype=torch.int32, device=gpu)
    st = N.lib().sa_fps_bucket_ex(b, n, m, t.data_ptr(), out.data_ptr(), m, 0, N.current_stream())
    assert st == 0
    assert np.array_equal(out.cpu().numpy(), oracle.farthest_point_sample(m, p))


def test_fps_bucket_all_identical_and_kitti_frame(gpu, oracle):
    N = pkg("utils._native")
    syn = pkg("synthetic")
    p = np.full((1, 8192, 3), 2.5, np.float32)
    t = _t(p, gpu)
    out = torch.empty((1, 100), dtype=torch.int32, device=gpu)
    assert N.lib().sa_fps_bucket_ex(1, 8192, 100, t.data_ptr(), out.data_ptr(), 100, 0, N.current_stream()) == 0
    assert (out.cpu().numpy() == 0).all()
    pts = np.ascontiguousarray(syn.kitti_like_batch(2, first_frame=55, dup_fraction=0.1)[:, :, :3])
    t = _t(pts, gpu)
    out = torch.empty((2, 4096), dtype=torch.int32, device=gpu)
    assert N.lib().sa_fps_bucket_ex(2, 16384, 4096, t.data_ptr(), out.data_ptr(), 4096, 0, N.current_stream()) == 0
    assert np.array_equal(out.cpu().numpy(), oracle.farthest_point_sample(4096, pts))


@pytest.mark.parametrize("n,c1", [(4096, 64), (512, 128), (700, 64), (130, 8), (257, 72)])
def test_calc_square_dist_split_pieces(gpu, oracle, n, c1):
    # the SA layer's own call: rows given as [xyz | features] without the concat (sa_calc_square_dist_split), the
    # symmetric upper-triangle path of the second-form kernel incl. ragged edge tiles and two K stages (3 + 128)
    import ctypes
    N = pkg("utils._native")
    rng = np.random.default_rng(n + c1)
    xyz = rng.normal(0, 20.0, (2, n, 3)).astype(np.float32)
    feat = rng.normal(0, 1.0, (2, n, c1)).astype(np.float32)
    tx, tf = _t(xyz, gpu), _t(feat, gpu)
    out = torch.empty((2, n, n), dtype=torch.float32, device=gpu)
    st = N.lib().sa_calc_square_dist_split(2, n, n, 3, c1, tx.data_ptr(), tf.data_ptr(), tx.data_ptr(), tf.data_ptr(),
                                           out.data_ptr(), N.current_stream())
    assert st == 0
    got = out.cpu().numpy()
    cat = np.concatenate([xyz, feat], -1)
    ref = oracle.calc_square_dist(cat, cat)
    assert np.array_equal(got, ref)
    assert np.array_equal(got, got.transpose(0, 2, 1))


@pytest.mark.parametrize("n,c", [(300, 7), (512, 131), (1000, 67), (4096, 35)])
def test_calc_square_dist_symmetric_path(gpu, oracle, n, c):
    # a is b (the F-FPS call): upper-triangle kernel with mirrored tiles, still bit-exact and bitwise symmetric
    M = pkg("utils.model_util")
    rng = np.random.default_rng(n + c)
    a = rng.normal(0, 1.5, (2, n, c)).astype(np.float32)
    ta = _t(a, gpu)
    got = M.calc_square_dist(ta, ta, norm=False).cpu().numpy()
    assert np.array_equal(got, oracle.calc_square_dist(a, a))
    assert np.array_equal(got, got.transpose(0, 2, 1))


def _run_grid_bq(gpu, xyz1, xyz2, rmins, rmaxs, nss, dilated):
    import ctypes
    N = pkg("utils._native")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    nb = len(rmaxs)
    t1, t2 = _t(xyz1, gpu), _t(xyz2, gpu)
    idx = [torch.full((b, m, s), -5, dtype=torch.int32, device=gpu) for s in nss]
    cnt = [torch.full((b, m), -5, dtype=torch.int32, device=gpu) for _ in nss]
    ws = torch.empty((N.lib().sa_query_ball_point_grid_ws_bytes(b, n, m) + 3) // 4, dtype=torch.int32, device=gpu)
    st = N.lib().sa_query_ball_point_grid(
        b, n, m, nb, (ctypes.c_float * nb)(*rmins), (ctypes.c_float * nb)(*rmaxs), (ctypes.c_int * nb)(*nss), int(dilated),
        t1.data_ptr(), t2.data_ptr(), (ctypes.c_void_p * nb)(*[t.data_ptr() for t in idx]),
        (ctypes.c_void_p * nb)(*[t.data_ptr() for t in cnt]), ws.data_ptr(), N.current_stream())
    assert st == 0
    return [t.cpu().numpy() for t in idx], [t.cpu().numpy() for t in cnt]


@pytest.mark.parametrize("n,m,scale,radii,nss,dil", [
    (5000, 640, 5.0, [0.4, 0.8, 1.6], [32, 32, 64], True),
    (16384, 1000, 30.0, [0.2, 0.4, 0.8], [32, 32, 64], True),
    (3000, 300, 2.0, [1.5], [16], False),
    (2048, 256, 1.0, [0.3, 5.0], [8, 100], True),          # second band larger than the whole cloud, nsample > 64
    (700, 90, 3.0, [0.5, 1.0, 2.0, 3.0], [4, 8, 16, 32], True),
])
def test_grid_ball_query_matches_oracle(gpu, oracle, n, m, scale, radii, nss, dil):
    rng = np.random.default_rng(n + m)
    b = 2
    xyz1 = _cloud(rng, b, n, scale=scale, dup=n // 10)
    xyz1[:, :, 1] *= 0.1
    inside = xyz1[:, rng.permutation(n)[:m - m // 4]]
    outside = _cloud(rng, b, m // 4, scale=scale * 1.6)                   # centres outside the cloud's bounding box
    xyz2 = np.concatenate([inside, outside], 1).astype(np.float32)
    rmins = [0.0] + radii[:-1]
    idx, cnt = _run_grid_bq(gpu, xyz1, xyz2, rmins, radii, nss, dil)
    for i in range(len(radii)):
        if dil:
            ridx, rcnt = oracle.query_ball_point_dilated(rmins[i], radii[i], nss[i], xyz1, xyz2)
        else:
            ridx, rcnt = oracle.query_ball_point(radii[i], nss[i], xyz1, xyz2)
        _check_ball(idx[i], cnt[i], ridx, rcnt)


@pytest.mark.parametrize("variant,radii,nss", [("dense", [0.2, 0.4, 0.8], [32, 32, 64]),      # every band of every query overflows
                                                ("rings64", [0.2, 0.4, 0.8], [32, 32, 64]),    # near-field balls overflow, far ones do not
                                                ("rings64", [0.4, 0.8, 1.6], [32, 32, 64]),
                                                ("dense", [0.5, 3.0], [192, 5]),               # the largest nsample the grid kernel takes
                                                ("rings64", [1.0], [200])])                    # beyond it: the plain scan kernels
def test_grid_ball_query_bands_with_more_hits_than_the_lds_list(gpu, oracle, variant, radii, nss):
    # round 4: a band whose hits exceed its 256-entry LDS list is cut down IN PLACE to its nsample smallest indices
    # (bisection on ballots) instead of being redone by an ordered scan of all n points -- same rows, same counts
    syn = pkg("synthetic")
    n, m = 16384, 384
    xyz1 = np.stack([syn.frame_of(variant, 11 + i, n)[:, :3] for i in range(2)])
    rng = np.random.default_rng(len(radii) + nss[0])
    xyz2 = np.ascontiguousarray(np.stack([x[rng.permutation(n)[:m]] for x in xyz1]))
    rmins = [0.0] + radii[:-1]
    idx, cnt = _run_grid_bq(gpu, xyz1, xyz2, rmins, radii, nss, True)
    over = 0
    for i in range(len(radii)):
        ridx, rcnt = oracle.query_ball_point_dilated(rmins[i], radii[i], nss[i], xyz1, xyz2)
        _check_ball(idx[i], cnt[i], ridx, rcnt)
        over += int((rcnt >= nss[i]).sum())
    assert over > 0                                               # the case really has full balls


@pytest.mark.parametrize("rmins,rmaxs,nss,dil", [
    ([0.0, 0.3, 0.2], [0.5, 0.9, 0.4], [16, 32, 8], True),        # dilated bands that overlap and leave gaps: the per-band bit test
    ([0.1, 0.4], [0.4, 1.2], [32, 64], True),                     # contiguous but not from 0: not the one-hot form either
    ([0.0, 0.0, 0.0], [0.3, 0.6, 1.2], [16, 32, 64], False),      # plain (nested) balls: a key carries several band bits
    ([0.0, 0.4], [0.4, 1.6], [64, 128], True),                    # sum(nsample) = 192: the most the sorting form takes
    ([0.0, 0.4], [0.4, 1.6], [64, 129], True),                    # 193: the list form
    ([0.0, 0.2, 0.4, 0.8], [0.2, 0.4, 0.8, 1.6], [8, 16, 32, 64], True),      # four bands
])
@pytest.mark.parametrize("variant", ["default", "rings64", "dense"])
def test_grid_ball_query_sorting_form_band_shapes(gpu, oracle, variant, rmins, rmaxs, nss, dil):
    # round 6 (ballquery_grid.hip, bq_grid_sort_kernel): one key list per query (index << 4 | band mask), sorted in
    # registers; a band's slot is the prefix count of its bit; overflowing lists are cut by bisection.  Band geometries the
    # backbone does not use, on sparse / ring-structured / all-full frames (long walks, cuts, 2- and 4-register sorts);
    # m = 333: ragged last wave.  Reference: tf_grouping_g.cu:215-255,308-357 through the oracle.
    syn = pkg("synthetic")
    n, m = 16384, 333
    xyz1 = np.stack([syn.frame_of(variant, 31 + i, n)[:, :3] for i in range(2)])
    rng = np.random.default_rng(len(nss) * 7 + nss[-1])
    xyz2 = np.ascontiguousarray(np.stack([x[rng.permutation(n)[:m]] for x in xyz1]))
    xyz2[:, ::5] += rng.normal(0, 0.05, xyz2[:, ::5].shape).astype(np.float32)      # some centres that are not points
    idx, cnt = _run_grid_bq(gpu, xyz1, xyz2, rmins, rmaxs, nss, dil)
    for i in range(len(nss)):
        if dil:
            ridx, rcnt = oracle.query_ball_point_dilated(rmins[i], rmaxs[i], nss[i], xyz1, xyz2)
        else:
            ridx, rcnt = oracle.query_ball_point(rmaxs[i], nss[i], xyz1, xyz2)
        _check_ball(idx[i], cnt[i], ridx, rcnt)


@pytest.mark.parametrize("variant", ["default", "rings64", "dense"])
def test_grid_ball_query_layer1_shape_equals_the_scan_kernel(gpu, variant):
    # the layer-1 call of 3dssd.yaml as the backbone makes it (4096 D-FPS centres of 16384 points, dilated 0.2 / 0.4 / 0.8,
    # nsample 32 / 32 / 64, 8 frames = the XCD-aware mapping, 8 consecutive queries per wave): idx and cnt bit-equal to the
    # brute-force scan kernel (ballquery.hip), which the oracle tests pin at small sizes
    import ctypes
    N, syn = pkg("utils._native"), pkg("synthetic")
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    lib = N.lib()
    b, n, m, nb = 8, 16384, 4096, 3
    xyz = torch.from_numpy(np.stack([syn.frame_of(variant, 50 + f, n)[:, :3] for f in range(b)])).to(gpu).contiguous()
    ctr = S.gather_point(xyz, S.farthest_point_sample(m, xyz)).contiguous()
    nss = [32, 32, 64]
    args = (b, n, m, nb, (ctypes.c_float * nb)(0.0, 0.2, 0.4), (ctypes.c_float * nb)(0.2, 0.4, 0.8), (ctypes.c_int * nb)(*nss), 1,
            xyz.data_ptr(), ctr.data_ptr())
    out = []
    for which in ("grid", "scan"):
        idx = [torch.full((b, m, s), -7, dtype=torch.int32, device=gpu) for s in nss]
        cnt = [torch.full((b, m), -7, dtype=torch.int32, device=gpu) for _ in nss]
        ptrs = ((ctypes.c_void_p * nb)(*[t.data_ptr() for t in idx]), (ctypes.c_void_p * nb)(*[t.data_ptr() for t in cnt]))
        if which == "grid":
            ws = torch.empty((lib.sa_query_ball_point_grid_ws_bytes(b, n, m) + 3) // 4, dtype=torch.int32, device=gpu)
            assert lib.sa_query_ball_point_grid(*args, *ptrs, ws.data_ptr(), N.current_stream()) == 0
        else:
            assert lib.sa_query_ball_point_multi(*args, *ptrs, N.current_stream()) == 0
        torch.cuda.synchronize()
        out.append((idx, cnt))
    for i in range(nb):
        assert torch.equal(out[0][1][i], out[1][1][i]), "cnt of band %d" % i
        assert torch.equal(out[0][0][i], out[1][0][i]), "idx of band %d" % i
    if variant != "default":
        assert int((out[0][1][2] >= 64).sum()) > 0                     # full balls (cut lists) really occur


@pytest.mark.parametrize("b,n,m", [(8, 3000, 64), (16, 2500, 96), (8, 2100, 30), (8, 2200, 40)])
def test_grid_ball_query_frame_to_xcd_mapping(gpu, oracle, b, n, m):
    # b a multiple of 8 and ceil(m / 4) workgroups per frame a multiple of 8: the query kernel remaps (frame, workgroup) so
    # that frame f runs on XCD f % 8 (ballquery_grid.hip); every frame different, every query still answered for its own
    # frame.  m = 30: a ragged last workgroup under the remap; m = 40 (10 workgroups per frame): the plain mapping.
    rng = np.random.default_rng(b * 1000 + m)
    xyz1 = _cloud(rng, b, n, scale=3.0, dup=n // 10)
    xyz1[:, :, 1] *= 0.1
    xyz2 = np.ascontiguousarray(xyz1[:, rng.permutation(n)[:m]] + rng.normal(0, 0.05, (b, m, 3)).astype(np.float32))
    radii, nss = [0.3, 0.8, 2.0], [16, 32, 64]
    rmins = [0.0] + radii[:-1]
    idx, cnt = _run_grid_bq(gpu, xyz1, xyz2, rmins, radii, nss, True)
    for i in range(3):
        ridx, rcnt = oracle.query_ball_point_dilated(rmins[i], radii[i], nss[i], xyz1, xyz2)
        _check_ball(idx[i], cnt[i], ridx, rcnt)


@pytest.mark.parametrize("b,n,c1", [(8, 2048, 16), (16, 1920, 5), (8, 1024, 16)])
def test_distance_matrix_frame_to_xcd_mapping(gpu, oracle, b, n, c1):
    # b a multiple of 8 and T (T + 1) / 2 tiles a multiple of 8 (n = 1920, 2048, 4096): the packed symmetric kernel remaps
    # (frame, tile) so that frame f is computed on XCD f % 8, with a per-frame rotation of the tile order (sqdist.hip);
    # n = 1024 (36 tiles) keeps the plain mapping.  Every frame different: a wrong frame or a missed tile shows.
    N = pkg("utils._native")
    rng = np.random.default_rng(b + n)
    xyz = _cloud(rng, b, n, scale=5.0)
    feat = rng.normal(0, 1, (b, n, c1)).astype(np.float32)
    tx, tf = _t(xyz, gpu), _t(feat, gpu)
    lib = N.lib()
    ws = torch.empty((lib.sa_calc_square_dist_ws_bytes(b, n, n, 3 + c1, 1) + 3) // 4, dtype=torch.float32, device=gpu)
    dist = torch.full((b, n, n), -1.0, dtype=torch.float32, device=gpu)
    assert lib.sa_calc_square_dist_self_ws(b, n, 3, c1, tx.data_ptr(), n, tf.data_ptr(), n, dist.data_ptr(), ws.data_ptr(),
                                           N.current_stream()) == 0
    f = np.concatenate([xyz, feat], -1)
    got = dist.cpu().numpy()
    for i in range(b):                                                  # frame by frame: bounded host memory
        assert np.array_equal(got[i], oracle.calc_square_dist(f[i:i + 1], f[i:i + 1])[0]), "frame %d" % i


def test_grid_ball_query_overflow_falls_back_to_full_scan(gpu, oracle):
    # a dense clump: > 512 candidates inside one ball -> the in-kernel ordered full scan path
    rng = np.random.default_rng(77)
    n, m = 4096, 64
    xyz1 = rng.normal(0, 0.05, (1, n, 3)).astype(np.float32)
    xyz1[:, 3000:] += 20.0
    xyz2 = np.ascontiguousarray(xyz1[:, :m])
    idx, cnt = _run_grid_bq(gpu, xyz1, xyz2, [0.0, 0.1], [0.1, 0.5], [64, 32], True)
    for i, (a, r, s) in enumerate([(0.0, 0.1, 64), (0.1, 0.5, 32)]):
        ridx, rcnt = oracle.query_ball_point_dilated(a, r, s, xyz1, xyz2)
        _check_ball(idx[i], cnt[i], ridx, rcnt)


def test_group_mlp_layer4_scale0_eight_wave_form_matches_the_four_wave_form(gpu, oracle):
    # 259 -> 256 -> 256 -> 512 in fp16: from 2048 nominal 32-row tiles on (b * m * ns / 32) the streamed-weight kernel of
    # csrc/mlp_rowwave.hip runs eight waves per workgroup instead of four.  16 frames take the eight-wave form, each
    # half of them alone (8 frames: 1024 tiles) the four-wave form: same rows, same arithmetic -> identical bits; and
    # both within the bar of the fp32 oracle.
    rng = np.random.default_rng(4160)
    b, n, m, ns, c = 16, 512, 256, 16, 256
    xyz = _cloud(rng, b, n, scale=6.0)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = xyz[:, :m] + rng.normal(0, 0.1, (b, m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    cnt[:, ::9] = 0
    pidx = _pad_like_ball_query(idx, cnt)
    ws, bs = _rand_layers(rng, [c + 3, 256, 256, 512])
    assert b * m * ns // 32 >= 2048 > (b // 2) * m * ns // 32
    both = _run_group_mlp(gpu, xyz, feat, new_xyz, pidx, cnt, ws, bs, precision="fp16")
    for h in range(2):
        s = slice(h * 8, h * 8 + 8)
        half = _run_group_mlp(gpu, xyz[s], feat[s], new_xyz[s], pidx[s], cnt[s], ws, bs, precision="fp16")
        assert np.array_equal(both[s], half), "frames %d.. differ between the 8-wave and the 4-wave form" % (h * 8)
    ref = oracle.group_mlp_max(xyz[:4], feat[:4], new_xyz[:4], pidx[:4], cnt[:4], ws, bs)
    err = np.abs(both[:4] - ref).max() / np.abs(ref).max()
    assert err < MLP_TOL, "relative error %g" % err
    assert (both[cnt == 0] == 0).all()


@pytest.mark.parametrize("rows,K,N,relu", [(25000, 384, 128, True), (12300, 768, 256, True), (6200, 1536, 512, False),
                                           (66000, 128, 64, True)])
def test_dense_128_row_blocks_equal_the_32_row_kernel(gpu, oracle, rows, K, N, relu):
    # sa_dense takes 128-row x 128-column blocks (csrc/mlp.hip dense128_kernel) when K is a multiple of 192, N of 128 and
    # there are >= 192 blocks -- the aggregation layers of a 32-frame replay.  Same arithmetic as the 32-row kernel: the
    # same rows in pieces too small for the block form (< 192 blocks) must give the same bits; rows is not a multiple of
    # 128 (ragged last block) and the output is checked against the oracle on a sample.
    N_ = pkg("utils._native")
    Wt = pkg("utils.weights")
    # (the narrow 128 -> 64 aggregation layer takes 128-row x 64-column blocks from 512 blocks on)
    wide = N % 128 == 0
    assert (K % 192 == 0 and -(-rows // 128) * (N // 128) >= 192) if wide else (K % 128 == 0 and N % 64 == 0 and -(-rows // 128) * (N // 64) >= 512)
    rng = np.random.default_rng(rows + K)
    x = rng.normal(0, 1, (rows, K)).astype(np.float32)
    w = rng.normal(0, 1 / np.sqrt(K), (K, N)).astype(np.float32)
    bias = rng.normal(0, 0.2, N).astype(np.float32)
    L = Wt.PackedLayer(w, bias, gpu)
    tx = _t(x, gpu)
    lib, st = N_.lib(), N_.current_stream()
    y = torch.full((rows + 1, N), -5.0, dtype=torch.float32, device=gpu)       # one guard row behind the output
    assert lib.sa_dense(rows, K, N, tx.data_ptr(), L.w.data_ptr(), L.bias.data_ptr(), int(relu), y.data_ptr(), st) == 0
    piece = (191 // (N // 128)) * 128 if wide else (511 // (N // 64)) * 128    # too few blocks: the 32-row kernel
    y2 = torch.full((rows + 1, N), -5.0, dtype=torch.float32, device=gpu)
    for a in range(0, rows, piece):
        n = min(piece, rows - a)
        assert lib.sa_dense(n, K, N, tx.data_ptr() + 4 * K * a, L.w.data_ptr(), L.bias.data_ptr(), int(relu),
                            y2.data_ptr() + 4 * N * a, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and bool((y[rows] == -5.0).all())
    sample = np.r_[0:300, rows - 300:rows]
    ref = oracle.dense(x[sample], w, bias, relu)
    err = np.abs(y.cpu().numpy()[sample] - ref).max() / np.abs(ref).max()
    assert err < MLP_TOL, "relative error %g" % err


def test_multi_workgroup_sampler_that_loses_a_partner_raises_a_sticky_error_instead_of_trapping(gpu, oracle):
    # VERDICT r4 weak #9 / ADVICE r4: the partner workgroups of a frame (csrc/fps_coop.hip) poll each other; until round 4
    # a poll that ran out executed __builtin_trap() -- a dead process.  Now: sticky error word, the launch ends, the next
    # sampler call returns -4, sa_coop_error_state(1) reads and clears it, and the sampler works again.
    N = pkg("utils._native")
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    lib = N.lib()
    assert lib.sa_coop_error_state(1) == 0 or lib.sa_coop_error_state(0) == 0
    rng = np.random.default_rng(5)
    p = rng.normal(0, 1, (2, 40000, 3)).astype(np.float32)          # 16 workgroups per frame
    x = _t(p, gpu)
    good = S.farthest_point_sample(24, x).cpu().numpy()
    assert np.array_equal(good, oracle.farthest_point_sample(24, p)) and lib.sa_coop_error_state(0) == 0
    temp = torch.empty((2, 40000), dtype=torch.float32, device=gpu)
    out = torch.zeros((2, 24), dtype=torch.int32, device=gpu)
    st = lib.sa_debug_fps_coop_orphan(2, 40000, 3, 24, x.data_ptr(), temp.data_ptr(), out.data_ptr(), 20000,
                                      torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()                                         # the launch ENDS (bounded polls), the process lives
    assert np.array_equal(out[0].cpu().numpy(), good[0])             # the complete frame is untouched by its neighbour's failure
    assert lib.sa_coop_error_state(0) & 1
    with pytest.raises(RuntimeError, match="partner workgroups"):
        S.farthest_point_sample(24, x)                               # sticky: refused until cleared
    assert lib.sa_coop_error_state(1) & 1 and lib.sa_coop_error_state(0) == 0
    assert np.array_equal(S.farthest_point_sample(24, x).cpu().numpy(), good)


def test_multi_workgroup_sampler_under_capture_is_opt_in(gpu, oracle):
    # a capturing stream gets the single-workgroup kernels (safe on any number of streams) unless the caller
    # opts into the plain multi-workgroup launch (sa_fps_ex3 flags bit 0: it keeps such launches on one stream).  Same picks.
    lib = pkg("utils._native").lib()
    rng = np.random.default_rng(6)
    p = rng.normal(0, 1, (2, 20000, 3)).astype(np.float32)
    x = _t(p, gpu)
    ref = oracle.farthest_point_sample(40, p)
    temp = torch.empty((2, 20000), dtype=torch.float32, device=gpu)
    outs = {}
    side = torch.cuda.Stream(device=gpu)
    for flags in (0, 1):
        out = torch.zeros((2, 40), dtype=torch.int32, device=gpu)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            st = lib.sa_fps_ex3(2, 20000, 3, 40, x.data_ptr(), 0, temp.data_ptr(), out.data_ptr(), 40, 0, None, 0, flags,
                                torch.cuda.current_stream().cuda_stream)
        assert st == 0
        g.replay()
        torch.cuda.synchronize()
        outs[flags] = out.cpu().numpy()
    assert np.array_equal(outs[0], ref) and np.array_equal(outs[1], ref)


def test_captured_multi_workgroup_sampler_does_not_read_what_its_exchange_words_held_before(gpu, oracle):
    # round 6 (tools/verify_layers.py): captured into a hipGraph, the hipMemsetAsync that zeroed the partners' exchange words
    # was not reliably ordered in front of the sampler's kernel node -- the workgroups then read whatever the region held
    # (in the executor: scratch of a later stage of the previous replay) and, where 16 bits of that matched the pick number,
    # disagreed about a winner: different picks in 2-10 % of the configs[4] replays, no time-out.  The words are now zeroed
    # by a KERNEL node (sa::zero_async).  Here the region is poisoned inside the graph, right in front of the sampler, with
    # words that pass for the partners' words of EVERY early pick; every replay must equal the oracle.  (On its own this graph
    # replays correctly with the memset node too -- the misordering needs the executor's stage graphs on three streams:
    # tests/test_pipeline_gpu.py::test_configs4_frames_replay_after_replay_equal_eager is the test that fails on that build.)
    lib = pkg("utils._native").lib()
    rng = np.random.default_rng(66)
    b, n, m = 4, 40000, 48                                            # 16 workgroups per frame
    p = rng.normal(0, 1, (b, n, 3)).astype(np.float32)
    x = _t(p, gpu)
    ref = oracle.farthest_point_sample(m, p)
    temp = torch.empty((b, n), dtype=torch.float32, device=gpu)
    words = temp.view(torch.int64).view(-1)[:b * 32]                  # 2 x 16 exchange words per frame
    # {value bits | pick number << 16 | tie key}: a huge value with the pick number of the word's parity slot
    poison = torch.tensor([(0x7F000000 << 32) | ((1 + (i // 16) % 2) << 16) | (i % 1024) for i in range(b * 32)],
                          dtype=torch.int64, device=gpu)
    out = torch.zeros((b, m), dtype=torch.int32, device=gpu)
    side = torch.cuda.Stream(device=gpu)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        words.copy_(poison)                                           # a kernel node in front of the sampler's nodes
        st = lib.sa_fps_ex3(b, n, 3, m, x.data_ptr(), 0, temp.data_ptr(), out.data_ptr(), m, 0, None, 0, 1,
                            torch.cuda.current_stream().cuda_stream)
    assert st == 0
    # (the misordering showed beside OTHER streams' kernels: a second stream keeps the chip busy during the replays)
    busy = torch.cuda.Stream(device=gpu)
    a = torch.randn((4096, 4096), device=gpu)
    for _ in range(24):
        out.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(busy):
            for _ in range(6):
                a = torch.tanh(a @ a) * 0.01
        with torch.cuda.stream(side):
            g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref)
    assert lib.sa_coop_error_state(0) == 0


@pytest.mark.gpu
def test_stand_alone_band_calls_over_one_point_set_share_the_grid(gpu, oracle):
    """tf_grouping.query_ball_point keeps the grid of the last point set: the per-band calls of the reference's
    layers_util.py:134-147 build it once.  Results must not depend on it: ascending radii (a finer grid cannot serve a
    wider band: rebuilt on the device), descending radii (a coarser grid serves a narrower band), contents changed in place
    (torch's version counter), and a new tensor -- all equal to the oracle."""
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(2025)
    b, n, m = 3, 4096, 512
    xyz1 = _cloud(rng, b, n, scale=4.0, dup=100)               # extent 8 m over 126 cells: cell size = the radius above 0.07
    xyz2 = np.concatenate([xyz1[:, :400], _cloud(rng, b, m - 400, scale=5.0)], 1)
    t1, t2 = _t(xyz1, gpu), _t(xyz2, gpu)
    assert not G.SHARE_GRID                                     # off by default: sharing is the caller's statement
    G.query_ball_point(0.4, 32, t1, t2)
    assert G._grid_of_last_call is None
    with G.shared_grid():
        for radii in ((0.2, 0.4, 0.8, 1.7), (1.7, 0.8, 0.4, 0.2), (0.4, 0.4, 3.0, 0.1)):
            for r in radii:
                idx, cnt = G.query_ball_point(r, 32, t1, t2)
                ridx, rcnt = oracle.query_ball_point(r, 32, xyz1, xyz2)
                _check_ball(idx.cpu().numpy(), cnt.cpu().numpy(), ridx, rcnt)
            g = G._grid_of_last_call
            assert g is not None and g[0]() is t1
        idx, cnt = G.query_ball_point_dilated(0.4, 0.8, 16, t1, t2)            # the dilated form through the same grid
        ridx, rcnt = oracle.query_ball_point_dilated(0.4, 0.8, 16, xyz1, xyz2)
        _check_ball(idx.cpu().numpy(), cnt.cpu().numpy(), ridx, rcnt)
        # contents replaced in place: same object, same pointer, new version
        xyz1b = _cloud(rng, b, n, scale=4.0, dup=50)
        t1.copy_(_t(xyz1b, gpu))
        idx, cnt = G.query_ball_point(0.4, 32, t1, t2)
        ridx, rcnt = oracle.query_ball_point(0.4, 32, xyz1b, xyz2)
        _check_ball(idx.cpu().numpy(), cnt.cpu().numpy(), ridx, rcnt)
        # a new tensor (possibly in the allocation the old one left)
        del t1
        xyz1c = _cloud(rng, b, n, scale=4.0, dup=10)
        t1 = _t(xyz1c, gpu)
        idx, cnt = G.query_ball_point(0.4, 32, t1, t2)
        ridx, rcnt = oracle.query_ball_point(0.4, 32, xyz1c, xyz2)
        _check_ball(idx.cpu().numpy(), cnt.cpu().numpy(), ridx, rcnt)
        # an inference tensor has no version counter: never shared, no exception
        with torch.inference_mode():
            ti = _t(xyz1c, gpu)
            idx3, cnt3 = G.query_ball_point(0.4, 32, ti, t2)
            idx3, cnt3 = G.query_ball_point(0.4, 32, ti, t2)
        assert torch.equal(idx, idx3) and torch.equal(cnt, cnt3)
    assert not G.SHARE_GRID and G._grid_of_last_call is None    # the block dropped its grid
    # and outside the block
    idx2, cnt2 = G.query_ball_point(0.4, 32, t1, t2)
    assert torch.equal(idx, idx2) and torch.equal(cnt, cnt2)
