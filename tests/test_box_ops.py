"""point-in-box operators, gather_by_mask and the gradients of the two gathers.

CPU part: hand-derived known answers for the oracle's restatement of point_inside_box_3d and friends
(tf_grouping_g.cu:27-209, tf_sampling_g.cu:339-384).  GPU part: the HIP kernels against the oracle, bit for bit,
through the reference's Python API names."""
import numpy as np
import pytest
import torch

from conftest import pkg


def _t(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


# ------------------------------------------------------------------------------------------------ oracle KATs
def test_oracle_box_membership_known_answers(oracle):
    # box: centre (10, bottom y 2, 20), l = 4 (along x at ry = 0), h = 1.5 (y from 0.5 to 2), w = 2 (along z)
    box = np.array([[[10, 2, 20, 4, 1.5, 2, 0.0]]], np.float32)
    pts = np.array([[[10, 1, 20],        # centre: in
                     [12, 1, 21],        # corner, on the boundary: in (closed intervals)
                     [12.001, 1, 20],    # just past +l/2: out
                     [10, 2, 20],        # y == by: in (y > by is the test)
                     [10, 2.001, 20],    # below the bottom: out
                     [10, 0.5, 20],      # by - y == h: in ((by - y) > h is the test)
                     [10, 0.499, 20],    # above the top: out
                     [10, 1, 21.001],    # past +w/2: out
                     [8, 1, 19]]], np.float32)   # opposite corner: in
    m = oracle.query_boxes_3d_mask(pts, box)
    assert m.tolist() == [[[1, 1, 0, 1, 0, 1, 0, 0, 1]]]
    # the same box turned by 90 degrees: the 4-long side now lies along z.  cos(pi/2) rounds to -4.37e-8, so
    # stay clear of the faces by more than that
    box90 = box.copy()
    box90[0, 0, 6] = np.float32(np.pi / 2)
    pts90 = np.array([[[10, 1, 21.9], [11.5, 1, 20], [10.9, 1, 20], [10, 1, 22.1]]], np.float32)
    assert oracle.query_boxes_3d_mask(pts90, box90).tolist() == [[[1, 0, 1, 0]]]
    idx, cnt = oracle.query_boxes_3d_points(4, pts, box)
    assert cnt.tolist() == [[4]] and idx.tolist() == [[[0, 1, 3, 5]]]
    idx, cnt = oracle.query_boxes_3d_points(8, pts, box)
    assert cnt.tolist() == [[5]] and idx.tolist() == [[[0, 1, 3, 5, 8, 0, 0, 0]]]      # padded with the first hit
    far = box.copy()
    far[0, 0, 0] = 500
    idx, cnt = oracle.query_boxes_3d_points(3, pts, far)
    assert cnt.tolist() == [[0]] and idx.tolist() == [[[0, 0, 0]]]
    # zero-size box: max_distance = 1e-20, only a point exactly at the centre line passes the pre-test
    zero = np.array([[[10, 2, 20, 0, 1.5, 0, 0.3]]], np.float32)
    assert oracle.query_boxes_3d_mask(pts, zero).tolist() == [[[1, 0, 0, 1, 0, 1, 0, 0, 0]]]


def test_oracle_points_iou_and_gather_by_mask_known_answers(oracle):
    pts = np.zeros((1, 10, 3), np.float32)
    pts[0, :, 0] = np.arange(10)                     # x = 0..9 on a line, y = 0, z = 0
    a = np.array([[[2.5, 1, 0, 6, 2, 2, 0]]], np.float32)     # x in [-0.5, 5.5]: points 0..5
    g = np.array([[[5.5, 1, 0, 6, 2, 2, 0], [50, 1, 0, 1, 1, 1, 0]]], np.float32)   # points 3..8 | none
    iou = np.array([[[0.5, 0.5]]], np.float32)
    out = oracle.query_points_iou(pts, a, g, iou)
    assert out[0, 0, 0] == np.float32(3.0 / 9.0)    # inside both: 3,4,5; inside either: 0..8
    assert out[0, 0, 1] == np.float32(0.0)          # gt without points: in = 0, un = 6
    assert oracle.query_points_iou(pts, a, g, np.array([[[0.0009, 0.5]]], np.float32))[0, 0, 0] == 0.0   # gated
    inp = np.arange(2 * 6 * 2, dtype=np.float32).reshape(2, 6, 2)
    mask = np.array([[0, 0.9, 1, 0, -1, 2], [0, 0, 0, 0, 0, 0]], np.float32)     # int(0.9) == 0, int(-1) != 0
    out = oracle.gather_by_mask(4, inp, mask)
    assert out[0].tolist() == [[4, 5], [8, 9], [10, 11], [4, 5]]                 # rows 2, 4, 5, then the first again
    assert (out[1] == 0).all()
    assert oracle.gather_by_mask(2, inp, mask)[0].tolist() == [[4, 5], [8, 9]]


def test_oracle_gather_gradients_known_answers(oracle):
    idx = np.array([[2, 0, 2]], np.int32)
    g = np.array([[[1, 10], [2, 20], [4, 40]]], np.float32)
    assert oracle.gather_point_grad(np.zeros((1, 4, 2)), idx, g).tolist() == [[[2, 20], [0, 0], [5, 50], [0, 0]]]
    gi = np.array([[[1, -1], [1, 3]]], np.int32)
    gg = np.arange(8, dtype=np.float32).reshape(1, 2, 2, 2)
    assert oracle.group_point_grad(np.zeros((1, 4, 2)), gi, gg).tolist() == [[[0, 0], [4, 6], [0, 0], [6, 7]]]


# --------------------------------------------------------------------------------------------- HIP vs oracle
def _boxes_on_points(rng, xyz, m, spread=0.0):
    b, n, _ = xyz.shape
    ctr = xyz[np.arange(b)[:, None], rng.integers(0, n, (b, m))]
    boxes = np.zeros((b, m, 7), np.float32)
    boxes[..., 0] = ctr[..., 0] + rng.normal(0, spread, (b, m))
    boxes[..., 3] = rng.uniform(1.0, 6.0, (b, m))
    boxes[..., 4] = rng.uniform(1.0, 2.5, (b, m))
    boxes[..., 5] = rng.uniform(0.8, 2.5, (b, m))
    boxes[..., 1] = ctr[..., 1] + boxes[..., 4] * rng.uniform(0.2, 0.8, (b, m))
    boxes[..., 2] = ctr[..., 2] + rng.normal(0, spread, (b, m))
    boxes[..., 6] = rng.uniform(-np.pi, np.pi, (b, m))
    return boxes.astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,nsample", [(2, 16384, 64, 512), (1, 1000, 37, 16), (3, 70, 5, 1), (2, 4096, 128, 64)])
def test_query_boxes_3d_points_and_mask_match_oracle(gpu, oracle, b, n, m, nsample):
    G, syn = pkg("utils.tf_ops.grouping.tf_grouping"), pkg("synthetic")
    rng = np.random.default_rng(n + m)
    xyz = syn.kitti_like_batch(b, n=n)[:, :, :3].copy()
    boxes = _boxes_on_points(rng, xyz, m)
    boxes[0, 0, 0] += 1000.0                                   # an empty box
    boxes[0, 1, 3:6] = 0.0                                     # a zero-size box
    boxes[-1, -1, 3:6] = (200.0, 20.0, 200.0)                  # a box holding (almost) the whole frame
    boxes[-1, -1, 1] = 10.0
    idx, cnt = G.query_boxes_3d_points(nsample, _t(xyz, gpu), _t(boxes, gpu))
    ridx, rcnt = oracle.query_boxes_3d_points(nsample, xyz, boxes)
    assert idx.dtype == torch.int32 and tuple(idx.shape) == (b, m, nsample) and tuple(cnt.shape) == (b, m)
    assert np.array_equal(cnt.cpu().numpy(), rcnt) and np.array_equal(idx.cpu().numpy(), ridx)
    assert rcnt[0, 0] == 0 and rcnt[-1, -1] == min(nsample, n) and (nsample == 1 or rcnt.max() > 1)
    mask = G.query_boxes_3d_mask(_t(xyz, gpu), _t(boxes, gpu))
    rmask = oracle.query_boxes_3d_mask(xyz, boxes)
    assert mask.dtype == torch.int32 and np.array_equal(mask.cpu().numpy(), rmask)
    # the two operators agree with each other: cnt == min(nsample, number of mask hits)
    assert np.array_equal(np.minimum(rmask.sum(-1), nsample), rcnt)


@pytest.mark.gpu
def test_boxes_on_lattice_points_boundaries(gpu, oracle):
    """axis-aligned boxes with faces exactly on half-integer lattice coordinates: every comparison is an equality case"""
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    g = np.arange(-4, 5, dtype=np.float32) * 0.5
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, -1, 3)
    boxes = np.array([[[0, 1, 0, 2, 2, 3, 0], [0.5, 0.5, -0.5, 1, 1, 1, np.pi], [0, 0, 0, 4, 0, 4, np.pi / 2],
                       [1, 2, 1, 1, 3, 2, -np.pi / 2]]], np.float32)
    m1 = G.query_boxes_3d_mask(_t(xyz, gpu), _t(boxes, gpu)).cpu().numpy()
    assert np.array_equal(m1, oracle.query_boxes_3d_mask(xyz, boxes)) and m1.sum() > 50
    i1, c1 = G.query_boxes_3d_points(40, _t(xyz, gpu), _t(boxes, gpu))
    i2, c2 = oracle.query_boxes_3d_points(40, xyz, boxes)
    assert np.array_equal(i1.cpu().numpy(), i2) and np.array_equal(c1.cpu().numpy(), c2)


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,a,g", [(2, 16384, 96, 7), (1, 500, 10, 1)])
def test_query_points_iou_matches_oracle(gpu, oracle, b, n, a, g):
    G, syn = pkg("utils.tf_ops.grouping.tf_grouping"), pkg("synthetic")
    rng = np.random.default_rng(a + g)
    xyz = syn.kitti_like_batch(b, n=n)[:, :, :3].copy()
    gt = _boxes_on_points(rng, xyz, g)
    anchors = gt[:, rng.integers(0, g, a)] + rng.normal(0, 0.3, (b, a, 7)).astype(np.float32)
    iou = rng.uniform(0, 0.01, (b, a, g)).astype(np.float32)          # ~10 % of the pairs fall below the 1e-3 gate
    got = G.query_points_iou(_t(xyz, gpu), _t(anchors, gpu), _t(gt, gpu), _t(iou, gpu)).cpu().numpy()
    ref = oracle.query_points_iou(xyz, anchors, gt, iou)
    assert np.array_equal(got, ref)
    assert (ref[iou < 1e-3] == 0).all() and ref.max() > 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,c,pn,density", [(3, 16384, 128, 256, 0.05), (2, 1000, 7, 64, 0.01), (2, 300, 4, 512, 0.5),
                                              (1, 64, 3, 8, 1.0)])
def test_gather_by_mask_matches_oracle(gpu, oracle, b, n, c, pn, density):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    rng = np.random.default_rng(n + pn)
    inp = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    mask = (rng.uniform(0, 1, (b, n)) < density).astype(np.float32) * rng.choice([1.0, 2.0, -1.0, 0.5], (b, n)).astype(np.float32)
    mask[-1] = 0 if b > 1 else mask[-1]                                 # a frame without any selected point
    got = S.gather_by_mask(pn, _t(inp, gpu), _t(mask, gpu)).cpu().numpy()
    assert got.shape == (b, pn, c) and np.array_equal(got, oracle.gather_by_mask(pn, inp, mask))


@pytest.mark.gpu
def test_gather_and_group_gradients_match_oracle(gpu, oracle):
    S, G = pkg("utils.tf_ops.sampling.tf_sampling"), pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(9)
    b, n, m, ns, c = 2, 500, 300, 16, 9
    inp = np.zeros((b, n, c), np.float32)
    idx = rng.integers(0, 40, (b, m)).astype(np.int32)                   # heavy collisions
    # small integers: every partial sum is exact in fp32, so the atomics' order cannot matter -> bit-exact check
    g_int = rng.integers(-8, 9, (b, m, c)).astype(np.float32)
    got = S.gather_point_grad(_t(inp, gpu), _t(idx, gpu), _t(g_int, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.gather_point_grad(inp, idx, g_int))
    g_f = rng.normal(0, 1, (b, m, c)).astype(np.float32)
    got = S.gather_point_grad(_t(inp, gpu), _t(idx, gpu), _t(g_f, gpu)).cpu().numpy()
    np.testing.assert_allclose(got, oracle.gather_point_grad(inp, idx, g_f), rtol=0, atol=2e-5)
    gidx = rng.integers(-1, 60, (b, m, ns)).astype(np.int32)             # -1 rows are skipped
    gg = rng.integers(-8, 9, (b, m, ns, c)).astype(np.float32)
    got = G.group_point_grad(_t(inp, gpu), _t(gidx, gpu), _t(gg, gpu)).cpu().numpy()
    assert np.array_equal(got, oracle.group_point_grad(inp, gidx, gg))
    # adjoint identity with the forward gathers: <group_point(x, idx), g> == <x, group_point_grad(x, idx, g)>
    x = rng.integers(-4, 5, (b, n, c)).astype(np.float32)
    fwd = G.group_point(_t(x, gpu), _t(gidx, gpu)).double()
    lhs = float((fwd * _t(gg, gpu).double()).sum())
    rhs = float((_t(x, gpu).double() * _t(got, gpu).double()).sum())
    assert lhs == rhs


@pytest.mark.gpu
def test_box_ops_reject_bad_arguments(gpu):
    G, S = pkg("utils.tf_ops.grouping.tf_grouping"), pkg("utils.tf_ops.sampling.tf_sampling")
    xyz = torch.zeros(2, 10, 3, device=gpu)
    with pytest.raises(ValueError, match="positive nsample"):
        G.query_boxes_3d_points(0, xyz, torch.zeros(2, 3, 7, device=gpu))
    with pytest.raises(ValueError, match="proposal shape"):
        G.query_boxes_3d_points(4, xyz, torch.zeros(2, 3, 6, device=gpu))
    with pytest.raises(ValueError, match="xyz shape"):
        G.query_boxes_3d_mask(torch.zeros(2, 10, 4, device=gpu), torch.zeros(2, 3, 7, device=gpu))
    with pytest.raises(ValueError, match="iou_matrix"):
        G.query_points_iou(xyz, torch.zeros(2, 3, 7, device=gpu), torch.zeros(2, 2, 7, device=gpu), torch.zeros(2, 3, 3, device=gpu))
    with pytest.raises(ValueError, match="positive proposal"):
        S.gather_by_mask(0, torch.zeros(2, 10, 4, device=gpu), torch.zeros(2, 10, device=gpu))
    with pytest.raises(ValueError, match="mask shape"):
        S.gather_by_mask(4, torch.zeros(2, 10, 4, device=gpu), torch.zeros(2, 9, device=gpu))
