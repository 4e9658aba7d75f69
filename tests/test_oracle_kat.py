"""Known-answer tests that pin the CPU oracle to the behavioural spec of the reference kernels
(SURVEY.md Appendix A).  Every expected value below is derived by hand from the reference source
lines cited in each test -- none of them comes from running the oracle.  (The reference itself has no
tests for these ops: "parity unpinned")"""
import numpy as np

f32 = np.float32


def test_fps_line(oracle):
    # tf_sampling_g.cu:123-178: start at 0; farthest from x=0 is x=10 (idx 4); then running min
    # distances are 1,4,9 for idx 1,2,3 -> idx 3.
    pts = np.zeros((1, 5, 3), f32)
    pts[0, :, 0] = [0, 1, 2, 3, 10]
    assert oracle.farthest_point_sample(3, pts).tolist() == [[0, 4, 3]]
    # all five: after x=3 joins, idx 1 keeps min(1, 4)=1 and idx 2 drops to min(4, 1)=1: a tie, the lower
    # thread (idx 1) wins, then idx 2
    assert oracle.farthest_point_sample(5, pts).tolist() == [[0, 4, 3, 1, 2]]


def test_fps_first_index_always_zero_and_batch_independent(oracle):
    pts = np.zeros((2, 4, 3), f32)
    pts[0, :, 1] = [5, 0, 1, 2]      # frame 0: start idx0 (y=5) -> farthest y=0 (idx1) -> y=2? min(9,4)=4 idx3 vs idx2 min(16,1)=1
    pts[1, :, 2] = [0, 0, 0, 7]      # frame 1: -> idx3, then all others tie at 0 -> lowest thread id = idx 0
    out = oracle.farthest_point_sample(3, pts)
    assert out[0].tolist() == [0, 1, 3]
    assert out[1].tolist() == [0, 3, 0]   # all remaining min-distances are 0: thread 0's (0 > -1) wins, :141,154


def test_fps_tiebreak_k_mod_1024(oracle):
    # tf_sampling_g.cu:142,154-171: thread t = k mod 1024 keeps its first strict max; the tree keeps the
    # left (smaller t) entry on ties.  Two equally far points at k=7 (t=7) and k=1030 (t=6): t=6 wins
    # although 7 < 1030.
    n = 2048
    pts = np.zeros((1, n, 3), f32)
    pts[0, 7, 0] = 3.0
    pts[0, 1030, 0] = -3.0
    assert oracle.farthest_point_sample(2, pts).tolist() == [[0, 1030]]
    # same thread (t=6): k=6 and k=1030 -> first strict max in ascending k -> 6
    pts = np.zeros((1, n, 3), f32)
    pts[0, 6, 1] = 3.0
    pts[0, 1030, 1] = -3.0
    assert oracle.farthest_point_sample(2, pts).tolist() == [[0, 6]]


def test_fps_fma_chain(oracle):
    # Decision A: d = fmaf(diff,diff,d) per channel (nvcc -fmad=true form of tf_sampling_g.cu:146-150).
    # For these three points the fused chain gives d(p1)=2.0037386 < d(p2)=2.0037389 (pick 2); separate
    # multiply+add rounds the other way (would pick 1).  Values found by search, answer from exact
    # rational arithmetic of the two formulas.
    o = [-0.6892402172088623, -0.11729754507541656, -0.16030718386173248]
    a = [0.6565782427787781, 0.3208279013633728, -0.18391664326190948]
    b = [-0.25111478567123413, -0.14090700447559357, 1.1855113506317139]
    pts = np.array([[o, a, b]], f32)
    assert oracle.farthest_point_sample(2, pts).tolist() == [[0, 2]]
    o = [0.707705020904541, -0.8554568886756897, 0.46334177255630493]
    a = [-0.8842937350273132, 0.019086552783846855, 0.24701596796512604]
    b = [1.5822484493255615, -1.0717827081680298, -1.1286571025848389]
    pts = np.array([[o, a, b]], f32)
    assert oracle.farthest_point_sample(2, pts).tolist() == [[0, 1]]


def test_fps_generic_channels(oracle):
    # c = 5 feature-space FPS (the generic kernel takes c channels, tf_sampling_g.cu:146-150)
    pts = np.zeros((1, 4, 5), f32)
    pts[0, 1, 4] = 2.0      # d = 4
    pts[0, 2, 0] = 1.0
    pts[0, 2, 3] = 2.0      # d = 5  -> first pick
    pts[0, 3, 2] = 1.0      # d = 1
    # after picking 2: min-dists: idx1: min(4, 1+4+4=9)=4 ; idx3: min(1, 1+4+1=6)=1 -> idx1
    assert oracle.farthest_point_sample(3, pts).tolist() == [[0, 2, 1]]


def test_fps_with_distance(oracle):
    # tf_sampling_g.cu:180-230: same loop with d = dist[old, k]
    D = np.array([[[0, 1, 5, 2],
                   [1, 0, 3, 9],
                   [5, 3, 0, 4],
                   [2, 9, 4, 0]]], f32)
    # old=0: temp=[0,1,5,2] -> 2 ; old=2: temp=min(.,[5,3,0,4])=[0,1,0,2] -> 3 ; old=3: [0,1,0,0] -> 1
    assert oracle.farthest_point_sample_with_distance(4, D).tolist() == [[0, 2, 3, 1]]


def test_fps_with_distance_negative_values(oracle):
    # best starts at -1 and besti at 0 (tf_sampling_g.cu:190-191): if every running minimum is <= -1 no
    # thread ever updates and index 0 is returned.
    D = np.full((1, 3, 3), -2.0, f32)
    assert oracle.farthest_point_sample_with_distance(3, D).tolist() == [[0, 0, 0]]
    # slightly negative values (expansion-form round-off) still order normally: -0.25 > -0.5 > -1
    D = np.array([[[0, -0.5, -0.25], [-0.5, 0, 7], [-0.25, 7, 0]]], f32)
    # old=0: temp=[0,-.5,-.25] -> idx0 has 0 (largest) -> picks 0 again
    assert oracle.farthest_point_sample_with_distance(2, D).tolist() == [[0, 0]]


def test_gather_point(oracle):
    inp = np.arange(2 * 4 * 2, dtype=f32).reshape(2, 4, 2)
    idx = np.array([[3, 0, 3], [1, 1, 2]], np.int32)
    out = oracle.gather_point(inp, idx)
    assert out.tolist() == [[[6, 7], [0, 1], [6, 7]], [[10, 11], [10, 11], [12, 13]]]


def _line(xs):
    p = np.zeros((1, len(xs), 3), f32)
    p[0, :, 0] = xs
    return p


def test_query_ball_point_first_nsample_and_padding(oracle):
    # tf_grouping_g.cu:215-255.  Points on a line at x=0,.25,.5,.75,1,.. ; query at x=0.5, radius 0.5.
    # |dx| < 0.5 strictly (:244): x=.25,.5,.75 -> idx 1,2,3 ; x=0 and x=1 are at exactly 0.5: excluded.
    xyz1 = _line([0, .25, .5, .75, 1, 1.25])
    xyz2 = _line([0.5])
    idx, cnt = oracle.query_ball_point(0.5, 5, xyz1, xyz2)
    assert cnt.tolist() == [[3]]
    assert idx.tolist() == [[[1, 2, 3, 1, 1]]]        # unused slots keep the first hit (:245-248)
    idx, cnt = oracle.query_ball_point(0.5, 2, xyz1, xyz2)
    assert cnt.tolist() == [[2]] and idx.tolist() == [[[1, 2]]]   # only the FIRST nsample (:237-239)


def test_query_ball_point_empty_ball_zero_filled(oracle):
    xyz1 = _line([0, 1, 2])
    xyz2 = _line([10])
    idx, cnt = oracle.query_ball_point(0.5, 4, xyz1, xyz2)
    assert cnt.tolist() == [[0]] and idx.tolist() == [[[0, 0, 0, 0]]]   # oracle decision D


def test_query_ball_point_radius_is_float32(oracle):
    # radius is a float32 attr (tf_grouping.cpp:59): 0.2 -> 0.20000000298...; a point at distance
    # float32(0.2) is NOT < radius; the next float32 below is.
    r32 = f32(0.2)
    below = np.nextafter(r32, f32(0))
    xyz1 = _line([r32, below])
    xyz2 = _line([0])
    idx, cnt = oracle.query_ball_point(0.2, 2, xyz1, xyz2)
    assert cnt.tolist() == [[1]] and idx.tolist() == [[[1, 1]]]


def test_query_ball_point_dilated(oracle):
    # tf_grouping_g.cu:308-357: hit iff d == 0 or min_r <= d < max_r.
    xyz1 = _line([0, .25, .5, .75, 1, 1.25, .5])
    xyz2 = _line([0.5])
    # band [0.25, 0.5): |dx|=.25 -> idx 1,3 ; plus d==0 -> idx 2 and the duplicate idx 6 ; .5 excluded
    idx, cnt = oracle.query_ball_point_dilated(0.25, 0.5, 4, xyz1, xyz2)
    assert cnt.tolist() == [[4]] and idx.tolist() == [[[1, 2, 3, 6]]]
    # band [0.5, 0.8): |dx| = .5 (idx 0,4), .75 (idx 5) ; the centre joins EVERY band via d == 0 (:337)
    idx, cnt = oracle.query_ball_point_dilated(0.5, 0.8, 8, xyz1, xyz2)
    assert cnt.tolist() == [[5]] and idx[0, 0].tolist() == [0, 2, 4, 5, 6, 0, 0, 0]
    # min_radius = 0 behaves like a plain ball without the 1e-20 clamp
    idx, cnt = oracle.query_ball_point_dilated(0.0, 0.3, 3, xyz1, xyz2)
    assert cnt.tolist() == [[3]] and idx.tolist() == [[[1, 2, 3]]]


def test_group_point(oracle):
    pts = np.arange(1 * 3 * 2, dtype=f32).reshape(1, 3, 2) + 1
    idx = np.array([[[2, -1], [0, 1]]], np.int32)
    out = oracle.group_point(pts, idx)
    assert out.tolist() == [[[[5, 6], [0, 0]], [[1, 2], [3, 4]]]]   # -1 -> 0.0 (tf_grouping_g.cu:373-375)


def test_calc_square_dist_small_integers(oracle):
    # model_util.py:144-160: |a|^2 + |b|^2 - 2ab, exact on small integers
    a = np.array([[[1, 2], [0, -1]]], f32)
    b = np.array([[[1, 2], [3, 0], [0, 0]]], f32)
    d = oracle.calc_square_dist(a, b)
    assert d.tolist() == [[[0, 8, 5], [10, 10, 1]]]


def test_fold_conv_bn_and_dense(oracle):
    # y = relu(gamma*(xW + b - mean)/sqrt(var+1e-3) + beta): one channel, hand numbers
    params = {"s/weights": np.array([[[[2.0]]]], f32), "s/biases": np.array([1.0], f32),
              "s/bn/gamma": np.array([3.0], f32), "s/bn/beta": np.array([-1.0], f32),
              "s/bn/moving_mean": np.array([0.5], f32), "s/bn/moving_variance": np.array([0.999], f32)}
    w, b = oracle.fold_conv_bn(params, "s")
    # var + eps = 1.0 -> scale 3 ; W' = 6 ; b' = (1 - .5)*3 - 1 = .5
    assert abs(w[0, 0] - 6.0) < 1e-6 and abs(b[0] - 0.5) < 1e-6
    y = oracle.dense(np.array([[1.0], [-1.0]], f32), w, b, relu=True)
    assert np.allclose(y[:, 0], [6.5, 0.0], atol=1e-6)


def test_group_mlp_max_hand(oracle):
    # layers_util.py:157-181: features first then relative xyz; max over samples; empty balls -> 0
    xyz = np.array([[[0, 0, 0], [1, 0, 0], [0, 2, 0]]], f32)
    feat = np.array([[[10.0], [20.0], [30.0]]], f32)
    new_xyz = np.array([[[1, 0, 0], [5, 5, 5]]], f32)
    idx = np.array([[[0, 1], [2, 2]]], np.int32)
    cnt = np.array([[2, 0]], np.int32)
    # one linear layer that copies [feat, dx, dy, dz] -> 4 outputs (identity), then ReLU
    Wm = np.eye(4, dtype=f32)
    out = oracle.group_mlp_max(xyz, feat, new_xyz, idx, cnt, [Wm], [np.zeros(4, f32)])
    # ball 0 rows: [10, -1, 0, 0] and [20, 0, 0, 0] -> relu -> max = [20, 0, 0, 0]
    assert out[0, 0].tolist() == [20, 0, 0, 0]
    assert out[0, 1].tolist() == [0, 0, 0, 0]      # cnt == 0 -> masked (:180)


def test_sa_layer_fs_ordering_and_offsets(oracle):
    # layers_util.py:93-98,108: 'FS' emits the F-FPS indices first, then the D-FPS indices.
    # 4 points: xyz spread along x; features make point 1 the farthest in feature space.
    xyz = np.array([[[0, 0, 0], [1, 0, 0], [2, 0, 0], [9, 0, 0]]], f32)
    feat = np.array([[[0.0], [100.0], [0.0], [0.0]]], f32)
    params = {}
    nx, npts, idx = oracle.pointnet_sa_module_msg(
        xyz, feat, [], [], [], True, [-1], ["FS"], [2], None, "s", False, params)
    # F-FPS: start 0, farthest in (xyz,feat) space is idx 1 ; D-FPS: start 0, farthest is idx 3
    assert idx.tolist() == [[0, 1, 0, 3]]
    assert nx[0, :, 0].tolist() == [0, 1, 0, 9]
    assert npts[0, :, 0].tolist() == [0, 100, 0, 0]      # no radii: gather_point(points, fps_idx), :186-187
    # two ranges with an offset: range [0:2] identity (npoint == size), range [2:] D-FPS -> +2 (:108)
    nx, npts, idx = oracle.pointnet_sa_module_msg(
        xyz, feat, [], [], [], True, [2, -1], ["F-FPS", "D-FPS"], [2, 1], None, "s", False, params)
    assert idx.tolist() == [[0, 1, 2]]
    # npoint 0 range is skipped (:87-89)
    nx, npts, idx = oracle.pointnet_sa_module_msg(
        xyz, feat, [], [], [], True, [2, -1], ["F-FPS", "D-FPS"], [2, 0], None, "s", False, params)
    assert idx.tolist() == [[0, 1]]


def test_vote_layer_clamp(oracle):
    # layers_util.py:12-24: offsets clipped to +-|MAX_TRANSLATE_RANGE| = (3,2,3)
    params = {"v/vote_offsets/weights": (np.eye(3, dtype=f32) * 10).reshape(1, 3, 3),
              "v/vote_offsets/biases": np.zeros(3, f32)}
    xyz = np.zeros((1, 2, 3), f32)
    pts = np.array([[[1, 1, -1], [0.1, -0.1, 0.2]]], f32)
    nx, f, off = oracle.vote_layer(xyz, pts, [], True, "v", params, (-3.0, -2.0, -3.0))
    assert off[0].tolist() == [[10, 10, -10], [1, -1, 2]]
    assert nx[0].tolist() == [[3, 2, -3], [1, -1, 2]]
