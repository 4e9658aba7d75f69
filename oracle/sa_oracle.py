"""CPU oracle for the 3DSSD set-abstraction hot path (numpy front-end of sa_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package (3dssd_amd/) never imports it.

PARITY PINNED to the reference's own device code (tf_sampling_g.cu / tf_grouping_g.cu compiled unmodified for
gfx950 into oracle/_ref, `make -C oracle ref_gpu`): tests/test_ref_pin_gpu.py (live, three-way with the HIP kernels)
and tests/golden/ref_gpu_pin.npz + tests/test_ref_golden_cpu.py (CPU suite).  See the header of sa_oracle.c for what
that covers and for the arithmetic decisions; the MLP / distance-matrix arithmetic (TensorFlow, cuBLAS) stays a
stated definition.

Function names, positional argument order and return arity follow the reference's Python
operator API (scalars first, tensors last):
  lib/utils/tf_ops/sampling/tf_sampling.py:24,43,54
  lib/utils/tf_ops/grouping/tf_grouping.py:53,68,114
  lib/utils/model_util.py:144
  lib/utils/layers_util.py:12-24,59-189
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsa_oracle.so")
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile sa_oracle.c -> libsa_oracle.so (gcc, a few seconds)."""
    src = os.path.join(_HERE, "sa_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsa_oracle.so"])
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = ctypes.CDLL(_SO)
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


# --------------------------------------------------------------------------- sampling ops
def farthest_point_sample(npoint, inp):
    """tf_sampling.py:43-51 -> tf_sampling_g.cu:123-178.  inp [b,n,c] -> int32 [b,npoint]."""
    inp, pi = _f(inp)
    b, n, c = inp.shape
    out = np.zeros((b, npoint), np.int32)
    temp = np.empty((b, n), np.float32)
    lib().orc_farthest_point_sample(b, n, c, int(npoint), pi, temp.ctypes.data_as(_f32p),
                                    out.ctypes.data_as(_i32p))
    return out


def farthest_point_sample_with_distance(npoint, dist):
    """tf_sampling.py:54-62 -> tf_sampling_g.cu:180-230.  dist [b,n,n] -> int32 [b,npoint]."""
    dist, pd = _f(dist)
    b, n, n2 = dist.shape
    assert n == n2
    out = np.zeros((b, npoint), np.int32)
    temp = np.empty((b, n), np.float32)
    lib().orc_farthest_point_sample_with_distance(b, n, int(npoint), pd,
                                                  temp.ctypes.data_as(_f32p),
                                                  out.ctypes.data_as(_i32p))
    return out


def gather_point(inp, idx):
    """tf_sampling.py:24-32 -> tf_sampling_g.cu:320-331."""
    inp, pi = _f(inp)
    idx, px = _i(idx)
    b, n, c = inp.shape
    m = idx.shape[1]
    out = np.empty((b, m, c), np.float32)
    lib().orc_gather_point(b, n, m, c, pi, px, out.ctypes.data_as(_f32p))
    return out


def calc_square_dist(a, b, norm=False):
    """model_util.py:144-160 (norm=False only; decision E of sa_oracle.c)."""
    assert not norm
    a, pa = _f(a)
    b, pb = _f(b)
    bs, n, c = a.shape
    m = b.shape[1]
    out = np.empty((bs, n, m), np.float32)
    lib().orc_calc_square_dist(bs, n, m, c, pa, pb, out.ctypes.data_as(_f32p))
    return out


# --------------------------------------------------------------------------- grouping ops
def query_ball_point(radius, nsample, xyz1, xyz2):
    """tf_grouping.py:53-66 -> tf_grouping_g.cu:215-255.  -> (idx [b,m,ns], pts_cnt [b,m])."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    cnt = np.empty((b, m), np.int32)
    lib().orc_query_ball_point(b, n, m, ctypes.c_float(np.float32(radius)), int(nsample), p1, p2,
                               idx.ctypes.data_as(_i32p), cnt.ctypes.data_as(_i32p))
    return idx, cnt


def query_ball_point_dilated(min_radius, max_radius, nsample, xyz1, xyz2):
    """tf_grouping.py:68-83 -> tf_grouping_g.cu:308-357."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    cnt = np.empty((b, m), np.int32)
    lib().orc_query_ball_point_dilated(b, n, m, ctypes.c_float(np.float32(min_radius)),
                                       ctypes.c_float(np.float32(max_radius)), int(nsample), p1,
                                       p2, idx.ctypes.data_as(_i32p), cnt.ctypes.data_as(_i32p))
    return idx, cnt


def group_point(points, idx):
    """tf_grouping.py:114-122 -> tf_grouping_g.cu:362-379."""
    points, pp = _f(points)
    idx, px = _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    lib().orc_group_point(b, n, c, m, ns, pp, px, out.ctypes.data_as(_f32p))
    return out


# --------------------------------------------------------------------------- rank-4 operators
def query_boxes_3d_points(nsample, xyz, proposals):
    """tf_grouping.py:39-50 -> tf_grouping_g.cu:44-95."""
    xyz, px = _f(xyz)
    proposals, pp = _f(proposals)
    b, n, _ = xyz.shape
    m = proposals.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    cnt = np.empty((b, m), np.int32)
    lib().orc_query_boxes_3d_points(b, n, m, int(nsample), px, pp, idx.ctypes.data_as(_i32p), cnt.ctypes.data_as(_i32p))
    return idx, cnt


def query_boxes_3d_mask(xyz, boxes_3d):
    """tf_grouping.py:15-24 -> tf_grouping_g.cu:98-134."""
    xyz, px = _f(xyz)
    boxes_3d, pb = _f(boxes_3d)
    b, n, _ = xyz.shape
    m = boxes_3d.shape[1]
    mask = np.empty((b, m, n), np.int32)
    lib().orc_query_boxes_3d_mask(b, n, m, px, pb, mask.ctypes.data_as(_i32p))
    return mask


def query_points_iou(xyz, anchors_3d, gt_boxes_3d, iou_matrix):
    """tf_grouping.py:26-37 -> tf_grouping_g.cu:137-209."""
    xyz, px = _f(xyz)
    anchors_3d, pa = _f(anchors_3d)
    gt_boxes_3d, pg = _f(gt_boxes_3d)
    iou_matrix, pi = _f(iou_matrix)
    b, n, _ = xyz.shape
    a, g = anchors_3d.shape[1], gt_boxes_3d.shape[1]
    out = np.empty((b, a, g), np.float32)
    lib().orc_query_points_iou(b, n, a, g, px, pa, pg, pi, out.ctypes.data_as(_f32p))
    return out


def gather_by_mask(proposal_num, inp, mask):
    """tf_sampling.py:76-85 -> tf_sampling_g.cu:356-384."""
    inp, pi = _f(inp)
    mask, pm = _f(mask)
    b, n, c = inp.shape
    out = np.empty((b, proposal_num, c), np.float32)
    lib().orc_gather_by_mask(b, n, c, int(proposal_num), pi, pm, out.ctypes.data_as(_f32p))
    return out


def gather_point_grad(inp, idx, out_g):
    """tf_sampling.py:38-42 -> tf_sampling_g.cu:339-351."""
    idx, px = _i(idx)
    out_g, pg = _f(out_g)
    b, n, c = np.shape(inp)
    m = idx.shape[1]
    inp_g = np.empty((b, n, c), np.float32)
    lib().orc_gather_point_grad(b, n, m, c, pg, px, inp_g.ctypes.data_as(_f32p))
    return inp_g


def group_point_grad(points, idx, grad_out):
    """tf_grouping.py:125-128 -> tf_grouping_g.cu:384-400."""
    idx, px = _i(idx)
    grad_out, pg = _f(grad_out)
    b, n, c = np.shape(points)
    _, m, ns = idx.shape
    out = np.empty((b, n, c), np.float32)
    lib().orc_group_point_grad(b, n, c, m, ns, pg, px, out.ctypes.data_as(_f32p))
    return out


def query_ball_point_withidx(radius, nsample, xyz1, xyz2, sort_idx):
    """tf_grouping.py:85-100 -> tf_grouping_g.cu:259-304."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    sort_idx, ps = _i(sort_idx)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    cnt = np.empty((b, m), np.int32)
    lib().orc_query_ball_point_withidx(b, n, m, ctypes.c_float(np.float32(radius)), int(nsample), p1, p2, ps,
                                       idx.ctypes.data_as(_i32p), cnt.ctypes.data_as(_i32p))
    return idx, cnt


def select_top_k(k, dist):
    """tf_grouping.py:103-113 -> tf_grouping_g.cu:404-443.  -> (idx [b,m,n], dist_out [b,m,n])."""
    dist, pd = _f(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), np.int32)
    out = np.empty((b, m, n), np.float32)
    lib().orc_selection_sort(b, n, m, int(k), pd, outi.ctypes.data_as(_i32p), out.ctypes.data_as(_f32p))
    return outi, out


def knn_point(k, xyz1, xyz2):
    """tf_grouping.py:130-160."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, m, n), np.float32)
    lib().orc_pairwise_sqdist(b, n, m, c, p1, p2, dist.ctypes.data_as(_f32p))
    outi, out = select_top_k(k, dist)
    return out[:, :, :k].copy(), outi[:, :, :k].copy()


def farthest_point_sample_with_preidx(npoint, inp, preidx):
    """tf_sampling.py:65-74 -> tf_sampling_g.cu:232-318."""
    inp, pi = _f(inp)
    preidx, pp = _i(preidx)
    b, n, c = inp.shape
    out = np.empty((b, npoint), np.int32)
    temp = np.empty((b, n), np.float32)
    lib().orc_farthest_point_sample_with_preidx(b, n, c, int(npoint), preidx.shape[1], pi, pp,
                                                temp.ctypes.data_as(_f32p), out.ctypes.data_as(_i32p))
    return out


def k_interpolate_grad(points, idx, weight, grad_out):
    """tf_interpolate.py:32-36,53-58 -> tf_interpolate_g.cu:115-140,167-189 (three = k == 3)."""
    idx, px = _i(idx)
    weight, pw = _f(weight)
    grad_out, pg = _f(grad_out)
    b, m, c = np.shape(points)
    n, k = idx.shape[1], idx.shape[2]
    out = np.empty((b, m, c), np.float32)
    lib().orc_k_interpolate_grad(b, n, c, m, k, pg, px, pw, out.ctypes.data_as(_f32p))
    return out


def points_pooling(pc, box_3d, pc_loc, l=7, h=7, w=7, sample_num=35):
    """points_pooling.py:10-21 -> tf_points_pooling_g.cu:36-118."""
    pc, ppc = _f(pc)
    box_3d, pb = _f(box_3d)
    pc_loc, pl = _f(pc_loc)
    bs, pn, pts, c = pc.shape
    feats = np.empty((bs, pn, l, h, w, sample_num, c), np.float32)
    idx = np.empty((bs, pn, l, h, w, sample_num), np.int32)
    num = np.empty((bs, pn, l, h, w), np.int32)
    pillars = np.empty((bs, pn, l, h, w, 3), np.float32)
    lib().orc_points_pooling(bs, pn, pts, c, l, h, w, sample_num, ppc, pb, pl, feats.ctypes.data_as(_f32p),
                             idx.ctypes.data_as(_i32p), num.ctypes.data_as(_i32p), pillars.ctypes.data_as(_f32p))
    return feats, idx, num, pillars


def points_pooling_grad(pc, out_idx, sampled_num_lists, features_grad):
    """points_pooling.py:22-29 -> tf_points_pooling_g.cu:131-153."""
    out_idx, pi = _i(out_idx)
    sampled_num_lists, ps = _i(sampled_num_lists)
    features_grad, pg = _f(features_grad)
    bs, pn, pts, c = np.shape(pc)
    _, _, l, h, w, sample_num = out_idx.shape
    out = np.empty((bs, pn, pts, c), np.float32)
    lib().orc_points_pooling_grad(bs, pn, pts, c, l, h, w, sample_num, pi, ps, pg, out.ctypes.data_as(_f32p))
    return out


def calc_iou(detections, groundtruths):
    """tf_evaluate.py:26-33 -> evaluate.cpp:1161-1194.  -> (iou_bev, iou_3d) [bs, det_num, gt_num]."""
    detections, pd = _f(detections)
    groundtruths, pg = _f(groundtruths)
    bs, dn, _ = detections.shape
    gn = groundtruths.shape[1]
    bev = np.empty((bs, dn, gn), np.float32)
    i3d = np.empty((bs, dn, gn), np.float32)
    lib().orc_calc_iou(bs, dn, gn, pd, pg, bev.ctypes.data_as(_f32p), i3d.ctypes.data_as(_f32p))
    return bev, i3d


def calc_iou_match(detections, groundtruths):
    """tf_evaluate.py:35-42 -> evaluate.cpp:1196-1227."""
    detections, pd = _f(detections)
    groundtruths, pg = _f(groundtruths)
    n = detections.shape[0]
    bev = np.empty((n,), np.float32)
    i3d = np.empty((n,), np.float32)
    lib().orc_calc_iou_match(n, pd, pg, bev.ctypes.data_as(_f32p), i3d.ctypes.data_as(_f32p))
    return bev, i3d


# --------------------------------------------------------------------------- MLP pieces
BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default, tf_util.py:424-444


def fold_conv_bn(params, scope, bn=True):
    """Fold conv + bias + inference BN into (W'[cin,cout], b'[cout]) in float64, return fp32.

    Variable names follow the reference's scopes (layers_util.py:175, tf_util.py:96,111,439-442).
    """
    w = np.asarray(params[scope + "/weights"], np.float64)
    w = w.reshape(-1, w.shape[-1])  # [1,1,cin,cout] / [1,cin,cout] -> [cin,cout]
    bias = np.asarray(params[scope + "/biases"], np.float64)
    if bn:
        g = np.asarray(params[scope + "/bn/gamma"], np.float64)
        beta = np.asarray(params[scope + "/bn/beta"], np.float64)
        mu = np.asarray(params[scope + "/bn/moving_mean"], np.float64)
        var = np.asarray(params[scope + "/bn/moving_variance"], np.float64)
        s = g / np.sqrt(var + BN_EPS)
        w = w * s[None, :]
        bias = (bias - mu) * s + beta
    return w.astype(np.float32), bias.astype(np.float32)


def dense(x, w, bias, relu=True):
    """y = relu(x @ w + bias) with the oracle's fmaf-chain order.  x [..., cin]."""
    x, px = _f(x)
    w, pw = _f(w)
    bias, pb = _f(bias)
    cin, cout = w.shape
    rows = x.size // cin
    y = np.empty(x.shape[:-1] + (cout,), np.float32)
    lib().orc_dense(rows, cin, cout, px, pw, pb, int(bool(relu)), y.ctypes.data_as(_f32p))
    return y


def group_mlp_max(xyz, points, new_xyz, idx, cnt, weights, biases):
    """One scale of layers_util.py:157-181 (mask, group, concat [feat, rel-xyz], MLP, max, mask)."""
    xyz, pxyz = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    idx, pidx = _i(idx)
    cnt, pcnt = _i(cnt)
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    if points is None:
        c, ppts = 0, None
    else:
        points, ppts = _f(points)
        c = points.shape[2]
    nl = len(weights)
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    bs_ = [np.ascontiguousarray(v, np.float32) for v in biases]
    dims = [c + 3] + [w.shape[1] for w in ws]
    assert ws[0].shape[0] == c + 3
    dims_a = (ctypes.c_int * (nl + 1))(*dims)
    W = (_f32p * nl)(*[w.ctypes.data_as(_f32p) for w in ws])
    B = (_f32p * nl)(*[v.ctypes.data_as(_f32p) for v in bs_])
    out = np.empty((b, m, dims[-1]), np.float32)
    lib().orc_group_mlp_max(b, n, m, ns, c, pxyz, ppts, pn, pidx, pcnt, nl, dims_a, W, B,
                            out.ctypes.data_as(_f32p))
    return out


# --------------------------------------------------------------------------- SA layer
def pointnet_sa_module_msg(xyz, points, radius_list, nsample_list, mlp_list, bn,
                           fps_sample_range_list, fps_method_list, npoint_list, former_fps_idx,
                           scope, dilated_group, params, vote_ctr=None, aggregation_channel=None,
                           aggregation_sa_feature=True, trace=None):
    """layers_util.py:59-189, inference mode.  Returns (new_xyz, new_points, fps_idx)."""
    bs = xyz.shape[0]
    cur = []
    last = 0
    for rng_, method, npoint in zip(fps_sample_range_list, fps_method_list, npoint_list):
        end = xyz.shape[1] if rng_ == -1 else last + rng_   # tf.slice size -1 = to the end
        tmp_xyz = xyz[:, last:end]
        tmp_points = points[:, last:end]
        if npoint == 0:                                      # :87-89
            last += rng_
            continue
        if vote_ctr is not None:                             # :90-92
            npoint = vote_ctr.shape[1]
            fps_idx = np.tile(np.arange(npoint, dtype=np.int32)[None], (bs, 1))
        elif method == "FS":                                 # :93-98, F-FPS indices first
            f = np.concatenate([tmp_xyz, tmp_points], -1)    # xyz first, :94
            d = calc_square_dist(f, f)
            i1 = farthest_point_sample_with_distance(npoint, d)
            i2 = farthest_point_sample(npoint, tmp_xyz)
            fps_idx = np.concatenate([i1, i2], -1)
        elif npoint == tmp_xyz.shape[1]:                     # :99-100
            fps_idx = np.tile(np.arange(npoint, dtype=np.int32)[None], (bs, 1))
        elif method == "F-FPS":                              # :101-104
            f = np.concatenate([tmp_xyz, tmp_points], -1)
            d = calc_square_dist(f, f)
            fps_idx = farthest_point_sample_with_distance(npoint, d)
        else:                                                # D-FPS :105-106
            fps_idx = farthest_point_sample(npoint, tmp_xyz)
        cur.append(fps_idx + last)                           # :108
        last += rng_
    fps_idx = np.concatenate(cur, -1).astype(np.int32)
    if former_fps_idx is not None:
        fps_idx = np.concatenate([fps_idx, former_fps_idx], -1)
    new_xyz = gather_point(vote_ctr if vote_ctr is not None else xyz, fps_idx)  # :116-119

    outs = []
    for i, (radius, nsample) in enumerate(zip(radius_list, nsample_list)):
        if dilated_group:                                    # :137-141
            min_r = 0.0 if i == 0 else radius_list[i - 1]
            idx, cnt = query_ball_point_dilated(min_r, radius, nsample, xyz, new_xyz)
        else:
            idx, cnt = query_ball_point(radius, nsample, xyz, new_xyz)
        ws, bs_ = [], []
        for j in range(len(mlp_list[i])):
            w, bb = fold_conv_bn(params, "%s/conv%d_%d" % (scope, i, j), bn)
            ws.append(w)
            bs_.append(bb)
        o = group_mlp_max(xyz, points, new_xyz, idx, cnt, ws, bs_)
        if trace is not None:
            trace.append(dict(scope=scope, scale=i, idx=idx, cnt=cnt, pooled=o))
        outs.append(o)
    if outs:
        new_points = np.concatenate(outs, -1)
        if aggregation_sa_feature:                           # :183-185
            w, bb = fold_conv_bn(params, scope + "/ensemble", bn)
            assert w.shape[1] == aggregation_channel
            new_points = dense(new_points, w, bb, relu=True)
    else:
        new_points = gather_point(points, fps_idx)           # :186-187
    return new_xyz, new_points, fps_idx


def vote_layer(xyz, points, mlp_list, bn, scope, params, max_translate_range):
    """layers_util.py:12-24.  Returns (xyz + clipped offsets, features, raw offsets)."""
    for i, _ch in enumerate(mlp_list):
        w, b = fold_conv_bn(params, "%s/vote_layer_%d" % (scope, i), bn)
        points = dense(points, w, b, relu=True)
    w, b = fold_conv_bn(params, scope + "/vote_offsets", bn=False)
    off = dense(points, w, b, relu=False)
    lo = np.asarray(max_translate_range, np.float32).reshape(1, 1, 3)   # negative numbers
    lim = np.minimum(np.maximum(off, lo), -lo)                           # :21-22
    return (xyz + lim).astype(np.float32), points, off


def sa_backbone(points_in, arch, params, max_translate_range=(-3.0, -2.0, -3.0),
                aggregation_sa_feature=True, trace=None):
    """single_stage_detector.py:115-125 + layer_builder.py:45-102 for SA_Layer / Vote_Layer rows.

    points_in [b,n,4] -> lists (xyz_list, feature_list, fps_idx_list); the backbone output is the
    last entry of xyz_list / feature_list.
    """
    points_in = np.ascontiguousarray(points_in, np.float32)
    xyz_list = [points_in[:, :, 0:3].copy()]
    feature_list = [points_in[:, :, 3:].copy()]
    fps_idx_list = [None]
    for row in arch:
        (xyz_index, feature_index, radius_list, nsample_list, mlp_list, bn, fps_range, fps_method,
         npoint_list, former_fps_idx, use_attention, layer_type, scope, dilated, vote_ctr_index,
         agg_channel) = row
        xyz_in = xyz_list[xyz_index[0]]
        feat_in = feature_list[feature_index[0]]
        if layer_type == "SA_Layer":
            assert not use_attention
            former = fps_idx_list[former_fps_idx] if former_fps_idx != -1 else None
            vote_ctr = xyz_list[vote_ctr_index] if vote_ctr_index != -1 else None
            xyz, feat, fidx = pointnet_sa_module_msg(
                xyz_in, feat_in, radius_list, nsample_list, mlp_list, bn, fps_range, fps_method,
                npoint_list, former, scope, dilated, params, vote_ctr=vote_ctr,
                aggregation_channel=agg_channel, aggregation_sa_feature=aggregation_sa_feature,
                trace=trace)
        elif layer_type == "Vote_Layer":
            xyz, feat, _off = vote_layer(xyz_in, feat_in, mlp_list, bn, scope, params,
                                         max_translate_range)
            fidx = None
        else:
            raise NotImplementedError(layer_type)
        xyz_list.append(xyz)
        feature_list.append(feat)
        fps_idx_list.append(fidx)
    return xyz_list, feature_list, fps_idx_list
