"""Grouping operators with the reference's Python API (lib/utils/tf_ops/grouping/tf_grouping.py),
on torch-ROCm tensors, backed by csrc/ballquery.hip and csrc/gather.hip through include/sa_ops.h.

Same names, positional order (scalars first), return arity/dtypes/shapes.  Forward only
(query ops are NoGradient in the reference, tf_grouping.py:66,83; GroupPoint's gradient,
tf_grouping.py:123-128, is out of scope).  Errors: ValueError with the reference's OP_REQUIRES
messages (lib/utils/tf_ops/grouping/tf_grouping.cpp:275-288,368-384,453-459).
Rows of empty balls are zero-filled (the reference leaves them unwritten).
"""
import torch

from .. import _tensor as T
from ... import _native as N


def _check_xyz(op, xyz1, xyz2):
    T.require(xyz1.dim() == 3 and xyz1.shape[2] == 3, "%s expects (batch_size, ndataset, 3) xyz1 shape." % op)
    T.require(xyz2.dim() == 3 and xyz2.shape[2] == 3, "%s expects (batch_size, npoint, 3) xyz2 shape." % op)
    T.require(xyz1.shape[0] == xyz2.shape[0], "%s expects xyz1 and xyz2 with the same batch_size" % op)


def query_ball_point(radius, nsample, xyz1, xyz2):
    """xyz1: (batch, ndataset, 3), xyz2: (batch, npoint, 3) ->
    idx (batch, npoint, nsample) int32, pts_cnt (batch, npoint) int32.   tf_grouping.py:53-66"""
    T.require(float(radius) > 0, "QueryBallPoint expects positive radius")
    T.require(int(nsample) > 0, "QueryBallPoint expects positive nsample")
    xyz1 = T.f32_cuda(xyz1, "xyz1")
    xyz2 = T.f32_cuda(xyz2, "xyz2")
    _check_xyz("QueryBallPoint", xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    st = N.lib().sa_query_ball_point(b, n, m, float(radius), int(nsample), xyz1.data_ptr(),
                                     xyz2.data_ptr(), idx.data_ptr(), cnt.data_ptr(), N.current_stream())
    N.check(st, "query_ball_point")
    return idx, cnt


def query_ball_point_dilated(min_radius, max_radius, nsample, xyz1, xyz2):
    """Ball query on the band min_radius <= d < max_radius (plus d == 0).   tf_grouping.py:68-83"""
    T.require(float(min_radius) >= 0, "QueryBallPointDilated expects positive min_radius")
    T.require(float(max_radius) > 0, "QueryBallPointDilated expects positive max_radius")
    T.require(int(nsample) > 0, "QueryBallPointDilated expects positive nsample")
    xyz1 = T.f32_cuda(xyz1, "xyz1")
    xyz2 = T.f32_cuda(xyz2, "xyz2")
    _check_xyz("QueryBallPointDilated", xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    st = N.lib().sa_query_ball_point_dilated(b, n, m, float(min_radius), float(max_radius), int(nsample),
                                             xyz1.data_ptr(), xyz2.data_ptr(), idx.data_ptr(),
                                             cnt.data_ptr(), N.current_stream())
    N.check(st, "query_ball_point_dilated")
    return idx, cnt


def group_point(points, idx):
    """points: (batch, ndataset, channel), idx: (batch, npoint, nsample) int32 ->
    (batch, npoint, nsample, channel); idx == -1 gives a zero row.   tf_grouping.py:114-122"""
    points = T.f32_cuda(points, "points")
    idx = T.i32_cuda(idx, "idx")
    T.require(points.dim() == 3, "GroupPoint expects (batch_size, num_points, channel) points shape")
    b, n, c = points.shape
    T.require(idx.dim() == 3 and idx.shape[0] == b, "GroupPoint expects (batch_size, npoints, nsample) idx shape")
    _, m, ns = idx.shape
    out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
    st = N.lib().sa_group_point(b, n, c, m, ns, points.data_ptr(), idx.data_ptr(), out.data_ptr(),
                                N.current_stream())
    N.check(st, "group_point")
    return out
