"""Interpolation operators with the reference's Python API (lib/utils/tf_ops/interpolation/tf_interpolate.py),
on torch-ROCm tensors, backed by csrc/interpolate.hip through include/sa_ops.h.

Same names, positional order, return arity/dtypes/shapes.  Forward only (ThreeNN is NoGradient in the reference,
tf_interpolate.py:19; the gradients of ThreeInterpolate / KInterpolate, :32-37,53-58, are out of scope).
Errors: ValueError with the reference's OP_REQUIRES messages (tf_interpolate.cpp:226-232,296-305).
"""
import torch

from .. import _tensor as T
from ... import _native as N


def three_nn(xyz1, xyz2):
    """xyz1: (b,n,3) unknown points, xyz2: (b,m,3) known points ->
    dist (b,n,3) float32 squared distances to the three nearest known points (ascending), idx (b,n,3) int32.
    tf_interpolate.py:8-18"""
    xyz1 = T.f32_cuda(xyz1, "xyz1")
    xyz2 = T.f32_cuda(xyz2, "xyz2")
    T.require(xyz1.dim() == 3 and xyz1.shape[2] == 3, "ThreeNN expects (b,n,3) xyz1 shape.")
    T.require(xyz2.dim() == 3 and xyz2.shape[2] == 3, "ThreeNN expects (b,m,3) xyz2 shape.")
    T.require(xyz1.shape[0] == xyz2.shape[0], "ThreeNN expects xyz1 and xyz2 with the same batch size")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    st = N.lib().sa_three_nn(b, n, m, xyz1.data_ptr(), xyz2.data_ptr(), dist.data_ptr(), idx.data_ptr(),
                             N.current_stream())
    N.check(st, "three_nn")
    return dist, idx


def _check_interp(op, points, idx, weight, k):
    T.require(points.dim() == 3, "%s expects (b,m,c) points shape" % op)
    T.require(idx.dim() == 3 and idx.shape[0] == points.shape[0] and (k is None or idx.shape[2] == k),
              "%s expects (b,n,%s) idx shape" % (op, "k" if k is None else k))
    T.require(tuple(weight.shape) == tuple(idx.shape), "%s expects weight with the shape of idx" % op)


def three_interpolate(points, idx, weight):
    """points: (b,m,c) known features, idx: (b,n,3) int32, weight: (b,n,3) -> (b,n,c).   tf_interpolate.py:21-31"""
    points = T.f32_cuda(points, "points")
    idx = T.i32_cuda(idx, "idx")
    weight = T.f32_cuda(weight, "weight")
    _check_interp("ThreeInterpolate", points, idx, weight, 3)
    b, m, c = points.shape
    n = idx.shape[1]
    out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
    st = N.lib().sa_three_interpolate(b, m, c, n, points.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr(),
                                      N.current_stream())
    N.check(st, "three_interpolate")
    return out


def k_interpolate(points, idx, weight):
    """points: (b,m,c), idx: (b,n,k) int32, weight: (b,n,k) -> (b,n,c).   tf_interpolate.py:42-52"""
    points = T.f32_cuda(points, "points")
    idx = T.i32_cuda(idx, "idx")
    weight = T.f32_cuda(weight, "weight")
    _check_interp("KInterpolate", points, idx, weight, None)
    b, m, c = points.shape
    n, k = idx.shape[1], idx.shape[2]
    out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
    st = N.lib().sa_k_interpolate(b, m, c, n, k, points.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr(),
                                  N.current_stream())
    N.check(st, "k_interpolate")
    return out


def _interpolate_grad(op, points, idx, weight, grad_out, k):
    points, idx = T.f32_cuda(points, "points"), T.i32_cuda(idx, "idx")
    weight, grad_out = T.f32_cuda(weight, "weight"), T.f32_cuda(grad_out, "grad_out")
    _check_interp(op, points, idx, weight, k)
    b, m, c = points.shape
    n = idx.shape[1]
    T.require(tuple(grad_out.shape) == (b, n, c), "%s expects (b,n,c) grad_out shape" % op)
    gp = torch.empty((b, m, c), dtype=torch.float32, device=points.device)
    N.check(N.lib().sa_k_interpolate_grad(b, n, c, m, idx.shape[2], grad_out.data_ptr(), idx.data_ptr(),
                                          weight.data_ptr(), gp.data_ptr(), N.current_stream()), op)
    return gp


def three_interpolate_grad(points, idx, weight, grad_out):
    """Gradient of three_interpolate w.r.t. points: [b,m,c] (float atomics; tf_interpolate.py:32-36)."""
    return _interpolate_grad("ThreeInterpolateGrad", points, idx, weight, grad_out, 3)


def k_interpolate_grad(points, idx, weight, grad_out):
    """Gradient of k_interpolate w.r.t. points: [b,m,c].   tf_interpolate.py:53-58"""
    return _interpolate_grad("KInterpolateGrad", points, idx, weight, grad_out, None)
