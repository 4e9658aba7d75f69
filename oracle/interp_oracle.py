"""CPU oracle for the interpolation operators of lib/utils/tf_ops/interpolation:
three_nn, three_interpolate, k_interpolate.

TEST INFRASTRUCTURE ONLY (see oracle/sa_oracle.py).

PARITY PINNED for three_nn and three_interpolate: the reference's op file carries CPU implementations
(threenn_cpu, threeinterpolate_cpu; tf_interpolate.cpp:86-156).  `make -C oracle ref` compiles them from
/root/reference into oracle/_ref/libtf_interpolate_ref.so (git-ignored; built in the build container only), the
numpy restatement below is checked against that library and against tests/golden/interp_ref.npz, vectors generated
from it by tests/golden/make_golden_interp.py.  Arithmetic of the CPU path (g++ -O2 on x86-64, no contraction):
every product and sum is a separate fp32 operation, d = ((dx*dx + dy*dy) + dz*dz), out = (p1*w1 + p2*w2) + p3*w3.
(The reference's CUDA kernels of the same ops would contract to FMAs under nvcc; the pinned path is the CPU one.)
k_interpolate has no CPU implementation in the reference: parity unpinned, restated in sa_oracle.c.
"""
import ctypes
import os

import numpy as np

from . import sa_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_SO = os.path.join(_HERE, "_ref", "libtf_interpolate_ref.so")
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
f32 = np.float32


def ref_lib():
    """The reference's own CPU functions (oracle/_ref), or None where they have not been built."""
    return ctypes.CDLL(_REF_SO) if os.path.exists(_REF_SO) else None


def ref_three_nn(xyz1, xyz2):
    lib = ref_lib()
    xyz1 = np.ascontiguousarray(xyz1, f32); xyz2 = np.ascontiguousarray(xyz2, f32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), f32); idx = np.empty((b, n, 3), np.int32)
    lib.threenn_cpu(b, n, m, xyz1.ctypes.data_as(_f32p), xyz2.ctypes.data_as(_f32p), dist.ctypes.data_as(_f32p),
                    idx.ctypes.data_as(_i32p))
    return dist, idx


def ref_three_interpolate(points, idx, weight):
    lib = ref_lib()
    points = np.ascontiguousarray(points, f32); idx = np.ascontiguousarray(idx, np.int32)
    weight = np.ascontiguousarray(weight, f32)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), f32)
    lib.threeinterpolate_cpu(b, m, c, n, points.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p),
                             weight.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
    return out


def ref_three_interpolate_grad(points_shape, idx, weight, grad_out):
    """threeinterpolate_grad_cpu (tf_interpolate.cpp:158-180); the op zeroes grad_points first (:356)."""
    lib = ref_lib()
    idx = np.ascontiguousarray(idx, np.int32); weight = np.ascontiguousarray(weight, f32)
    grad_out = np.ascontiguousarray(grad_out, f32)
    b, m, c = points_shape
    n = idx.shape[1]
    gp = np.zeros((b, m, c), f32)
    lib.threeinterpolate_grad_cpu(b, n, c, m, grad_out.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p),
                                  weight.ctypes.data_as(_f32p), gp.ctypes.data_as(_f32p))
    return gp


def three_nn(xyz1, xyz2):
    """tf_interpolate.py:8-18 -> tf_interpolate.cpp:86-132.  xyz1 [b,n,3] unknown, xyz2 [b,m,3] known ->
    (dist [b,n,3] squared distances ascending, idx [b,n,3]); equal distances keep index order (strict '<'
    insertion); with fewer than three known points the tail stays (1e40 -> inf, index 0)."""
    xyz1 = np.asarray(xyz1, f32); xyz2 = np.asarray(xyz2, f32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.full((b, n, 3), np.inf, f32); idx = np.zeros((b, n, 3), np.int32)
    for bi in range(b):
        for s in range(0, n, 2048):
            q = xyz1[bi, s:s + 2048]
            dx = (xyz2[bi, None, :, 0] - q[:, None, 0]).astype(f32)
            dy = (xyz2[bi, None, :, 1] - q[:, None, 1]).astype(f32)
            dz = (xyz2[bi, None, :, 2] - q[:, None, 2]).astype(f32)
            d = ((dx * dx).astype(f32) + (dy * dy).astype(f32)).astype(f32)
            d = (d + (dz * dz).astype(f32)).astype(f32)
            order = np.argsort(d, axis=1, kind="stable")[:, :3]
            k = order.shape[1]
            idx[bi, s:s + 2048, :k] = order
            dist[bi, s:s + 2048, :k] = np.take_along_axis(d, order, 1)
    return dist, idx


def three_interpolate(points, idx, weight):
    """tf_interpolate.py:21-31 -> tf_interpolate.cpp:134-156.  points [b,m,c], idx/weight [b,n,3] -> [b,n,c]."""
    points = np.asarray(points, f32); weight = np.asarray(weight, f32)
    b = points.shape[0]
    out = []
    for bi in range(b):
        p = points[bi][idx[bi]]                                  # [n,3,c]
        w = weight[bi][:, :, None]
        t = ((p[:, 0] * w[:, 0]).astype(f32) + (p[:, 1] * w[:, 1]).astype(f32)).astype(f32)
        out.append((t + (p[:, 2] * w[:, 2]).astype(f32)).astype(f32))
    return np.stack(out)


def k_interpolate(points, idx, weight):
    """tf_interpolate.py:42-52 -> tf_interpolate_g.cu:142-165 (no CPU implementation in the reference)."""
    points = np.ascontiguousarray(points, f32); idx = np.ascontiguousarray(idx, np.int32)
    weight = np.ascontiguousarray(weight, f32)
    b, m, c = points.shape
    n, k = idx.shape[1], idx.shape[2]
    out = np.empty((b, n, c), f32)
    O.lib().orc_k_interpolate(b, m, c, n, k, points.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p),
                              weight.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
    return out
