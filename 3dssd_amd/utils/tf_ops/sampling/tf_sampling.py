"""Sampling operators with the reference's Python API (lib/utils/tf_ops/sampling/tf_sampling.py),
on torch-ROCm tensors instead of tf.Tensor, backed by the gfx950 kernels in csrc/fps.hip and
csrc/gather.hip through the C ABI of include/sa_ops.h.

Same names, positional order (scalars first), return dtypes/shapes.  Forward only: the reference
registers FPS as NoGradient (tf_sampling.py:52,63); GatherPoint's gradient (tf_sampling.py:38-42) is
out of scope.  Argument errors are raised as ValueError with the reference's OP_REQUIRES messages
(lib/utils/tf_ops/sampling/tf_sampling.cpp:136,142,169,175,241,246).
"""
import torch

from .. import _tensor as T
from ... import _native as N


def farthest_point_sample(npoint, inp):
    """inp: (batch, ndataset, c) float32 -> (batch, npoint) int32.   tf_sampling.py:43-51"""
    T.require(int(npoint) > 0, "FarthestPointSample expects positive npoint")
    inp = T.f32_cuda(inp, "inp")
    T.require(inp.dim() == 3, "FarthestPointSample expects (batch_size,num_points,c) inp shape")
    b, n, c = inp.shape
    out = torch.empty((b, int(npoint)), dtype=torch.int32, device=inp.device)
    # allocate_temp [b,n] of tf_sampling.cpp:153; only read by the global-scratch kernel
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device) if (c != 3 or n > 16384) else None
    st = N.lib().sa_farthest_point_sample(b, n, c, int(npoint), inp.data_ptr(),
                                          temp.data_ptr() if temp is not None else None,
                                          out.data_ptr(), N.current_stream())
    N.check(st, "farthest_point_sample")
    return out


def farthest_point_sample_with_distance(npoint, dist):
    """dist: (batch, ndataset, ndataset) float32 -> (batch, npoint) int32.   tf_sampling.py:54-62"""
    T.require(int(npoint) > 0, "FarthestPointSampleWithDistance expects positive npoint")
    dist = T.f32_cuda(dist, "dist")
    T.require(dist.dim() == 3 and dist.shape[1] == dist.shape[2],
              "FarthestPointSampleWithDistance expects (batch_size,num_points,num_points) inp shape")
    b, n, _ = dist.shape
    out = torch.empty((b, int(npoint)), dtype=torch.int32, device=dist.device)
    temp = torch.empty((b, n), dtype=torch.float32, device=dist.device) if n > 16384 else None
    st = N.lib().sa_farthest_point_sample_with_distance(b, n, int(npoint), dist.data_ptr(),
                                                        temp.data_ptr() if temp is not None else None,
                                                        out.data_ptr(), N.current_stream())
    N.check(st, "farthest_point_sample_with_distance")
    return out


def gather_point(inp, idx):
    """inp: (batch, ndataset, c) float32, idx: (batch, npoints) int32 -> (batch, npoints, c).
    tf_sampling.py:24-32"""
    inp = T.f32_cuda(inp, "inp")
    idx = T.i32_cuda(idx, "idx")
    T.require(inp.dim() == 3, "GatherPoint expects (batch_size,num_points,c) inp shape")
    b, n, c = inp.shape
    T.require(idx.dim() == 2 and idx.shape[0] == b, "GatherPoint expects (batch_size,num_result) idx shape")
    m = idx.shape[1]
    out = torch.empty((b, m, c), dtype=torch.float32, device=inp.device)
    st = N.lib().sa_gather_point(b, n, m, c, inp.data_ptr(), idx.data_ptr(), out.data_ptr(),
                                 N.current_stream())
    N.check(st, "gather_point")
    return out


def gather_point_grad(inp, idx, out_g):
    """Gradient of gather_point w.r.t. inp: out_g [b,m,c] scatter-added to [b,n,c] (float atomics, like
    scatteraddpointKernel, tf_sampling_g.cu:339-351).   tf_sampling.py:38-42 / GatherPointGrad"""
    inp, idx, out_g = T.f32_cuda(inp, "inp"), T.i32_cuda(idx, "idx"), T.f32_cuda(out_g, "out_g")
    T.require(inp.dim() == 3, "GatherPointGradGpuOp expects (batch_size,num_points,c) inp")
    b, n, c = inp.shape
    T.require(idx.dim() == 2 and idx.shape[0] == b, "GatherPointGradGpuOp expects (batch_size,num_result) idx shape")
    m = idx.shape[1]
    T.require(tuple(out_g.shape) == (b, m, c), "GatherPointGradGpuOp expects (batch_size,num_result,c) out_g shape")
    inp_g = torch.empty((b, n, c), dtype=torch.float32, device=inp.device)
    N.check(N.lib().sa_gather_point_grad(b, n, m, c, out_g.data_ptr(), idx.data_ptr(), inp_g.data_ptr(),
                                         N.current_stream()), "gather_point_grad")
    return inp_g


def gather_by_mask(proposal_num, inp, mask):
    """The first proposal_num rows of inp [b,n,c] whose mask [b,n] (float) truncates to non-zero, in point order;
    missing rows repeat the first selected one; a frame with no selected row gives zeros.  -> [b,proposal_num,c].
    tf_sampling.py:76-85"""
    T.require(int(proposal_num) > 0, "GatherByMask expects positive proposal number")
    inp, mask = T.f32_cuda(inp, "inp"), T.f32_cuda(mask, "mask")
    T.require(inp.dim() == 3, "GatherByMask expects (bs,num_points,c) inp shape")
    b, n, c = inp.shape
    T.require(tuple(mask.shape) == (b, n), "GatherByMask expects (bs,num_points) mask shape")
    out = torch.empty((b, int(proposal_num), c), dtype=torch.float32, device=inp.device)
    sel = torch.empty((b, int(proposal_num)), dtype=torch.int32, device=inp.device)
    N.check(N.lib().sa_gather_by_mask(b, n, c, int(proposal_num), inp.data_ptr(), mask.data_ptr(), out.data_ptr(),
                                      sel.data_ptr(), N.current_stream()), "gather_by_mask")
    return out


def farthest_point_sample_with_preidx(npoint, inp, preidx):
    """FPS continued from an already chosen set: the running minimum starts as the distance to the nearest of
    preidx [b,m1]; the first output is the point farthest from that set (lowest index on ties), then npoint-1
    ordinary iterations.  inp [b,n,c] -> int32 [b,npoint].   tf_sampling.py:65-74"""
    T.require(int(npoint) > 0, "FarthestPointSampleWithPreidx expects positive npoint")
    inp, preidx = T.f32_cuda(inp, "inp"), T.i32_cuda(preidx, "preidx")
    T.require(inp.dim() == 3, "FarthestPointSampleWithPreidx expects (batch_size,num_points,c) inp shape")
    b, n, c = inp.shape
    T.require(preidx.dim() == 2 and preidx.shape[0] == b, "FarthestPointSampleWithPreidx expects (batch_size,m1) preidx shape")
    out = torch.empty((b, int(npoint)), dtype=torch.int32, device=inp.device)
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)
    N.check(N.lib().sa_farthest_point_sample_with_preidx(b, n, c, int(npoint), preidx.shape[1], inp.data_ptr(),
                                                         preidx.data_ptr(), temp.data_ptr(), out.data_ptr(),
                                                         N.current_stream()), "farthest_point_sample_with_preidx")
    return out


def prob_sample(inp, inpr):
    """tf_sampling.py:8-16 of the reference (inverse-CDF sampling).  It has NO caller in the reference (SURVEY.md section 2,
    out of scope) and was removed from this library in round 4; the name stays so that code written against the
    reference's module gets a clear error instead of an AttributeError at import."""
    raise NotImplementedError("prob_sample is outside the set-abstraction path (no caller in the reference, SURVEY.md section 2): "
                              "not provided by 3dssd_amd -- see DESIGN.md section 8")
