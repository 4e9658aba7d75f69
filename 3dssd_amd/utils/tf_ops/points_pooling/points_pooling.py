"""points_pooling with the reference's Python API (lib/utils/tf_ops/points_pooling/points_pooling.py), on torch-ROCm
tensors, backed by csrc/pooling.hip through include/sa_ops.h.  Forward plus the gradient as a plain function
(`points_pooling_grad`, no autograd registration); errors are ValueError with the reference's OP_REQUIRES messages
(tf_points_pooling.cpp:64-99)."""
import torch

from .. import _tensor as T
from ... import _native as N


def points_pooling(pc, box_3d, pc_loc, l=7, h=7, w=7, sample_num=35):
    """pc [bs, proposal_num, pts, c], box_3d [bs, proposal_num, 6] (cx, bottom y, cz, l, h, w), pc_loc [bs, proposal_num,
    pts, 3] -> (out_features [bs, P, l, h, w, sample_num, c], out_idx [bs, P, l, h, w, sample_num] int32,
    out_points_num [bs, P, l, h, w] int32, pillars [bs, P, l, h, w, 3]).   points_pooling.py:10-21"""
    T.require(int(l) > 0, "PointsPooling method expects positive length")
    T.require(int(h) > 0, "PointsPooling method expects positive height")
    T.require(int(w) > 0, "PointsPooling method expects positive width")
    T.require(int(sample_num) > 0, "PointsPooling method expects positive sample number")
    pc, box_3d, pc_loc = T.f32_cuda(pc, "pc"), T.f32_cuda(box_3d, "box_3d"), T.f32_cuda(pc_loc, "pc_loc")
    T.require(pc.dim() == 4, "PointsPooling expects (bs, proposal_num, num_points, channel) pc shape")
    bs, pn, pts, c = pc.shape
    T.require(tuple(box_3d.shape) == (bs, pn, 6), "PointsPooling expects (bs, proposal_num, 6) proposal shape")
    T.require(tuple(pc_loc.shape) == (bs, pn, pts, 3), "PointsPooling expects (bs, proposal_num, num_points, 3) pc_loc shape")
    l, h, w, sample_num = int(l), int(h), int(w), int(sample_num)
    dev = pc.device
    feats = torch.empty((bs, pn, l, h, w, sample_num, c), dtype=torch.float32, device=dev)
    idx = torch.empty((bs, pn, l, h, w, sample_num), dtype=torch.int32, device=dev)
    num = torch.empty((bs, pn, l, h, w), dtype=torch.int32, device=dev)
    pillars = torch.empty((bs, pn, l, h, w, 3), dtype=torch.float32, device=dev)
    N.check(N.lib_extra().sa_points_pooling(bs, pn, pts, c, l, h, w, sample_num, pc.data_ptr(), box_3d.data_ptr(), pc_loc.data_ptr(),
                                      feats.data_ptr(), idx.data_ptr(), num.data_ptr(), pillars.data_ptr(),
                                      N.current_stream()), "points_pooling")
    return feats, idx, num, pillars


def points_pooling_grad(pc, out_idx, sampled_num_lists, features_grad):
    """Gradient w.r.t. pc [bs, P, pts, c] (float atomics like the reference, tf_points_pooling_g.cu:131-153).
    points_pooling.py:22-29"""
    pc, features_grad = T.f32_cuda(pc, "pc"), T.f32_cuda(features_grad, "features_grad")
    out_idx, sampled_num_lists = T.i32_cuda(out_idx, "out_idx"), T.i32_cuda(sampled_num_lists, "sampled_num_lists")
    T.require(pc.dim() == 4, "PointsPoolingGrad expects (bs, proposal_num, num_points, channel) pc shape")
    bs, pn, pts, c = pc.shape
    T.require(out_idx.dim() == 6 and tuple(out_idx.shape[:2]) == (bs, pn), "PointsPoolingGrad expects (bs, proposal_num, l, h, w, sample_num) out_idx shape")
    _, _, l, h, w, sample_num = out_idx.shape
    T.require(tuple(sampled_num_lists.shape) == (bs, pn, l, h, w), "PointsPoolingGrad expects (bs, proposal_num, l, h, w) sampled_num_lists shape")
    T.require(tuple(features_grad.shape) == (bs, pn, l, h, w, sample_num, c),
              "PointsPoolingGrad expects (bs, proposal_num, l, h, w, sample_num, channel) features_grad shape")
    out = torch.empty((bs, pn, pts, c), dtype=torch.float32, device=pc.device)
    N.check(N.lib_extra().sa_points_pooling_grad(bs, pn, pts, c, l, h, w, sample_num, out_idx.data_ptr(), sampled_num_lists.data_ptr(),
                                           features_grad.data_ptr(), out.data_ptr(), N.current_stream()), "points_pooling_grad")
    return out
