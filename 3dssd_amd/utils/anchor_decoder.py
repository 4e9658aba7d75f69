"""decode_dist_anchor_free with the reference's signature (lib/utils/anchor_decoder.py:86-112) on torch-ROCm
tensors, plus the fused form the detector uses (decode + sigmoid + BEV box in one kernel, csrc/head.hip)."""
import torch

from . import _native as N
from .tf_ops import _tensor as T


def decode_scores_bev(center_xyz, pred_reg, pred_cls, angle_cls_num):
    """center_xyz [bs,n,3], pred_reg [bs,n,6+2A], pred_cls [bs,n,C] -> boxes [bs,n,7], scores [bs,n,C], bev [bs,n,4]."""
    center_xyz = T.f32_cuda(center_xyz, "center_xyz")
    pred_reg = T.f32_cuda(pred_reg, "pred_reg")
    pred_cls = T.f32_cuda(pred_cls, "pred_cls")
    bs, n, _ = center_xyz.shape
    A, C = int(angle_cls_num), pred_cls.shape[-1]
    T.require(pred_reg.shape[-1] == 6 + 2 * A, "pred_reg must be [bs,n,6+2*ANGLE_CLS_NUM]")
    boxes = torch.empty((bs, n, 7), dtype=torch.float32, device=center_xyz.device)
    scores = torch.empty((bs, n, C), dtype=torch.float32, device=center_xyz.device)
    bev = torch.empty((bs, n, 4), dtype=torch.float32, device=center_xyz.device)
    st = N.lib().sa_decode_anchor_free(bs, n, A, C, center_xyz.data_ptr(), pred_reg.data_ptr(), pred_cls.data_ptr(),
                                       boxes.data_ptr(), scores.data_ptr(), bev.data_ptr(), N.current_stream())
    N.check(st, "decode_anchor_free")
    return boxes, scores, bev


def decode_dist_anchor_free(center_xyz, det_forced_6_distance, det_angle_cls, det_angle_res, is_training=False):
    """anchor_decoder.py:86-112: [bs,n,3], [bs,n,6], [bs,n,A], [bs,n,A] -> pred_anchors_3d [bs,n,7]."""
    A = det_angle_cls.shape[-1]
    reg = torch.cat([det_forced_6_distance, det_angle_cls, det_angle_res], -1).contiguous()
    zeros = torch.zeros(reg.shape[:2] + (1,), dtype=torch.float32, device=reg.device)
    boxes, _s, _b = decode_scores_bev(center_xyz, reg, zeros, A)
    return boxes
