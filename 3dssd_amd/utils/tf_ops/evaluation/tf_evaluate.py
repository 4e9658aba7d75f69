"""calc_iou / calc_iou_match / calc_iou_match_warper with the reference's Python API
(lib/utils/tf_ops/evaluation/tf_evaluate.py:26-55) on torch-ROCm tensors, backed by csrc/iou.hip.  The KITTI
`evaluate` op (precision / AOS curves from result files, tf_evaluate.py:9-24) is an offline CPU tool of the reference
and is not provided."""
import torch

from .. import _tensor as T
from ... import _native as N


def calc_iou(detections, groundtruths):
    """detections [bs, dets_num, 7], groundtruths [bs, gt_num, 7] (x, bottom y, z, l, h, w, ry) ->
    (iou_bev, iou_3d), both [bs, dets_num, gt_num].   tf_evaluate.py:26-33"""
    detections, groundtruths = T.f32_cuda(detections, "detections"), T.f32_cuda(groundtruths, "groundtruths")
    T.require(detections.dim() == 3 and detections.shape[2] == 7, "Calculate IoU expects (bs, -1, 7) detections shape")
    bs, dn, _ = detections.shape
    T.require(groundtruths.dim() == 3 and groundtruths.shape[0] == bs and groundtruths.shape[2] == 7,
              "Calculate IoU expects (bs, -1, 7) gt shape")
    gn = groundtruths.shape[1]
    bev = torch.empty((bs, dn, gn), dtype=torch.float32, device=detections.device)
    i3d = torch.empty((bs, dn, gn), dtype=torch.float32, device=detections.device)
    N.check(N.lib_extra().sa_calc_iou(bs, dn, gn, detections.data_ptr(), groundtruths.data_ptr(), bev.data_ptr(), i3d.data_ptr(),
                                N.current_stream()), "calc_iou")
    return bev, i3d


def calc_iou_match(detections, groundtruths):
    """Row-by-row IoU: detections [n, 7] against groundtruths [n, 7] -> (iou_bev [n], iou_3d [n]).   tf_evaluate.py:35-42"""
    detections, groundtruths = T.f32_cuda(detections, "detections"), T.f32_cuda(groundtruths, "groundtruths")
    T.require(detections.dim() == 2 and detections.shape[1] == 7, "Calculate IoU expects (-1, 7) detections shape")
    n = detections.shape[0]
    T.require(tuple(groundtruths.shape) == (n, 7), "Calculate IoU expects (-1, 7) gt shape")
    bev = torch.empty((n,), dtype=torch.float32, device=detections.device)
    i3d = torch.empty((n,), dtype=torch.float32, device=detections.device)
    N.check(N.lib_extra().sa_calc_iou_match(n, detections.data_ptr(), groundtruths.data_ptr(), bev.data_ptr(), i3d.data_ptr(),
                                      N.current_stream()), "calc_iou_match")
    return bev, i3d


def calc_iou_match_warper(detections, groundtruths):
    """[..., 7] x [..., 7] -> (iou_bev [...], iou_3d [...]).   tf_evaluate.py:45-55"""
    shape = tuple(detections.shape[:-1])
    bev, i3d = calc_iou_match(detections.reshape(-1, 7), groundtruths.reshape(-1, 7))
    return bev.reshape(shape), i3d.reshape(shape)
