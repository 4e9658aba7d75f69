"""PostProcessor with the reference's interface (lib/builder/postprocessor.py:10-123): per frame and class,
tf.image.non_max_suppression on the BEV boxes, then gather.  Output tensors are fixed-size (max_output_size rows
per class, padded: index -1, zero boxes/scores) with a count, where the reference has TF dynamic shapes and can
therefore only stack batch size 1 (lib/core/evaluator.py:145-147)."""
import torch

from ..utils import _native as N
from ..utils.tf_ops import _tensor as T


class PostProcessor:
    def __init__(self, stage, cls_num, max_output_size=100, nms_threshold=0.1):
        # cfg.MODEL.FIRST_STAGE.MAX_OUTPUT_NUM / NMS_THRESH, configs/kitti/3dssd/3dssd.yaml:70-71
        assert stage == 0
        self.max_output_size = int(max_output_size)
        self.nms_threshold = float(nms_threshold)
        self.cls_num = int(cls_num)

    def nms(self, bev, pred_score):
        """bev [bs,n,4], pred_score [bs,n,cls] -> idx [bs,cls,max_out] int32 (-1 padded), cnt [bs,cls]."""
        bev = T.f32_cuda(bev, "bev")
        pred_score = T.f32_cuda(pred_score, "pred_score")
        bs, n, C = pred_score.shape
        idx = torch.empty((bs, C, self.max_output_size), dtype=torch.int32, device=bev.device)
        cnt = torch.empty((bs, C), dtype=torch.int32, device=bev.device)
        st = N.lib().sa_nms_bev(bs, n, C, self.max_output_size, self.nms_threshold, bev.data_ptr(),
                                pred_score.data_ptr(), idx.data_ptr(), cnt.data_ptr(), N.current_stream())
        N.check(st, "nms_bev")
        return idx, cnt

    def forward(self, pred_anchors_3d, pred_score, output_dict, bev=None):
        """pred_anchors_3d [bs,n,1,7] or [bs,n,7] (class-agnostic boxes of the anchor-free head), pred_score
        [bs,n,cls].  Appends pred_3d_bbox [bs,cls*max_out,7], pred_3d_score, pred_3d_cls_category, plus the raw
        nms_idx / nms_cnt."""
        T.require(pred_anchors_3d.dim() == 3 or (pred_anchors_3d.dim() == 4 and pred_anchors_3d.shape[2] == 1),
                  "PostProcessor: class-aware boxes [bs,n,cls,7] are not supported (the anchor-free head is "
                  "class-agnostic: [bs,n,1,7] or [bs,n,7])")
        boxes = pred_anchors_3d.reshape(pred_anchors_3d.shape[0], pred_anchors_3d.shape[1], 7).contiguous()
        bs, n, _ = boxes.shape
        if bev is None:
            bev = torch.empty((bs, n, 4), dtype=torch.float32, device=boxes.device)
            st = N.lib().sa_boxes_to_bev(bs * n, boxes.data_ptr(), bev.data_ptr(), N.current_stream())
            N.check(st, "boxes_to_bev")
        idx, cnt = self.nms(bev, pred_score)
        C, K = self.cls_num, self.max_output_size
        valid = idx >= 0
        safe = idx.clamp(min=0).long()
        flat = safe.reshape(bs, C * K)
        gb = torch.gather(boxes, 1, flat[..., None].expand(-1, -1, 7)) * valid.reshape(bs, C * K, 1)
        sc = torch.gather(pred_score.transpose(1, 2).reshape(bs, C, n), 2, safe) * valid
        cat = torch.arange(C, device=boxes.device, dtype=torch.int32)[None, :, None].expand(bs, C, K)
        output_dict.setdefault("pred_3d_bbox", []).append(gb)
        output_dict.setdefault("pred_3d_score", []).append(sc.reshape(bs, C * K))
        output_dict.setdefault("pred_3d_cls_category", []).append(torch.where(valid, cat, torch.full_like(cat, -1)).reshape(bs, C * K))
        output_dict.setdefault("nms_idx", []).append(idx)
        output_dict.setdefault("nms_cnt", []).append(cnt)
        return output_dict
