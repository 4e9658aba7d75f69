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

    def class_unaware_format(self, pred_anchors_3d, pred_score):
        """postprocessor.py:24-44 (RPN proposals of a class-aware head): pred_anchors_3d [bs,n,1|cls,7], pred_score
        [bs,n,cls] -> ([bs,n,1,7], [bs,n,1]): the best class's score and, for class-aware boxes, that class's box
        (first maximum on ties, like tf.argmax; the reference's one-hot multiply-and-sum selects exactly that row)."""
        pred_score = T.f32_cuda(pred_score, "pred_score")
        boxes = T.f32_cuda(pred_anchors_3d, "pred_anchors_3d")
        T.require(boxes.dim() == 4 and boxes.shape[3] == 7, "class_unaware_format expects boxes [bs,n,1|cls,7]")
        score, cls = pred_score.max(dim=-1, keepdim=True)
        if boxes.shape[2] == 1:
            return boxes, score
        T.require(boxes.shape[2] == pred_score.shape[2], "class-aware boxes need one box per score channel")
        # torch.max returns the FIRST maximum for ties on ROCm as well; stated explicitly through argmax of (score == max)
        first = (pred_score == score).int().argmax(dim=-1, keepdim=True)
        sel = torch.gather(boxes, 2, first[..., None].expand(-1, -1, 1, 7))
        return sel, score

    def forward(self, pred_anchors_3d, pred_score, output_dict, bev=None):
        """postprocessor.py:49-123.  pred_anchors_3d [bs,n,7] / [bs,n,1,7] (class-agnostic boxes of the anchor-free head)
        or [bs,n,cls,7] (class-aware: class i is suppressed on box set reg_i = min(i, k-1), :76-80), pred_score
        [bs,n,cls'].  cls' != cls_num -> class_unaware_format first (:56-59).  Appends pred_3d_bbox
        [bs,cls*max_out,7], pred_3d_score, pred_3d_cls_category (fixed size: max_out rows per class, padded with zero
        boxes / scores and category -1), plus the raw nms_idx / nms_cnt."""
        T.require(pred_anchors_3d.dim() in (3, 4) and pred_anchors_3d.shape[-1] == 7, "PostProcessor: boxes must be [bs,n,(k,)7]")
        if pred_anchors_3d.dim() == 3:
            pred_anchors_3d = pred_anchors_3d[:, :, None, :]
        pred_score = T.f32_cuda(pred_score, "pred_score")
        if pred_score.shape[-1] != self.cls_num:
            pred_anchors_3d, pred_score = self.class_unaware_format(pred_anchors_3d, pred_score)
        k = pred_anchors_3d.shape[2]
        if k > 1:
            return self._forward_class_aware(T.f32_cuda(pred_anchors_3d, "pred_anchors_3d"), pred_score, output_dict)
        return self._forward_agnostic(pred_anchors_3d, pred_score, output_dict, bev)

    def _forward_class_aware(self, boxes4, pred_score, output_dict):
        """One NMS per class on that class's own boxes (reg_i = min(i, k-1))."""
        bs, n, k, _ = boxes4.shape
        C, K = self.cls_num, self.max_output_size
        T.require(pred_score.shape[-1] == C, "pred_score must have cls_num channels here")
        bev_all = torch.empty((bs, n, k, 4), dtype=torch.float32, device=boxes4.device)
        st = N.lib().sa_boxes_to_bev(bs * n * k, boxes4.data_ptr(), bev_all.data_ptr(), N.current_stream())
        N.check(st, "boxes_to_bev")
        idx_l, cnt_l = [], []
        for i in range(C):
            r = min(i, k - 1)
            s_i = pred_score[:, :, i:i + 1].contiguous()
            sub = PostProcessor(0, 1, K, self.nms_threshold)
            idx, cnt = sub.nms(bev_all[:, :, r].contiguous(), s_i)                    # [bs,1,K], [bs,1]
            idx_l.append(idx)
            cnt_l.append(cnt)
        idx_all, cnt_all = torch.cat(idx_l, 1).contiguous(), torch.cat(cnt_l, 1)      # [bs,C,K], [bs,C]
        # the kept rows of every class from ITS box set (min(i, k-1)): one launch (csrc/head.hip, sa_nms_gather with kbox = k)
        scores = pred_score.contiguous()
        gb = torch.empty((bs, C * K, 7), dtype=torch.float32, device=boxes4.device)
        sc = torch.empty((bs, C * K), dtype=torch.float32, device=boxes4.device)
        cat = torch.empty((bs, C * K), dtype=torch.int32, device=boxes4.device)
        st = N.lib().sa_nms_gather(bs, n, C, K, k, boxes4.contiguous().data_ptr(), scores.data_ptr(), idx_all.data_ptr(), gb.data_ptr(),
                                   sc.data_ptr(), cat.data_ptr(), N.current_stream())
        N.check(st, "nms_gather")
        output_dict.setdefault("pred_3d_bbox", []).append(gb)
        output_dict.setdefault("pred_3d_score", []).append(sc)
        output_dict.setdefault("pred_3d_cls_category", []).append(cat)
        output_dict.setdefault("nms_idx", []).append(idx_all)
        output_dict.setdefault("nms_cnt", []).append(cnt_all)
        return output_dict

    def _forward_agnostic(self, pred_anchors_3d, pred_score, output_dict, bev=None):
        boxes = pred_anchors_3d.reshape(pred_anchors_3d.shape[0], pred_anchors_3d.shape[1], 7).contiguous()
        bs, n, _ = boxes.shape
        if bev is None:
            bev = torch.empty((bs, n, 4), dtype=torch.float32, device=boxes.device)
            st = N.lib().sa_boxes_to_bev(bs * n, boxes.data_ptr(), bev.data_ptr(), N.current_stream())
            N.check(st, "boxes_to_bev")
        idx, cnt = self.nms(bev, pred_score)
        C, K = self.cls_num, self.max_output_size
        # the kept rows as fixed-size tensors: one launch (csrc/head.hip), no torch gather / where on the data path
        gb = torch.empty((bs, C * K, 7), dtype=torch.float32, device=boxes.device)
        sc = torch.empty((bs, C * K), dtype=torch.float32, device=boxes.device)
        cat = torch.empty((bs, C * K), dtype=torch.int32, device=boxes.device)
        st = N.lib().sa_nms_gather(bs, n, C, K, 1, boxes.data_ptr(), pred_score.data_ptr(), idx.data_ptr(), gb.data_ptr(),
                                   sc.data_ptr(), cat.data_ptr(), N.current_stream())
        N.check(st, "nms_gather")
        output_dict.setdefault("pred_3d_bbox", []).append(gb)
        output_dict.setdefault("pred_3d_score", []).append(sc)
        output_dict.setdefault("pred_3d_cls_category", []).append(cat)
        output_dict.setdefault("nms_idx", []).append(idx)
        output_dict.setdefault("nms_cnt", []).append(cnt)
        return output_dict
