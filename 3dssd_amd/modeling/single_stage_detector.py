"""Points -> boxes: the inference forward of the reference's SingleStageDetector
(lib/modeling/single_stage_detector.py:115-125 network_forward, :195-228 test_forward) with the SA backbone of
backbone.py, the 'Det' head (head_builder.py:81-113), anchor-free decoding and per-class BEV NMS.

`DetectionHead` is everything behind the backbone as a sequence of C-ABI launches on the current stream -- no torch
operator on the data path, no allocation that depends on data -- so the staged executor (pipeline.SAPipeline(tail=...))
captures it behind stage B and its tickets return boxes and scores (`Ticket.detections()`); `SingleStageDetector` is the
eager form of the same calls, which is what the executor's outputs are compared with bit for bit."""
import torch

from ..backbone import SABackbone
from ..builder.postprocessor import PostProcessor
from ..utils import _native as N
from ..utils import head_util
from ..utils.anchor_decoder import decode_scores_bev
from ..utils.layers_util import _dense

# the keys a ticket of the executor hands out (a batch's part of the package's static tensors)
DETECTION_KEYS = ("pred_3d_bbox", "pred_3d_score", "pred_3d_cls_category", "nms_idx", "nms_cnt", "pred_anchors_3d", "pred_score")


def _joined(tensors):
    """The tensors of a list-index selection along the point axis (head_builder.py:91-92 `tf.concat(..., axis=1)`): the
    tensor itself for one index (3dssd.yaml:68), one strided block copy per part otherwise -- no torch.cat."""
    if len(tensors) == 1:
        return tensors[0]
    b, c = tensors[0].shape[0], tensors[0].shape[2]
    out = torch.empty((b, sum(t.shape[1] for t in tensors), c), dtype=torch.float32, device=tensors[0].device)
    off = 0
    for t in tensors:
        N.copy_blocks([(t, out[:, off:off + t.shape[1]], b, t.shape[1], c)])
        off += t.shape[1]
    return out


class DetectionHead:
    def __init__(self, variables, head_cfg, cls_num=1, angle_cls_num=12, max_output_size=100, nms_threshold=0.1):
        # head_cfg: one HEAD row [xyz_index, feature_index, op_type, mlp_list, bn, layer_type, scope]
        # (configs/kitti/3dssd/3dssd.yaml:68)
        self.vs = variables
        (self.xyz_index, self.feature_index, self.op_type, self.mlp_list, self.bn, self.layer_type,
         self.scope) = head_cfg
        assert self.op_type == "conv1d" and self.layer_type == "Det"
        self.cls_num = int(cls_num)          # Sigmoid: pred_cls_channel = number of classes (head_builder.py:35-38)
        self.angle_cls_num = int(angle_cls_num)
        head_util.ANGLE_CLS_NUM = self.angle_cls_num
        self.postprocessor = PostProcessor(0, self.cls_num, max_output_size, nms_threshold)

    def features(self, xyz_list, feature_list):
        """head_builder.py:81-113: -> output dict with pred_cls / pred_offset / pred_angle_* / pred_reg_raw and the key points."""
        out = {}
        xyz = _joined([xyz_list[i] for i in self.xyz_index])
        feat = _joined([feature_list[i] for i in self.feature_index])
        pre = self.scope + "/" if self.scope else ""
        for i, _ch in enumerate(self.mlp_list):                                   # head_builder.py:97-98
            feat = _dense(feat, self.vs.layer(pre + "conv1d_%d" % i, self.bn), relu=True)
        head_util.box_regression_head(feat, self.cls_num, 1, 6, self.bn, False, False, None, None, out,
                                      scope=self.scope, variables=self.vs)
        out["key_output_xyz"] = [xyz]
        out["key_output_feature"] = [feat]
        return out

    def decode(self, out, index=0):
        """single_stage_detector.py:195-228: decode, sigmoid, BEV boxes, NMS, gather."""
        base_xyz = out["key_output_xyz"][index]
        boxes, scores, bev = decode_scores_bev(base_xyz, out["pred_reg_raw"][index], out["pred_cls"][index],
                                               self.angle_cls_num)
        out["pred_anchors_3d"] = [boxes]
        out["pred_score"] = [scores]
        out["pred_bev"] = [bev]
        self.postprocessor.forward(boxes, scores, out, bev=bev)
        return out

    def __call__(self, lists):
        """lists = (xyz_list, feature_list, fps_idx_list) of SABackbone.forward -> {key: tensor} (DETECTION_KEYS)."""
        out = self.decode(self.features(lists[0], lists[1]))
        return {k: out[k][0] for k in DETECTION_KEYS}


def detection_tail(head_cfg, **kw):
    """A `tail=` for SAPipeline: built from the pipeline's own network (its VariableStore holds the head's layers)."""
    def make(net):
        return DetectionHead(net.variables, head_cfg, **kw)
    make._is_tail_factory = True
    return make


class SingleStageDetector:
    def __init__(self, arch, head_cfg, params, device="cuda:0", cls_num=1, angle_cls_num=12,
                 max_translate_range=(-3.0, -2.0, -3.0), max_output_size=100, nms_threshold=0.1, backbone=None):
        self.backbone = backbone if backbone is not None else SABackbone(arch, params, device, max_translate_range)
        self.vs = self.backbone.variables
        self.head = DetectionHead(self.vs, head_cfg, cls_num, angle_cls_num, max_output_size, nms_threshold)
        self.cls_num, self.angle_cls_num, self.postprocessor = self.head.cls_num, self.head.angle_cls_num, self.head.postprocessor
        self.scope, self.bn, self.mlp_list = self.head.scope, self.head.bn, self.head.mlp_list

    def network_forward(self, point_cloud):
        xyz_list, feature_list, _ = self.backbone(point_cloud)
        return self.head.features(xyz_list, feature_list)

    def test_forward(self, out, index=0):
        return self.head.decode(out, index)

    def __call__(self, point_cloud):
        return self.test_forward(self.network_forward(point_cloud))
