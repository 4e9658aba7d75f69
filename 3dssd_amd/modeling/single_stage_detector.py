"""Points -> boxes: the inference forward of the reference's SingleStageDetector
(lib/modeling/single_stage_detector.py:115-125 network_forward, :195-228 test_forward) with the SA backbone of
backbone.py, the 'Det' head (head_builder.py:81-113), anchor-free decoding and per-class BEV NMS."""
import torch

from ..backbone import SABackbone
from ..builder.postprocessor import PostProcessor
from ..utils import head_util
from ..utils.anchor_decoder import decode_scores_bev
from ..utils.layers_util import _dense


class SingleStageDetector:
    def __init__(self, arch, head_cfg, params, device="cuda:0", cls_num=1, angle_cls_num=12,
                 max_translate_range=(-3.0, -2.0, -3.0), max_output_size=100, nms_threshold=0.1):
        # head_cfg: one HEAD row [xyz_index, feature_index, op_type, mlp_list, bn, layer_type, scope]
        # (configs/kitti/3dssd/3dssd.yaml:68)
        self.backbone = SABackbone(arch, params, device, max_translate_range)
        self.vs = self.backbone.variables
        (self.xyz_index, self.feature_index, self.op_type, self.mlp_list, self.bn, self.layer_type,
         self.scope) = head_cfg
        assert self.op_type == "conv1d" and self.layer_type == "Det"
        self.cls_num = int(cls_num)          # Sigmoid: pred_cls_channel = number of classes (head_builder.py:35-38)
        self.angle_cls_num = int(angle_cls_num)
        head_util.ANGLE_CLS_NUM = self.angle_cls_num
        self.postprocessor = PostProcessor(0, self.cls_num, max_output_size, nms_threshold)

    def network_forward(self, point_cloud):
        xyz_list, feature_list, _ = self.backbone(point_cloud)
        out = {}
        xyz = torch.cat([xyz_list[i] for i in self.xyz_index], 1)
        feat = torch.cat([feature_list[i] for i in self.feature_index], 1)
        pre = self.scope + "/" if self.scope else ""
        for i, _ch in enumerate(self.mlp_list):                                   # head_builder.py:97-98
            feat = _dense(feat, self.vs.layer(pre + "conv1d_%d" % i, self.bn), relu=True)
        head_util.box_regression_head(feat, self.cls_num, 1, 6, self.bn, False, False, None, None, out,
                                      scope=self.scope, variables=self.vs)
        out["key_output_xyz"] = [xyz]
        out["key_output_feature"] = [feat]
        return out

    def test_forward(self, out, index=0):
        base_xyz = out["key_output_xyz"][index]
        boxes, scores, bev = decode_scores_bev(base_xyz, out["pred_reg_raw"][index], out["pred_cls"][index],
                                               self.angle_cls_num)
        out["pred_anchors_3d"] = [boxes]
        out["pred_score"] = [scores]
        out["pred_bev"] = [bev]
        self.postprocessor.forward(boxes, scores, out, bev=bev)
        return out

    def __call__(self, point_cloud):
        return self.test_forward(self.network_forward(point_cloud))
