"""Detection head with the reference's function name and argument meaning (lib/utils/head_util.py:26-59):
box_regression_head builds pred_cls / pred_offset / pred_angle_cls / pred_angle_res from the head features with
1x1 conv1d layers (the same split-bf16 MFMA dense kernel as the aggregation layers)."""
from . import weights as W
from .layers_util import _dense

ANGLE_CLS_NUM = 12   # cfg.MODEL.ANGLE_CLS_NUM, configs/kitti/3dssd/3dssd.yaml:38


def box_regression_head(feature_input, pred_cls_channel, pred_reg_base_num, pred_reg_channel_num, bn, is_training,
                        pred_attr_velo, conv_op, bn_decay, output_dict, scope="", variables=None):
    """head_util.py:26-59.  feature_input [bs, points_num, c].  Appends to output_dict (keys of
    lib/dataset/maps_dict.py): pred_cls [bs,n,cls], pred_offset [bs,n,base,reg], pred_angle_cls / pred_angle_res
    [bs,n,base,ANGLE_CLS_NUM]; also returns the un-split pred_reg [bs,n,base*(reg+2A)] for the fused decoder."""
    assert not pred_attr_velo, "attribute / velocity heads (nuScenes) are outside this step"
    vs = variables or W.default_variables()
    pre = scope + "/" if scope else ""
    bs, points_num, _ = feature_input.shape
    pred_cls = _dense(feature_input, vs.layer(pre + "pred_cls_base", bn), relu=True)
    pred_cls = _dense(pred_cls, vs.layer(pre + "pred_cls", False), relu=False)
    assert pred_cls.shape[-1] == pred_cls_channel
    pred_reg = _dense(feature_input, vs.layer(pre + "pred_reg_base", bn), relu=True)
    pred_reg = _dense(pred_reg, vs.layer(pre + "pred_reg", False), relu=False)
    ch = pred_reg_channel_num + ANGLE_CLS_NUM * 2
    assert pred_reg.shape[-1] == pred_reg_base_num * ch
    r4 = pred_reg.view(bs, points_num, pred_reg_base_num, ch)
    output_dict.setdefault("pred_cls", []).append(pred_cls)
    output_dict.setdefault("pred_offset", []).append(r4[..., :pred_reg_channel_num])
    output_dict.setdefault("pred_angle_cls", []).append(r4[..., pred_reg_channel_num:pred_reg_channel_num + ANGLE_CLS_NUM])
    output_dict.setdefault("pred_angle_res", []).append(r4[..., pred_reg_channel_num + ANGLE_CLS_NUM:])
    output_dict.setdefault("pred_reg_raw", []).append(pred_reg)
    return
