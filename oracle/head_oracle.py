"""CPU oracle for the step after the SA backbone: detection head,
anchor-free box decoding, sigmoid scores and per-class BEV NMS.

TEST INFRASTRUCTURE ONLY (see oracle/sa_oracle.py).  PARITY UNPINNED: the reference has no tests or vectors for
this step either; the hand-derived KATs in tests/test_head.py pin it.

Follows
  lib/modeling/head_builder.py:81-113, lib/utils/head_util.py:26-59        (head)
  lib/utils/anchor_decoder.py:6-14,86-112                                    (decode_class2angle, decode_dist_anchor_free)
  lib/modeling/single_stage_detector.py:195-228                              (test_forward, sigmoid)
  lib/utils/box_3d_utils.py:25-58, lib/utils/anchors_util.py:11-50           (box_3d_to_anchor, project_to_bev)
  lib/builder/postprocessor.py:49-123                                        (per-class tf.image.non_max_suppression)

Third-party arithmetic: tf.image.non_max_suppression (TensorFlow 1.4.0, README.md:28; kernel
tensorflow/core/kernels/non_max_suppression_op.cc, not vendored) -- restated from its published algorithm:
candidates in decreasing score order, a candidate is kept iff its IoU with every already kept box is <= the
threshold, at most max_output_size kept; IoU on corner-normalised boxes, 0 when either area is <= 0.
Pinned here (TF leaves them open): equal scores are ordered by ascending index; |cos ry|, |sin ry| and the sigmoid
are evaluated in float64 and rounded to float32 (float32 libm results differ between hosts and GPUs in the last
ulp); every other operation is a single float32 operation in the order the reference's Python writes it.
"""
import numpy as np

from . import sa_oracle as O

f32 = np.float32


def box_regression_head(features, params, cls_channel, angle_cls_num=12, scope="", bn=True, mlp_list=(128,)):
    """head_builder.py:97-108 + head_util.py:26-59.  features [b,n,c] -> (pred_cls [b,n,cls],
    pred_offset [b,n,6], pred_angle_cls [b,n,A], pred_angle_res [b,n,A]) for the anchor-free head (base num 1)."""
    pre = scope + "/" if scope else ""
    x = features
    for i, _ch in enumerate(mlp_list):
        w, b = O.fold_conv_bn(params, pre + "conv1d_%d" % i, bn)
        x = O.dense(x, w, b, relu=True)
    w, b = O.fold_conv_bn(params, pre + "pred_cls_base", bn)
    c = O.dense(x, w, b, relu=True)
    w, b = O.fold_conv_bn(params, pre + "pred_cls", False)
    pred_cls = O.dense(c, w, b, relu=False)
    assert pred_cls.shape[-1] == cls_channel
    w, b = O.fold_conv_bn(params, pre + "pred_reg_base", bn)
    r = O.dense(x, w, b, relu=True)
    w, b = O.fold_conv_bn(params, pre + "pred_reg", False)
    pred_reg = O.dense(r, w, b, relu=False)                      # [b,n,6+2A]
    A = angle_cls_num
    assert pred_reg.shape[-1] == 6 + 2 * A
    return pred_cls, pred_reg[..., :6], pred_reg[..., 6:6 + A], pred_reg[..., 6 + A:]


def decode_class2angle(pred_cls, pred_res_norm, bin_size, bin_interval, bin_offset=0.0):
    """anchor_decoder.py:6-14.  pred_cls int [...], pred_res_norm [..., bin_size]."""
    res = np.take_along_axis(pred_res_norm, pred_cls[..., None], -1)[..., 0].astype(f32)
    return ((pred_cls.astype(f32) + res + f32(bin_offset)) * f32(bin_interval)).astype(f32)


def decode_dist_anchor_free(center_xyz, det_forced_6_distance, det_angle_cls, det_angle_res, angle_cls_num=12):
    """anchor_decoder.py:86-112 -> boxes [b,n,7] = [cx, cy(bottom), cz, l, h, w, ry]."""
    cls = np.argmax(det_angle_cls, -1)                           # first maximum, like tf.argmax
    ang = decode_class2angle(cls, det_angle_res, angle_cls_num, 2 * np.pi / angle_cls_num)
    t = det_forced_6_distance[..., :3].astype(f32)
    half = det_forced_6_distance[..., 3:6].astype(f32)
    ctr = (center_xyz.astype(f32) + t).astype(f32)
    ctr[..., 1] = (ctr[..., 1] + half[..., 1]).astype(f32)      # + (0, half_y, 0), :104-107
    lhw = np.maximum(half * f32(2.0), f32(0.1)).astype(f32)
    return np.concatenate([ctr, lhw, ang[..., None]], -1).astype(f32)


def sigmoid_f32(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)


def box_3d_to_bev(boxes):
    """box_3d_utils.py:25-58 (ortho_rotate=False) then anchors_util.py:11-50 -> [x_min, z_min, x_max, z_max]."""
    x, z = boxes[..., 0].astype(f32), boxes[..., 2].astype(f32)
    l, w, ry = boxes[..., 3].astype(f32), boxes[..., 5].astype(f32), boxes[..., 6]
    c = np.abs(np.cos(ry.astype(np.float64))).astype(f32)
    s = np.abs(np.sin(ry.astype(np.float64))).astype(f32)
    dimx = ((l * c).astype(f32) + (w * s).astype(f32)).astype(f32)
    dimz = ((w * c).astype(f32) + (l * s).astype(f32)).astype(f32)
    hx, hz = (dimx / f32(2.0)).astype(f32), (dimz / f32(2.0)).astype(f32)
    return np.stack([x - hx, z - hz, x + hx, z + hz], -1).astype(f32)


def _iou(a, b):
    ymin_i, xmin_i = min(a[0], a[2]), min(a[1], a[3])
    ymax_i, xmax_i = max(a[0], a[2]), max(a[1], a[3])
    ymin_j, xmin_j = min(b[0], b[2]), min(b[1], b[3])
    ymax_j, xmax_j = max(b[0], b[2]), max(b[1], b[3])
    area_i = f32(f32(ymax_i - ymin_i) * f32(xmax_i - xmin_i))
    area_j = f32(f32(ymax_j - ymin_j) * f32(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return f32(0.0)
    iy = max(f32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), f32(0.0))
    ix = max(f32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), f32(0.0))
    inter = f32(iy * ix)
    return f32(inter / f32(f32(area_i + area_j) - inter))


def non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """tf.image.non_max_suppression (TF 1.4) restated.  boxes [n,4] f32, scores [n] f32 -> kept indices."""
    boxes = np.asarray(boxes, f32)
    scores = np.asarray(scores, f32)
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    thr = f32(iou_threshold)
    keep = []
    for i in order:
        if len(keep) >= max_output_size:
            break
        ok = True
        for j in reversed(keep):
            if _iou(boxes[i], boxes[j]) > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return np.asarray(keep, np.int32)


def postprocess(pred_boxes, pred_score, max_output_size=100, nms_threshold=0.1):
    """postprocessor.py:49-123 for class-agnostic boxes [b,n,7] and scores [b,n,cls]: per frame and class the kept
    indices (padded with -1 to max_output_size) and their count."""
    b, n, cls = pred_score.shape
    bev = box_3d_to_bev(pred_boxes)
    idx = np.full((b, cls, max_output_size), -1, np.int32)
    cnt = np.zeros((b, cls), np.int32)
    for bi in range(b):
        for c in range(cls):
            k = non_max_suppression(bev[bi], pred_score[bi, :, c], max_output_size, nms_threshold)
            idx[bi, c, :len(k)] = k
            cnt[bi, c] = len(k)
    return idx, cnt, bev


def detect(xyz, features, params, cls_channel=1, angle_cls_num=12, max_output_size=100, nms_threshold=0.1):
    """backbone output -> (boxes [b,n,7], scores [b,n,cls], nms idx [b,cls,max_out], cnt [b,cls])."""
    pc, po, pac, par = box_regression_head(features, params, cls_channel, angle_cls_num)
    boxes = decode_dist_anchor_free(xyz, po, pac, par, angle_cls_num)
    scores = sigmoid_f32(pc)
    idx, cnt, _ = postprocess(boxes, scores, max_output_size, nms_threshold)
    return boxes, scores, idx, cnt
