"""Writes tests/golden/head_ref.npz: outputs of the REFERENCE'S OWN box-decoding code for the step after the SA
backbone, imported from /root/reference and executed under a numpy-backed stand-in for the
handful of `tensorflow` functions those files call (the way make_golden_kitti.py pins the input path):

    lib/utils/anchor_decoder.py:6-14,86-112   decode_class2angle, decode_dist_anchor_free          (TF graph code)
    lib/utils/box_3d_utils.py:25-58           box_3d_to_anchor      (has a numpy branch of its own: called on arrays)
    lib/utils/anchors_util.py:11-50           project_to_bev        (numpy branch)
    lib/builder/postprocessor.py:24-44        PostProcessor.class_unaware_format                    (TF graph code)
    lib/builder/postprocessor.py:49-123       PostProcessor.forward: the reg_i = min(i, cls_dim - 1) box selection, the
                                              per-class loop, gather / concat / category plumbing

What the stand-in is: every op is the obvious numpy call on float32 arrays (tf.maximum -> np.maximum, tf.one_hot ->
comparison with arange, tf.argmax -> np.argmax = first maximum like TF, python-float constants stay weak scalars as
TF converts them to the tensor's dtype), tensors are an ndarray subclass with get_shape().as_list().  It executes the
reference's statements in the reference's order: the arithmetic decisions (what is added to what, in which dtype) are
the reference's, only the elementwise kernels are numpy's instead of Eigen's.  The one op that cannot come from the
reference is tf.image.non_max_suppression (TensorFlow's C++ kernel): oracle/head_oracle.non_max_suppression, the
restatement of its published algorithm, is plugged in there and says so -- NMS stays "restated", the code AROUND it
is pinned.

Runs in the build container only (needs /root/reference):   python tests/golden/make_golden_head.py
tests/test_head.py (CPU: oracle vs fixture; GPU: csrc/head.hip vs fixture) reads the file."""
import os
import sys
import types

import numpy as np

REF = "/root/reference/lib"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


class T(np.ndarray):
    """ndarray with the two tf.Tensor methods the reference calls."""
    def get_shape(self):
        shp = list(self.shape)
        return types.SimpleNamespace(as_list=lambda: shp)


def t(a):
    return np.asarray(a).view(T)


def make_tf_stub(nms):
    tf = types.ModuleType("tensorflow")
    tf.Tensor = T
    tf.float32, tf.int32, tf.int64 = np.float32, np.int32, np.int64

    def cast(x, dt):
        return t(np.asarray(x).astype(dt))

    def one_hot(idx, depth, on_value=1, off_value=0, axis=-1):
        assert axis == -1
        idx = np.asarray(idx)
        return t(np.where(idx[..., None] == np.arange(depth), on_value, off_value))

    tf.cast = cast
    tf.one_hot = one_hot
    tf.argmax = lambda x, axis=None: t(np.argmax(np.asarray(x), axis=axis))
    tf.reduce_sum = lambda x, axis=None, keepdims=False: t(np.sum(np.asarray(x), axis=axis, keepdims=keepdims, dtype=np.asarray(x).dtype))
    tf.reduce_max = lambda x, axis=None, keepdims=False: t(np.max(np.asarray(x), axis=axis, keepdims=keepdims))
    tf.expand_dims = lambda x, axis: t(np.expand_dims(np.asarray(x), axis))
    tf.zeros_like = lambda x: t(np.zeros_like(np.asarray(x)))
    tf.ones_like = lambda x: t(np.ones_like(np.asarray(x)))
    tf.stack = lambda xs, axis=0: t(np.stack([np.asarray(x) for x in xs], axis=axis))
    tf.unstack = lambda x, axis=0: [t(a) for a in np.moveaxis(np.asarray(x), axis, 0)]
    tf.concat = lambda xs, axis=0: t(np.concatenate([np.asarray(x) for x in xs], axis=axis))
    tf.split = lambda x, n, axis=0: [t(a) for a in np.split(np.asarray(x), n, axis=axis)]
    tf.maximum = lambda a, b: t(np.maximum(a, b))
    tf.abs = lambda x: t(np.abs(np.asarray(x)))
    tf.cos = lambda x: t(np.cos(np.asarray(x)))
    tf.sin = lambda x: t(np.sin(np.asarray(x)))
    tf.round = lambda x: t(np.round(np.asarray(x)))
    tf.gather = lambda p, i: t(np.asarray(p)[np.asarray(i)])
    tf.image = types.SimpleNamespace(
        non_max_suppression=lambda boxes, scores, max_output_size, iou_threshold: t(nms(np.asarray(boxes), np.asarray(scores),
                                                                                      max_output_size, iou_threshold)))
    return tf


def attrdict(**kw):
    return types.SimpleNamespace(**kw)


def main():
    sys.path.insert(0, ROOT)
    from oracle import head_oracle as H
    tf = make_tf_stub(H.non_max_suppression)
    sys.modules["tensorflow"] = tf
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    cfg = attrdict(MODEL=attrdict(ANGLE_CLS_NUM=12,
                                  FIRST_STAGE=attrdict(MAX_OUTPUT_NUM=100, NMS_THRESH=0.1),
                                  SECOND_STAGE=attrdict(MAX_OUTPUT_NUM=100, NMS_THRESH=0.1)))
    core = types.ModuleType("core")
    core_config = types.ModuleType("core.config")
    core_config.cfg = cfg
    core.config = core_config
    sys.modules["core"], sys.modules["core.config"] = core, core_config
    sys.path.insert(0, REF)
    import utils.anchor_decoder as AD
    import utils.box_3d_utils as B3
    import utils.anchors_util as AU
    import builder.postprocessor as PP
    import dataset.maps_dict as maps_dict

    rng = np.random.default_rng(20260926)
    st = {}
    A = 12
    # ---------------- decode: random heads + the edge rows (angle-class ties, bin edges, lhw clamp at 0.1, negative halves)
    bs, n = 2, 300
    xyz = rng.uniform(-30, 60, (bs, n, 3)).astype(np.float32)
    dist6 = rng.normal(0, 1.5, (bs, n, 6)).astype(np.float32)
    acls = rng.normal(0, 1, (bs, n, A)).astype(np.float32)
    ares = rng.uniform(-0.6, 0.6, (bs, n, A)).astype(np.float32)
    dist6[0, 0, 3:6] = [0.05, 0.05, 0.05]          # 2 * 0.05 = 0.1: exactly the clamp value
    dist6[0, 1, 3:6] = [0.04, -1.0, 0.0499999]     # below the clamp / negative half-size
    dist6[0, 2, 3:6] = [0.05000001, 3.0, 40.0]
    acls[0, 3, :] = 0.25                            # all classes tie -> first maximum (class 0)
    acls[0, 4, :] = 0.0
    acls[0, 4, [5, 9]] = 1.0                        # two-way tie -> class 5
    acls[0, 5, :] = -np.arange(A)                   # class 0
    acls[0, 6, :] = np.arange(A)                    # class 11 (last bin)
    ares[0, 6, 11] = 0.5                            # (11 + 0.5) * pi/6: just below 2 pi
    ares[0, 5, 0] = -0.5                            # (0 - 0.5) * pi/6: negative angle
    ares[1, 0, :] = 0.0
    boxes = np.asarray(AD.decode_dist_anchor_free(t(xyz), t(dist6), t(acls), t(ares), False))
    ang = np.asarray(AD.decode_class2angle(t(np.argmax(acls, -1)), t(ares), A, 2 * np.pi / A))
    assert boxes.dtype == np.float32 and boxes.shape == (bs, n, 7)
    st.update(dec_xyz=xyz, dec_dist6=dist6, dec_acls=acls, dec_ares=ares, dec_boxes=boxes, dec_angle=ang)

    # ---------------- BEV: box_3d_to_anchor + project_to_bev on the decoded boxes and on axis-aligned / diagonal cases
    extra = np.array([[0, 0, 10, 4, 1.5, 2, 0.0], [0, 0, 10, 4, 1.5, 2, np.pi / 2], [0, 0, 10, 4, 1.5, 2, np.pi / 4],
                      [0, 0, 10, 4, 1.5, 2, -np.pi / 4], [5, 1, 20, 0.1, 0.1, 0.1, 3.0], [5, 1, 20, 3.9, 1.6, 1.6, 2 * np.pi],
                      [-7, 1, 33, 3.9, 1.6, 1.6, np.pi], [-7, 1, 33, 3.9, 1.6, 1.6, 1e-4]], np.float32)
    b7 = np.concatenate([boxes.reshape(-1, 7), extra], 0).astype(np.float32)
    anchors = B3.box_3d_to_anchor(b7)               # the reference's numpy branch, float32 in -> float32 out
    bev = AU.project_to_bev(anchors)
    assert anchors.dtype == np.float32 and bev.dtype == np.float32
    anchors64 = B3.box_3d_to_anchor(b7.astype(np.float64))      # the same statements in float64: the exact-rounding check
    bev64 = AU.project_to_bev(anchors64)
    st.update(bev_boxes=b7, bev_anchors=anchors, bev_out=bev, bev_out_f64=bev64)

    # ---------------- class_unaware_format (postprocessor.py:24-44): class-aware boxes [bs, n, cls, 7] + scores with ties
    cls = 3
    pb = rng.normal(0, 5, (2, 40, cls, 7)).astype(np.float32)
    ps = rng.uniform(0, 1, (2, 40, cls)).astype(np.float32)
    ps[0, 0] = [0.5, 0.5, 0.5]
    ps[0, 1] = [0.1, 0.7, 0.7]
    pp = PP.PostProcessor(0, 1)
    ub, us = pp.class_unaware_format(t(pb), t(ps))
    ub1, us1 = pp.class_unaware_format(t(pb[:, :, :1]), t(ps))
    st.update(cu_boxes=pb, cu_scores=ps, cu_out_boxes=np.asarray(ub), cu_out_scores=np.asarray(us),
              cu1_out_boxes=np.asarray(ub1), cu1_out_scores=np.asarray(us1))

    # ---------------- PostProcessor.forward plumbing (NMS = the restatement, see the module docstring)
    def run_forward(cls_num, boxes4, scores, tag):
        p = PP.PostProcessor(0, cls_num)
        out = {maps_dict.PRED_3D_BBOX: [], maps_dict.PRED_3D_SCORE: [], maps_dict.PRED_3D_CLS_CATEGORY: []}
        # the reference stacks per-frame results of DIFFERENT lengths only at batch size 1 (evaluator.py:145-147)
        p.forward(t(boxes4), t(scores), out)
        st[tag + "_boxes"], st[tag + "_scores"] = boxes4, scores
        st[tag + "_out_bbox"] = np.asarray(out[maps_dict.PRED_3D_BBOX][0])
        st[tag + "_out_score"] = np.asarray(out[maps_dict.PRED_3D_SCORE][0])
        st[tag + "_out_cat"] = np.asarray(out[maps_dict.PRED_3D_CLS_CATEGORY][0])

    def scene(m, c, aware):
        ctr = rng.uniform([-20, 0, 5], [20, 2, 60], (m, 3))
        k = c if aware else 1
        bx = np.zeros((1, m, k, 7), np.float32)
        for j in range(k):
            bx[0, :, j, :3] = ctr + rng.normal(0, 0.4, (m, 3))
            bx[0, :, j, 3:6] = rng.uniform([3.2, 1.4, 1.5], [4.5, 1.8, 1.9], (m, 3))
            bx[0, :, j, 6] = rng.uniform(-np.pi, np.pi, m)
        sc = rng.uniform(0, 1, (1, m, c)).astype(np.float32)
        sc[0, 3] = sc[0, 2]                          # equal scores
        bx[0, 5] = bx[0, 4]                          # identical boxes
        return bx, sc

    b1, s1 = scene(256, 1, False)
    run_forward(1, b1, s1, "pp1")                    # 3dssd.yaml: one class, class-agnostic boxes
    b3, s3 = scene(256, 3, False)
    run_forward(3, b3, s3, "pp3")                    # 3-class head, agnostic boxes: reg_i = 0 for every class
    b3a, s3a = scene(200, 3, True)
    run_forward(3, b3a, s3a, "pp3a")                 # class-aware boxes: class i uses box set i
    run_forward(1, b3a, s3a, "ppu")                  # cls_num (1) != score channels (3): class_unaware_format first
    np.savez_compressed(os.path.join(HERE, "head_ref.npz"), **st)
    print("wrote head_ref.npz:", {k: v.shape for k, v in st.items()})


if __name__ == "__main__":
    main()
