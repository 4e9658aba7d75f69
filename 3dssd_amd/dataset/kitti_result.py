"""Detections -> KITTI result files: the output side of the reference's tester
(lib/dataset/dataloader/kitti_dataloader.py:459-490 `save_predictions`, with lib/utils/box_3d_utils.py:62-87
`get_box3d_corners_helper_np`, lib/utils/anchors_util.py:54-91 `project_to_image_space_corners`,
lib/utils/kitti_util.py:329-349 `project_to_image`).  Host numpy like the reference: a few hundred boxes per frame.

Boxes are (x, bottom-centre y, z, l, h, w, ry) in rectified camera coordinates -- what the detector's postprocessor
returns; P is the 3x4 projection matrix of camera 2 from the frame's calibration file."""
import os

import numpy as np

KITTI_IMG_SHAPE = (375, 1242)          # anchors_util.py:54 default


def box3d_corners(centers, headings, sizes):
    """(N,3), (N,), (N,3)=(l,h,w) -> (N,8,3): corners of boxes standing on their bottom face (y from 0 to -h),
    turned about the y axis by [[c,0,s],[0,1,0],[-s,0,c]] and moved to the centres.   box_3d_utils.py:62-87"""
    centers, headings, sizes = np.asarray(centers), np.asarray(headings), np.asarray(sizes)
    l, h, w = sizes[:, 0], sizes[:, 1], sizes[:, 2]
    z = np.zeros_like(l)
    xc = np.stack([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2], 1)
    yc = np.stack([z, z, z, z, -h, -h, -h, -h], 1)
    zc = np.stack([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2], 1)
    c, s = np.cos(headings)[:, None], np.sin(headings)[:, None]
    x = c * xc + s * zc
    zz = -s * xc + c * zc
    return np.stack([x, yc, zz], -1) + centers[:, None, :]


def project_to_image(pts_3d, P):
    """(n,3) rect points through the 3x4 matrix P -> (n,2) pixel coordinates.   kitti_util.py:329-349"""
    pts_3d = np.asarray(pts_3d)
    hom = np.concatenate([pts_3d, np.ones((pts_3d.shape[0], 1), pts_3d.dtype)], 1)
    uvw = hom @ np.asarray(P).T
    return uvw[:, :2] / uvw[:, 2:3]


def project_to_image_space_corners(corners, P, img_shape=KITTI_IMG_SHAPE):
    """(N,8,3) box corners -> (N,4) float32 image boxes [x1, y1, x2, y2], clipped to the image.   anchors_util.py:54-91"""
    corners = np.asarray(corners)
    if corners.ndim != 3 or corners.shape[1] != 8:
        raise ValueError("Invalid shape for anchors %s, should be (N, 8, 3)" % (corners.shape,))
    pts = project_to_image(corners.reshape(-1, 3), P).reshape(-1, 8, 2)
    h, w = img_shape
    x1 = np.minimum(np.maximum(pts[:, :, 0].min(1), 0), w)
    y1 = np.minimum(np.maximum(pts[:, :, 1].min(1), 0), h)
    x2 = np.minimum(np.maximum(pts[:, :, 0].max(1), 0), w)
    y2 = np.minimum(np.maximum(pts[:, :, 1].max(1), 0), h)
    return np.stack([x1, y1, x2, y2], -1).astype(np.float32)


def kitti_result_lines(boxes_3d, scores, categories, P, cls_list=("Car",), cls_thresh=0.0, img_shape=KITTI_IMG_SHAPE):
    """One text line per detection with score >= cls_thresh, in the order given:
    `type 0.00 0 -10 x1 y1 x2 y2 h w l x y z ry score` (truncation / occlusion / alpha are not estimated: 0, 0, -10).
    kitti_dataloader.py:472-490"""
    boxes_3d = np.asarray(boxes_3d, np.float32).reshape(-1, 7)
    scores = np.asarray(scores, np.float32).reshape(-1)
    categories = np.asarray(categories).reshape(-1)
    keep = np.where(scores >= cls_thresh)[0]
    boxes_3d, scores, categories = boxes_3d[keep], scores[keep], categories[keep]
    if len(keep) == 0:
        return []
    corners = box3d_corners(boxes_3d[:, :3], boxes_3d[:, -1], boxes_3d[:, 3:-1])
    box2d = project_to_image_space_corners(corners, P, img_shape)
    lines = []
    for i in range(len(scores)):
        b = boxes_3d[i]
        lines.append("%s %0.2f %d %d " % (cls_list[int(categories[i])], 0.0, 0, -10) +
                     "%0.2f %0.2f %0.2f %0.2f " % tuple(float(v) for v in box2d[i]) +
                     "%0.2f %0.2f %0.2f " % (float(b[4]), float(b[5]), float(b[3])) +
                     "%0.2f %0.2f %0.2f %0.2f " % (float(b[0]), float(b[1]), float(b[2]), float(b[6])) +
                     "%0.9f" % float(scores[i]))
    return lines


def save_predictions(result_dir, sample_name, boxes_3d, scores, categories, P, cls_list=("Car",), cls_thresh=0.0):
    """Write `<result_dir>/<sample_name as %06d>.txt` (an empty file when nothing passes the threshold)."""
    os.makedirs(result_dir, exist_ok=True)
    path = os.path.join(result_dir, "%06d.txt" % int(sample_name))
    with open(path, "w") as f:
        for line in kitti_result_lines(boxes_3d, scores, categories, P, cls_list, cls_thresh):
            f.write(line + "\n")
    return path


def detections_of_frame(out, frame, index=0):
    """The fixed-size, padded outputs of PostProcessor.forward (builder/postprocessor.py: `pred_3d_bbox` [bs, cls*K, 7],
    `pred_3d_score`, `pred_3d_cls_category` with -1 on the padding rows) -> the (boxes, scores, categories) of one
    frame as numpy arrays without the padding, in the postprocessor's order (class-major, NMS selection order)."""
    boxes = out["pred_3d_bbox"][index][frame].detach().cpu().numpy()
    scores = out["pred_3d_score"][index][frame].detach().cpu().numpy()
    cats = out["pred_3d_cls_category"][index][frame].detach().cpu().numpy()
    keep = cats >= 0
    return boxes[keep], scores[keep], cats[keep]
