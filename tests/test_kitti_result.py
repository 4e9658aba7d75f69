"""Detections -> KITTI result lines (dataset/kitti_result.py): hand-evaluated corner / projection / formatting cases."""
import importlib
import os

import numpy as np
import pytest

R = importlib.import_module("3dssd_amd.dataset.kitti_result")


def test_box_corners_known_answers():
    # l = 4 along x, h = 2 up (negative y), w = 2 along z, bottom centre at (10, 1, 20), heading 0
    c = R.box3d_corners(np.array([[10.0, 1.0, 20.0]]), np.array([0.0]), np.array([[4.0, 2.0, 2.0]]))[0]
    assert c.shape == (8, 3)
    assert c[:4, 1].tolist() == [1, 1, 1, 1] and c[4:, 1].tolist() == [-1, -1, -1, -1]       # bottom face, then top face
    assert c[0].tolist() == [12, 1, 21] and c[1].tolist() == [12, 1, 19] and c[2].tolist() == [8, 1, 19] and c[3].tolist() == [8, 1, 21]
    # heading +90 degrees: x' = c x + s z, z' = -s x + c z  ->  (2, 0, 1) goes to (1, 0, -2)
    c90 = R.box3d_corners(np.array([[0.0, 0.0, 0.0]]), np.array([np.pi / 2]), np.array([[4.0, 2.0, 2.0]]))[0]
    np.testing.assert_allclose(c90[0], [1, 0, -2], atol=1e-12)
    np.testing.assert_allclose(np.sort(np.abs(c90[:, 0])), [1] * 8, atol=1e-12)                 # the long side now lies along z
    np.testing.assert_allclose(np.sort(np.abs(c90[:, 2])), [2] * 8, atol=1e-12)


def test_projection_and_clipping():
    P = np.array([[700.0, 0, 600, 0], [0, 700.0, 180, 0], [0, 0, 1, 0]])
    pts = np.array([[0.0, 0.0, 10.0], [1.0, -1.0, 10.0], [2.0, 1.0, 20.0]])
    uv = R.project_to_image(pts, P)
    np.testing.assert_allclose(uv, [[600, 180], [670, 110], [670, 215]], atol=1e-9)
    with_t = R.project_to_image(pts[:1], np.array([[700.0, 0, 600, 70], [0, 700.0, 180, -35], [0, 0, 1, 0.0]]))
    np.testing.assert_allclose(with_t, [[607, 176.5]], atol=1e-9)
    corners = R.box3d_corners(np.array([[0.0, 1.0, 10.0], [-30.0, 1.0, 5.0]]), np.array([0.0, 0.3]), np.array([[4.0, 2.0, 2.0], [4.0, 2.0, 2.0]]))
    b = R.project_to_image_space_corners(corners, P)
    assert b.dtype == np.float32 and b.shape == (2, 4)
    # first box: nearest face z = 9: x in [-2, 2] -> u = 600 -+ 700*2/9; y in [-1, 1] -> v = 180 -+ 700/9
    np.testing.assert_allclose(b[0], [600 - 1400 / 9, 180 - 700 / 9, 600 + 1400 / 9, 180 + 700 / 9], rtol=1e-6)
    assert b[1, 0] == 0 and b[1, 2] == 0 and 0 <= b[1, 1] <= b[1, 3] <= 375                    # far left of the image: clipped to x = 0
    with pytest.raises(ValueError):
        R.project_to_image_space_corners(np.zeros((3, 7, 3)), P)


def test_result_lines_and_file(tmp_path):
    P = np.array([[700.0, 0, 600, 0], [0, 700.0, 180, 0], [0, 0, 1, 0]])
    boxes = np.array([[0.0, 1.0, 10.0, 4.0, 2.0, 2.0, 0.0], [3.0, 1.5, 30.0, 3.9, 1.5, 1.6, -1.234], [1.0, 1.0, 15.0, 1.0, 1.0, 1.0, 0.5]], np.float32)
    scores = np.array([0.9, 0.123456789, 0.05], np.float32)
    cats = np.array([0, 0, 1])
    lines = R.kitti_result_lines(boxes, scores, cats, P, cls_list=("Car", "Pedestrian"), cls_thresh=0.1)
    assert len(lines) == 2                                                                      # the third is below the threshold
    assert lines[0] == "Car 0.00 0 -10 444.44 102.22 755.56 257.78 2.00 2.00 4.00 0.00 1.00 10.00 0.00 0.899999976"
    f = lines[1].split()
    assert f[0] == "Car" and f[1:4] == ["0.00", "0", "-10"] and len(f) == 16
    assert f[8:11] == ["1.50", "1.60", "3.90"] and f[11:15] == ["3.00", "1.50", "30.00", "-1.23"] and f[15] == "0.123456791"
    path = R.save_predictions(str(tmp_path / "kitti_result"), 42, boxes, scores, cats, P, cls_list=("Car", "Pedestrian"), cls_thresh=0.1)
    assert os.path.basename(path) == "000042.txt" and open(path).read().splitlines() == lines
    empty = R.save_predictions(str(tmp_path / "kitti_result"), 43, boxes, scores, cats, P, cls_thresh=0.95)
    assert open(empty).read() == ""
    assert R.kitti_result_lines(np.zeros((0, 7)), np.zeros(0), np.zeros(0), P) == []


def test_detections_of_frame_drops_padding():
    import torch
    out = {"pred_3d_bbox": [torch.arange(2 * 4 * 7, dtype=torch.float32).reshape(2, 4, 7)],
           "pred_3d_score": [torch.tensor([[0.9, 0.5, 0.0, 0.0], [0.8, 0.0, 0.0, 0.0]])],
           "pred_3d_cls_category": [torch.tensor([[0, 0, -1, -1], [0, -1, -1, -1]], dtype=torch.int32)]}
    b, s, c = R.detections_of_frame(out, 0)
    assert b.shape == (2, 7) and s.tolist() == pytest.approx([0.9, 0.5]) and c.tolist() == [0, 0]
    b, s, c = R.detections_of_frame(out, 1)
    assert b.shape == (1, 7) and b[0, 0] == 28.0
