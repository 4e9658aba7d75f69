"""KITTI velodyne scan -> the [16384, 4] frame the SA backbone consumes.

Host-side numpy, mirroring what the reference does before the first SA layer:
  lib/utils/kitti_util.py:323-326     load_velo_scan        (.bin = float32 [N, 4]: x, y, z, intensity)
  lib/utils/kitti_util.py:80-118      Calibration           (P2, Tr_velo_to_cam, R0_rect from the calib .txt)
  lib/utils/kitti_util.py:169-199     project_velo_to_rect  (cart2hom @ V2C^T, then R0 @ .)
  lib/utils/kitti_util.py:204-214     project_rect_to_image (cart2hom @ P^T, divide by depth)
  lib/utils/points_filter.py:8-52     get_point_filter (strict inequalities), get_point_filter_in_image
                                      (0 <= u < width, 0 <= v < height, z >= 0)
  lib/dataset/dataloader/kitti_dataloader.py:83-84,186-196,137-151
                                      extents = POINT_CLOUD_RANGE reshaped [3, 2]; crop; random choice of
                                      POINTS_NUM_FOR_TRAINING points (without replacement when there are
                                      enough, else every point once plus draws with replacement)
Arithmetic stays in float64 until the end like the reference (np.dot on float64 calibration matrices); the
returned frame is float32.  The random draw takes a numpy Generator so that callers can make it reproducible
(the reference uses the global np.random state).
"""
import numpy as np

KITTI_POINT_CLOUD_RANGE = (-40.0, 40.0, -5.0, 3.0, 0.0, 70.0)    # configs/kitti/3dssd/3dssd.yaml:3
KITTI_POINTS_NUM = 16384                                          # MODEL.POINTS_NUM_FOR_TRAINING


def load_velo_scan(path):
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)


def read_calib_file(path):
    data = {}
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip()
            if not line:
                continue
            key, value = line.split(":", 1)
            try:
                data[key] = np.array([float(x) for x in value.split()])
            except ValueError:
                pass
    return data


class Calibration:
    """P2 [3,4], Tr_velo_to_cam [3,4], R0_rect [3,3] of one KITTI frame (dict as read_calib_file returns, or a path)."""

    def __init__(self, calibs):
        if isinstance(calibs, str):
            calibs = read_calib_file(calibs)
        self.P = np.reshape(calibs["P2"], [3, 4])
        self.V2C = np.reshape(calibs["Tr_velo_to_cam"], [3, 4])
        self.R0 = np.reshape(calibs["R0_rect"], [3, 3])

    @staticmethod
    def cart2hom(p):
        return np.hstack((p, np.ones((p.shape[0], 1))))

    def project_velo_to_rect(self, pts_velo):
        ref = np.dot(self.cart2hom(pts_velo), self.V2C.T)
        return np.dot(self.R0, ref.T).T

    def project_rect_to_image(self, pts_rect):
        uvw = np.dot(self.cart2hom(pts_rect), self.P.T)
        uvw[:, 0] /= uvw[:, 2]
        uvw[:, 1] /= uvw[:, 2]
        return uvw[:, 0:2]


def point_filter_extents(pts_rect, extents):
    e = np.reshape(np.asarray(extents, np.float64), [3, 2])
    p = np.asarray(pts_rect)
    return ((p[:, 0] > e[0, 0]) & (p[:, 0] < e[0, 1]) & (p[:, 1] > e[1, 0]) & (p[:, 1] < e[1, 1]) &
            (p[:, 2] > e[2, 0]) & (p[:, 2] < e[2, 1]))


def point_filter_in_image(pts_rect, calib, height, width):
    with np.errstate(divide="ignore", invalid="ignore"):
        uv = calib.project_rect_to_image(pts_rect)
        inside = (uv[:, 0] >= 0) & (uv[:, 0] < width) & (uv[:, 1] >= 0) & (uv[:, 1] < height)
    return np.logical_and(inside, pts_rect[:, 2] >= 0)


def crop_frame(scan, calib, image_shape, extents=KITTI_POINT_CLOUD_RANGE):
    """scan [N,4] velodyne -> [K,4] (rect x, y, z, intensity) of the points inside the image and the range."""
    pts = calib.project_velo_to_rect(scan[:, :3])
    keep = np.where(np.logical_and(point_filter_in_image(pts, calib, image_shape[0], image_shape[1]),
                                   point_filter_extents(pts, extents)))[0]
    return np.concatenate([pts[keep], scan[keep, 3:4].astype(np.float64)], axis=-1)


def resample(points, num=KITTI_POINTS_NUM, rng=None):
    """Exactly `num` rows: a random subset without replacement when there are enough points, else every point once
    (shuffled) followed by draws with replacement (kitti_dataloader.py:137-151)."""
    rng = np.random.default_rng() if rng is None else rng
    n = points.shape[0]
    if n == 0:
        raise ValueError("no point survives the crop")
    idx = np.arange(n)
    if n >= num:
        pick = rng.choice(idx, num, replace=False)
    else:
        pick = np.concatenate([rng.choice(idx, n, replace=False), rng.choice(idx, num - n, replace=True)])
    return points[pick]


def prepare_frame(scan, calib, image_shape, num=KITTI_POINTS_NUM, extents=KITTI_POINT_CLOUD_RANGE, rng=None):
    """velodyne scan [N,4] -> float32 [num,4] network input (rect coordinates + intensity)."""
    return np.ascontiguousarray(resample(crop_frame(scan, calib, image_shape, extents), num, rng), np.float32)
