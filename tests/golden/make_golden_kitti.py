"""Writes tests/golden/kitti_input_ref.npz: outputs of the REFERENCE'S OWN numpy functions for the KITTI input path
(SURVEY.md 8f rank 3), imported from /root/reference under stub `tensorflow` / `cv2` modules (both are only imported
at module level there, never used by these functions):
    lib/utils/kitti_util.py      Calibration (read_calib_file, project_velo_to_rect, project_rect_to_image), load_velo_scan
    lib/utils/points_filter.py   get_point_filter, get_point_filter_in_image
and of the resampling statements of lib/dataset/dataloader/kitti_dataloader.py:137-151, whose text is extracted from
the file and executed here (the enclosing class cannot be imported: core.config, data providers, ...).
Runs in the build container only (needs /root/reference):   python tests/golden/make_golden_kitti.py
tests/test_kitti_input.py compares 3dssd_amd/dataset/kitti_input.py with the file."""
import hashlib
import os
import sys
import tempfile
import textwrap
import types

import numpy as np

REF = "/root/reference/lib"
HERE = os.path.dirname(os.path.abspath(__file__))

CALIB_TXT = """P0: 7.070493e+02 0.000000e+00 6.040814e+02 0.000000e+00 0.000000e+00 7.070493e+02 1.805066e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P1: 7.070493e+02 0.000000e+00 6.040814e+02 -3.797842e+02 0.000000e+00 7.070493e+02 1.805066e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P2: 7.070493e+02 0.000000e+00 6.040814e+02 4.575831e+01 0.000000e+00 7.070493e+02 1.805066e+02 -3.454157e-01 0.000000e+00 0.000000e+00 1.000000e+00 4.981016e-03
P3: 7.070493e+02 0.000000e+00 6.040814e+02 -3.341081e+02 0.000000e+00 7.070493e+02 1.805066e+02 2.330660e+00 0.000000e+00 0.000000e+00 1.000000e+00 3.201153e-03
R0_rect: 9.999128e-01 1.009263e-02 -8.511932e-03 -1.012729e-02 9.999406e-01 -4.037671e-03 8.470675e-03 4.123522e-03 9.999556e-01
Tr_velo_to_cam: 6.927964e-03 -9.999722e-01 -2.757829e-03 -2.457729e-02 -1.162982e-03 2.749836e-03 -9.999955e-01 -6.127237e-02 9.999753e-01 6.931141e-03 -1.143899e-03 -3.321029e-01
Tr_imu_to_velo: 9.999976e-01 7.553071e-04 -2.035826e-03 -8.086759e-01 -7.854027e-04 9.998898e-01 -1.482298e-02 3.195559e-01 2.024406e-03 1.482454e-02 9.998881e-01 -7.997231e-01
"""


def synthetic_scan(seed, n):
    """velodyne-frame points: forward x in [0, 80], y in [-40, 40], z in [-2.5, 1], intensity in [0, 1]; a few exactly on
    the crop boundaries are appended by the caller"""
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(0, 80, n), rng.uniform(-40, 40, n), rng.uniform(-2.5, 1.0, n), rng.uniform(0, 1, n)],
                    -1).astype(np.float32)


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha1(a.tobytes()).hexdigest()


def reference_resample_lines():
    src = open(os.path.join(REF, "dataset/dataloader/kitti_dataloader.py")).read().splitlines()
    a = next(i for i, l in enumerate(src) if "# randomly choose points" in l)
    b = next(i for i, l in enumerate(src) if i > a and "sampled_idx = np.concatenate" in l)
    return textwrap.dedent("\n".join(src[a:b + 1]))


def main():
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import utils.kitti_util as ku
    import utils.points_filter as pf

    store = {"calib_txt": np.array(CALIB_TXT)}
    with tempfile.TemporaryDirectory() as d:
        cpath = os.path.join(d, "000000.txt")
        open(cpath, "w").write(CALIB_TXT)
        calib = ku.Calibration(cpath)
        scan = synthetic_scan(7, 30000)
        spath = os.path.join(d, "000000.bin")
        scan.tofile(spath)
        assert np.array_equal(ku.load_velo_scan(spath), scan)
    store["P"], store["V2C"], store["R0"] = calib.P, calib.V2C, calib.R0
    pts = scan[:, :3]
    rect = calib.project_velo_to_rect(pts)                                  # kitti_dataloader.py:179
    uv = calib.project_rect_to_image(rect)
    extents = np.reshape([-40.0, 40.0, -5.0, 3.0, 0.0, 70.0], [3, 2])       # kitti_dataloader.py:83-84, 3dssd.yaml:3
    h, w = 370, 1224
    m_img = pf.get_point_filter_in_image(rect, calib, h, w)                 # :181
    m_ext = pf.get_point_filter(rect, extents)                              # :182
    keep = np.where(np.logical_and(m_img, m_ext))[0]                        # :183-184
    store.update(scan_seed=7, scan_n=30000, scan_sha1=np.array(sha(scan)), image_shape=np.array([h, w]),
                 rect_head=rect[:256], uv_head=uv[:256], rect_sha1=np.array(sha(rect)), uv_sha1=np.array(sha(uv)),
                 mask_image=np.packbits(m_img), mask_extents=np.packbits(m_ext), keep=keep.astype(np.int32))
    # boundary semantics: strict extents, 0 <= u < w, z >= 0
    edge = np.array([[-40.0, 0.0, 10.0], [39.99999, 0.0, 10.0], [0.0, 3.0, 10.0], [0.0, -5.0, 10.0], [0.0, 0.0, 0.0],
                     [0.0, 0.0, 70.0], [0.0, 1.0, 69.99999], [0.0, 1.0, 1e-9]])
    store["edge_pts"] = edge
    store["edge_extents"] = pf.get_point_filter(edge, extents)
    store["edge_image"] = pf.get_point_filter_in_image(edge, calib, h, w)
    # resampling (kitti_dataloader.py:137-151), executed from the reference's own text with the global numpy RNG
    code = reference_resample_lines()
    cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(POINTS_NUM_FOR_TRAINING=16384))
    for name, n_pts, seed in (("many", 20000, 11), ("few", 5000, 12), ("exact", 16384, 13)):
        ns = {"np": np, "cfg": cfg, "points": np.zeros((n_pts, 4))}
        np.random.seed(seed)
        exec(code, ns)
        store["resample_%s_n" % name] = n_pts
        store["resample_%s_seed" % name] = seed
        store["resample_%s_idx" % name] = ns["sampled_idx"].astype(np.int32)
    out = os.path.join(HERE, "kitti_input_ref.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes; kept", len(keep), "of", len(scan))


if __name__ == "__main__":
    main()
