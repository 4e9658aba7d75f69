"""Generates the committed golden fixtures in tests/golden/*.npz from the CPU oracle on seeded inputs.

The reference cannot be imported or built here (TensorFlow 1.4 + CUDA) and ships no
vectors of its own, so these are REGRESSION pins of the oracle (whose semantics are pinned by the
hand-derived KATs in tests/test_oracle_kat.py), and the vectors the HIP path is compared against on the
GPU box, where the oracle is also re-run.  Run:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import sa_oracle as O  # noqa: E402


def ops_case():
    rng = np.random.default_rng(20260925)
    b, n, m = 2, 1500, 96
    xyz = rng.uniform(-6, 6, (b, n, 3)).astype(np.float32)
    xyz[:, n - 150:] = xyz[:, :150]                      # duplicated points: ties
    feat = rng.normal(0, 1, (b, n, 5)).astype(np.float32)
    fps = O.farthest_point_sample(m, xyz)
    ffeat = np.concatenate([xyz, feat], -1)
    d = O.calc_square_dist(ffeat[:, :400], ffeat[:, :400])
    ffps = O.farthest_point_sample_with_distance(64, d)
    ctr = O.gather_point(xyz, fps)
    qi, qc = O.query_ball_point(1.2, 16, xyz, ctr)
    di, dc = O.query_ball_point_dilated(1.2, 2.4, 32, xyz, ctr)
    grp = O.group_point(feat, qi)
    return dict(xyz=xyz, feat=feat, fps=fps, dist_row0=d[:, 0], dist_sum=np.float64(d.astype(np.float64).sum()),
                ffps=ffps, ctr=ctr, qi=qi, qc=qc, di=di, dc=dc, grp_sum=grp.astype(np.float64).sum(axis=(2, 3)))


def sa_layer_case():
    cfgs = importlib.import_module("3dssd_amd.configs")
    syn = importlib.import_module("3dssd_amd.synthetic")
    arch = [[[0], [0], [0.5, 1.0], [16, 32], [[16, 16, 32], [16, 32, 48]], True,
             [-1], ["FS"], [128], -1, False, "SA_Layer", "layer1", True, -1, 64]]
    params = syn.random_backbone_params(arch)
    pts = syn.kitti_like_batch(2, n=2048, first_frame=42)
    row = arch[0]
    nx, nf, idx = O.pointnet_sa_module_msg(pts[:, :, :3], pts[:, :, 3:] * 40.0, row[2], row[3], row[4], row[5],
                                           row[6], row[7], row[8], None, row[12], row[13], params,
                                           aggregation_channel=row[15])
    return dict(frame_ids=np.array([42, 43]), fps_idx=idx, new_xyz=nx, new_points=nf.astype(np.float32))


def backbone_case():
    cfgs = importlib.import_module("3dssd_amd.configs")
    syn = importlib.import_module("3dssd_amd.synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    params = syn.random_backbone_params(arch)
    pts = syn.kitti_like_batch(1, first_frame=3)
    xl, fl, il = O.sa_backbone(pts, arch, params, cfgs.KITTI_MAX_TRANSLATE_RANGE)
    return dict(frame_id=np.array([3]), fps1=il[1][:, :256], fps2=il[2], fps3=il[3], fps4=il[4],
                out_xyz=xl[-1], out_feat=fl[-1].astype(np.float32))


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **ops_case())
    np.savez_compressed(os.path.join(HERE, "sa_layer_fs.npz"), **sa_layer_case())
    np.savez_compressed(os.path.join(HERE, "kitti_backbone_frame3.npz"), **backbone_case())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
