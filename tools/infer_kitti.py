"""Velodyne sweeps in, KITTI result files out -- the inference loop of the reference's tester
(lib/core/evaluator.py + kitti_dataloader.py:459-490) on the MI355X path:

    python tools/infer_kitti.py --velodyne DIR --calib DIR --out DIR [--checkpoint PREFIX_OR_DIR] [--batch 8]

Every `NNNNNN.bin` of --velodyne is cropped / resampled to 16 384 points (dataset/kitti_input.py), run through
SingleStageDetector (backbone + head + decode + NMS) and written as `NNNNNN.txt` (dataset/kitti_result.py).
Weights: a TensorFlow-1 checkpoint of the reference (read without TensorFlow, utils/tf_checkpoint.py), or -- with no
--checkpoint -- seeded random weights of the same architecture (plumbing runs only; the boxes mean nothing)."""
import argparse
import glob
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pkg(name):
    return importlib.import_module("3dssd_amd." + name)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--velodyne", required=True)
    ap.add_argument("--calib", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--cls-thresh", type=float, default=0.3)
    ap.add_argument("--image-shape", type=int, nargs=2, default=[375, 1242], metavar=("H", "W"))
    args = ap.parse_args(argv)

    cfgs, syn, K, Rz = pkg("configs"), pkg("synthetic"), pkg("dataset.kitti_input"), pkg("dataset.kitti_result")
    dev = torch.device("cuda:0")
    if args.checkpoint:
        tfc = pkg("utils.tf_checkpoint")
        prefix = tfc.latest_checkpoint(args.checkpoint) if os.path.isdir(args.checkpoint) else args.checkpoint
        params = tfc.load_checkpoint(prefix)
    else:
        params = syn.random_backbone_params(cfgs.KITTI_3DSSD_ARCH)
        syn.random_head_params(512, 1, cfgs.KITTI_ANGLE_CLS_NUM, params=params)
    det = pkg("modeling.single_stage_detector").SingleStageDetector(
        cfgs.KITTI_3DSSD_ARCH, cfgs.KITTI_3DSSD_HEAD, params, dev, cls_num=1, angle_cls_num=cfgs.KITTI_ANGLE_CLS_NUM,
        max_output_size=cfgs.KITTI_MAX_OUTPUT_NUM, nms_threshold=cfgs.KITTI_NMS_THRESH)

    files = sorted(glob.glob(os.path.join(args.velodyne, "*.bin")))
    rng = np.random.default_rng(0)
    written = []
    for i in range(0, len(files), args.batch):
        chunk = files[i:i + args.batch]
        frames, calibs, names = [], [], []
        for f in chunk:
            name = os.path.splitext(os.path.basename(f))[0]
            calib = K.Calibration(os.path.join(args.calib, name + ".txt"))
            frames.append(K.prepare_frame(K.load_velo_scan(f), calib, tuple(args.image_shape), rng=rng))
            calibs.append(calib)
            names.append(name)
        out = det(torch.from_numpy(np.stack(frames)).to(dev))
        torch.cuda.synchronize()
        for j, (name, calib) in enumerate(zip(names, calibs)):
            boxes, scores, cats = Rz.detections_of_frame(out, j)
            written.append(Rz.save_predictions(args.out, int(name), boxes, scores, cats, calib.P, cls_list=("Car",),
                                               cls_thresh=args.cls_thresh))
    print("wrote %d result files to %s" % (len(written), args.out))
    return written


if __name__ == "__main__":
    main()
