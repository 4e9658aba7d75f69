"""Stand-alone operator micro-benchmark at the shapes of configs/kitti/3dssd/3dssd.yaml (batch 8): achieved
GB/s of the HBM-bound ops (group_point, gather_point, ball query) against the 8 TB/s roofline, and the API-level
ops the fused backbone does not call (group_point).  One JSON line per case."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

G = importlib.import_module("3dssd_amd.utils.tf_ops.grouping.tf_grouping")
S = importlib.import_module("3dssd_amd.utils.tf_ops.sampling.tf_sampling")
syn = importlib.import_module("3dssd_amd.synthetic")
dev = torch.device("cuda:0")
HBM = 8000.0


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


B = 8
pts = torch.from_numpy(syn.kitti_like_batch(B)).to(dev)
xyz = pts[:, :, :3].contiguous()
out = []
# (n, m, ns, C) of every scale of the backbone: group_point on features and on xyz, materialised like the reference
for name, n, m, ns, C in [("L1.s2", 16384, 4096, 64, 1), ("L2.s0", 4096, 1024, 32, 64), ("L2.s2", 4096, 1024, 64, 64),
                          ("L3.s0", 1024, 512, 32, 128), ("L4.s1", 512, 256, 32, 256)]:
    feat = torch.randn(B, n, C, device=dev)
    x = torch.randn(B, n, 3, device=dev)
    idx = torch.randint(0, n, (B, m, ns), device=dev, dtype=torch.int32)
    for label, src, c in (("features", feat, C), ("xyz", x, 3)):
        ms = timeit(lambda: G.group_point(src, idx))
        by = B * (m * ns * c * 4 + m * ns * 4 + n * c * 4)          # output once + idx + source once
        out.append(dict(op="group_point", case=name, tensor=label, shape=[B, m, ns, c], ms=round(ms, 4),
                        mbytes=round(by / 1e6, 2), gbs=round(by / ms / 1e6, 1), hbm_frac=round(by / ms / 1e6 / HBM, 4)))
# large materialisation (nuScenes-scale stress, configs[4]): 65536 points
n, m, ns, C = 65536, 4096, 64, 64
feat = torch.randn(B, n, C, device=dev)
idx = torch.randint(0, n, (B, m, ns), device=dev, dtype=torch.int32)
ms = timeit(lambda: G.group_point(feat, idx), iters=10)
by = B * (m * ns * C * 4 + m * ns * 4 + n * C * 4)
out.append(dict(op="group_point", case="stress 65536", tensor="features", shape=[B, m, ns, C], ms=round(ms, 4),
                mbytes=round(by / 1e6, 2), gbs=round(by / ms / 1e6, 1), hbm_frac=round(by / ms / 1e6 / HBM, 4)))
# ball query (all three bands of layer 1, through the reference API one band at a time, and fused)
fidx = S.farthest_point_sample(4096, xyz)
ctr = S.gather_point(xyz, fidx)
ms3 = timeit(lambda: [G.query_ball_point_dilated(a, b_, s, xyz, ctr) for a, b_, s in ((0.0, 0.2, 32), (0.2, 0.4, 32), (0.4, 0.8, 64))], iters=10)
by = B * (16384 * 12 + 4096 * 12) * 3 + B * 4096 * (32 + 32 + 64 + 3) * 4
out.append(dict(op="query_ball_point_dilated x3 (API, one band per call)", case="L1", ms=round(ms3, 4), mbytes=round(by / 1e6, 2),
                gbs=round(by / ms3 / 1e6, 1), hbm_frac=round(by / ms3 / 1e6 / HBM, 5), pair_evals=B * 3 * 16384 * 4096))
ms = timeit(lambda: S.gather_point(torch.randn(1, 1, 1, device=dev).expand(B, 512, 256).contiguous(), fidx[:, :256] % 512), iters=5)
# BASELINE.json configs[2]: F-FPS isolated, 16384 -> 4096 on 3 + 64 channels, batch 32, through the reference API
# (farthest_point_sample on a c-channel tensor -- the matrix form of the backbone would need a 1.07 GB matrix per
# frame): fps_coop.hip, 16 cooperating workgroups per frame, 16 frames per launch.
pf = torch.randn(32, 16384, 67, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
S.farthest_point_sample(8, pf); torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record(); S.farthest_point_sample(4096, pf); t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1)
out.append(dict(op="farthest_point_sample c=67 (configs[2])", case="32 x 16384 -> 4096", ms=round(ms, 2),
                frames_per_s=round(32 / ms * 1e3, 1), pair_evals=32 * 16384 * 4095, note="fps_coop.hip: 16 workgroups per frame, points in registers (single-workgroup tiled kernel: 1117 ms)"))
# BASELINE.json configs[4]: 65536-point frames, batch 16: layer-1 D-FPS (cooperative kernel) and ball query
p64 = torch.from_numpy(syn.kitti_like_batch(16, n=65536)).to(dev)[:, :, :3].contiguous()
S.farthest_point_sample(8, p64); torch.cuda.synchronize()
t0.record(); f64 = S.farthest_point_sample(4096, p64); t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1)
out.append(dict(op="farthest_point_sample c=3 (configs[4])", case="16 x 65536 -> 4096", ms=round(ms, 2),
                frames_per_s=round(16 / ms * 1e3, 1), note="fps_coop.hip: 16 workgroups per frame (single-workgroup global-scratch kernel: 135 ms)"))
c64 = S.gather_point(p64, f64)
ms = timeit(lambda: G.query_ball_point_dilated(0.4, 0.8, 64, p64, c64), iters=5)
by = 16 * (65536 * 12 + 4096 * 12 + 4096 * 65 * 4)
out.append(dict(op="query_ball_point_dilated (configs[4])", case="16 x 65536, m=4096, r 0.4-0.8, ns 64", ms=round(ms, 4),
                mbytes=round(by / 1e6, 2), gbs=round(by / ms / 1e6, 1)))
# BASELINE.json configs[4], whole backbone: 16 frames of 65536 points through the same ARCHITECTURE rows (SURVEY.md 8d
# Config 5: only layer 1 sees the larger frame), eager launches on one stream (the cooperative FPS is not graph-captured)
cfgs = importlib.import_module("3dssd_amd.configs")
net = importlib.import_module("3dssd_amd.backbone").SABackbone(cfgs.KITTI_3DSSD_ARCH, syn.random_backbone_params(cfgs.KITTI_3DSSD_ARCH),
                                                             dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)
f64 = torch.from_numpy(syn.kitti_like_batch(16, n=65536)).to(dev)
ms = timeit(lambda: net(f64), iters=5, warm=2)
out.append(dict(op="SA backbone (configs[4])", case="16 x 65536-pt frames, one stream, eager", ms=round(ms, 2),
                frames_per_s=round(16 / ms * 1e3, 1)))
# points -> boxes: backbone + Det head + decode + per-class BEV NMS, batch 8, one stream, eager;
# the reference's README quotes "more than 25 FPS" for the whole detector (one frame at a time, other hardware)
M_ = importlib.import_module("3dssd_amd.modeling.single_stage_detector")
params = syn.random_backbone_params(cfgs.KITTI_3DSSD_ARCH)
syn.random_head_params(512, 1, cfgs.KITTI_ANGLE_CLS_NUM, params=params)
det = M_.SingleStageDetector(cfgs.KITTI_3DSSD_ARCH, cfgs.KITTI_3DSSD_HEAD, params, dev, cls_num=1, angle_cls_num=cfgs.KITTI_ANGLE_CLS_NUM,
                             max_output_size=cfgs.KITTI_MAX_OUTPUT_NUM, nms_threshold=cfgs.KITTI_NMS_THRESH)
for nb in (8, 1):
    fr = torch.from_numpy(syn.kitti_like_batch(nb)).to(dev)
    ms = timeit(lambda: det(fr), iters=10, warm=3)
    out.append(dict(op="detector points -> boxes", case="%d x 16384-pt frames, one stream, eager" % nb, ms=round(ms, 3),
                    frames_per_s=round(nb / ms * 1e3, 1)))
for o in out:
    print(json.dumps(o))
