"""configs[2] / configs[4] FPS datapoints: cooperative multi-workgroup kernel (fps_coop.hip) vs SA_FPS_COOP=0.
Run:  python tools/fps_coop_bench.py            (one JSON line per case)"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = importlib.import_module("3dssd_amd.utils.tf_ops.sampling.tf_sampling")
syn = importlib.import_module("3dssd_amd.synthetic")
dev = torch.device("cuda:0")
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def once(x, m):
    S.farthest_point_sample(8, x)
    torch.cuda.synchronize()
    t0.record()
    r = S.farthest_point_sample(m, x)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1), r


tag = "coop" if os.environ.get("SA_FPS_COOP", "1") != "0" else "single-workgroup"
pf = torch.randn(32, 16384, 67, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
ms, r2 = once(pf, 4096)
print(json.dumps(dict(op="farthest_point_sample c=67 (configs[2])", case="32 x 16384 -> 4096", kernel=tag, ms=round(ms, 2),
                      frames_per_s=round(32 / ms * 1e3, 1), us_per_pick=round(ms * 1e3 / 4095 / 2, 2), checksum=int(r2.long().sum()))))
p64 = torch.from_numpy(syn.kitti_like_batch(16, n=65536)).to(dev)[:, :, :3].contiguous()
ms, r4 = once(p64, 4096)
print(json.dumps(dict(op="farthest_point_sample c=3 (configs[4])", case="16 x 65536 -> 4096", kernel=tag, ms=round(ms, 2),
                      frames_per_s=round(16 / ms * 1e3, 1), us_per_pick=round(ms * 1e3 / 4095, 2), checksum=int(r4.long().sum()))))
