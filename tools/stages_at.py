"""Single-stream stage table of the backbone at a given launch shape, for A/B of module-level switches:
   python tools/stages_at.py <frames> [attr=value ...]     e.g.  python tools/stages_at.py 32 MLP_GEMM_CHAIN=True
attr=value pairs are set on 3dssd_amd.utils.layers_util before the run (data=<variant> selects the frame generator).
Library variants via SA3D_LIB."""
import ast
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = lambda m: importlib.import_module("3dssd_amd." + m)


def main():
    frames = int(sys.argv[1])
    lu = pkg("utils.layers_util")
    data = "default"
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        if k == "data":                                      # synthetic.DATA_VARIANTS: default | dup10 | dense | rings64
            data = v
            continue
        setattr(lu, k, ast.literal_eval(v))
    dev = torch.device("cuda:0")
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE, True, None)
    x = torch.from_numpy(np.stack([syn.frame_of(data, f, 16384) for f in range(frames)])).to(dev)
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    st = bench.profile_stages(lambda: net(x), 6)
    tot = 0.0
    for s in st:
        us = s["avg_ms"] * 1e3 * s["calls_per_step"]
        if not s["kernel"].startswith("sa_fps"):
            tot += us
        print("%-28s %-84s %8.1f us" % (s["kernel"], s["label"][:84], us))
    print("frames %d %s: non-FPS total %.1f us = %.1f us per 8 frames" % (frames, " ".join(sys.argv[2:]), tot, tot * 8 / frames))


main()
