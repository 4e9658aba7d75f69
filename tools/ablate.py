"""Marginal cost of every kernel class on the 16-slot throughput: the class's C-ABI calls are SKIPPED by a proxy around
the ctypes library (results are then garbage -- a timing experiment that lives here, not in the package or in bench.py)
and ms/step is compared with the full run.     usage: gpurun -- 'python tools/ablate.py > gpurun_out/ablate.txt'
(ball query and FPS cannot be ablated this way: their outputs shape the work of everything downstream)"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch

pkg = lambda n: importlib.import_module("3dssd_amd." + n)
LAYER_OF_M = {4096: "layer1", 1024: "layer2", 512: "layer3", 256: "layer4"}


class SkipProxy:
    def __init__(self, real, skip):
        self._real, self._skip = real, skip

    def __getattr__(self, name):
        fn = getattr(self._real, name)

        def wrapped(*a):
            cls = None
            if name in ("sa_group_mlp_max_layer", "sa_group_mlp_max"):
                cls = "mlp:" + LAYER_OF_M.get(a[3] if name.endswith("layer") else a[2], "?")
            elif name in ("sa_dense", "sa_vote_tail"):
                cls = "dense"
            elif name.startswith("sa_calc_square_dist"):
                cls = "sqdist"
            elif name == "sa_group_mlp_plan":
                cls = "plan"
            return 0 if cls in self._skip else fn(*a)
        return wrapped if name.startswith("sa_") and not name.endswith("_ws_bytes") else fn


COALESCE = int(os.environ.get("ABLATE_COALESCE", "4"))
MODE = os.environ.get("ABLATE_MODE", "slots")          # "staged": ABLATE_COALESCE=8 ABLATE_STREAMS=4
STREAMS = int(os.environ.get("ABLATE_STREAMS", "16"))


def run(skip, steps=512, warmup=64):
    native = pkg("utils._native")
    real = native.lib()
    native._LIB = SkipProxy(real, set(skip))
    try:
        cfgs, syn = pkg("configs"), pkg("synthetic")
        arch = cfgs.KITTI_3DSSD_ARCH
        pipe = pkg("pipeline").SAPipeline(arch, syn.random_backbone_params(arch), "cuda:0", batch=8, points=16384, streams=STREAMS,
                                          check_overflow=False, coalesce=COALESCE, mode=MODE)
    finally:
        native._LIB = real            # the graphs are captured: replays no longer go through ctypes
    batches = [torch.from_numpy(syn.kitti_like_batch(8, first_frame=8 * i)).cuda() for i in range(20)]
    torch.cuda.synchronize()
    for i in range(warmup):
        pipe.submit(batches[i % 20], sync_source=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.submit(batches[i % 20], sync_source=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if __name__ == "__main__":
    # one process per configuration (a pipeline's graphs keep their memory pools): argv = comma-separated classes, or "none"
    for a in sys.argv[1:] or ["none"]:
        skip = [] if a == "none" else a.split(",")
        print("%-70s ms/step %.4f" % (",".join(skip) or "none", run(skip)), flush=True)
