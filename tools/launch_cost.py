"""What does ONE more dependent launch cost under load?  The backbone through SAPipeline (16 slots, graphs) with K extra
4-byte sa_copy_blocks launches appended to every slot's chain; prints ms/step for each K.
usage: python tools/launch_cost.py [K ...]      (GPU box)"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda m: importlib.import_module("3dssd_amd." + m)


class Padded:
    def __init__(self, net, k, dev):
        self.net, self.k = net, k
        self.a = torch.zeros(1, 1, 16, device=dev)
        self.b = torch.zeros(1, 1, 16, device=dev)
        self.N = pkg("utils._native")

    def __call__(self, inp):
        r = self.net(inp)
        for _ in range(self.k):
            self.N.copy_blocks([(self.a, self.b, 1, 1, 16)])
        return r

    def raise_if_overflow(self):
        self.net.raise_if_overflow()


def main():
    ks = [int(a) for a in sys.argv[1:]] or [0, 16, 32]
    dev = torch.device("cuda:0")
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    params = syn.random_backbone_params(arch)
    P = pkg("pipeline")
    base = pkg("backbone").SABackbone(arch, params, dev, cfgs.KITTI_MAX_TRANSLATE_RANGE, True, None)
    batches = [torch.from_numpy(np.stack([syn.frame_of("default", 8 * i + f, 16384) for f in range(8)])).to(dev) for i in range(20)]
    for k in ks:
        pipe = P.SAPipeline(arch, params, dev, net=Padded(base, k, dev), max_translate_range=cfgs.KITTI_MAX_TRANSLATE_RANGE,
                            mode="slots", streams=16)
        res = []
        for rep in range(3):
            for i in range(24):
                pipe.submit(batches[i % 20], sync_source=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(128):
                pipe.submit(batches[i % 20], sync_source=False)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 128 * 1e3)
        print("extra launches per step %3d: ms/step %s" % (k, " ".join("%.4f" % r for r in res)), flush=True)
        del pipe


main()
