"""Workload of the round-5 roofline evidence: ONLY full 128-frame launches.

Warm-up and measured passes are eager passes of the backbone over the SAME launch shape the staged executor uses (16
batches x 8 frames), on one stream; nothing else runs in the process (no partly filled packages, no priming at other
sizes), so rocprofv3's per-kernel averages ARE per-128-frame figures.  In front of every C-ABI call a MARKER kernel is
launched (vote_translate_kernel on one point: not otherwise on the fused path), five in a row in front of every pass:
tools/summarize_128f.py cuts the kernel trace / counter rows at the markers and joins them with calls.json (this
script's output: per call its label, algorithmic flops and bytes, and for the grouped MLP the EXECUTED
flops from the plan headers).

    rocprofv3 --kernel-trace --stats ... -- python tools/prof_128f.py OUT_DIR [data] [passes]"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = lambda m: importlib.import_module("3dssd_amd." + m)


def main():
    out_dir = sys.argv[1]
    data = sys.argv[2] if len(sys.argv) > 2 else "default"
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    frames = 128
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device("cuda:0")
    cfgs, syn, lu = pkg("configs"), pkg("synthetic"), pkg("utils.layers_util")
    native = pkg("utils._native")
    arch = cfgs.KITTI_3DSSD_ARCH
    net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE, True, None,
                                     dfps_side_stream=5)
    x = torch.from_numpy(np.stack([syn.frame_of(data, f, 16384) for f in range(frames)])).to(dev)
    real = native.lib()
    one = torch.zeros((1, 3), dtype=torch.float32, device=dev)
    one_out = torch.zeros((1, 3), dtype=torch.float32, device=dev)

    def marker(k=1):
        for _ in range(k):
            real.sa_vote_translate(1, one.data_ptr(), one.data_ptr(), -3.0, -2.0, -3.0, one_out.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)

    class Proxy:
        def __init__(self):
            self.calls = []

        def __getattr__(self, name):
            fn = getattr(real, name)
            if not name.startswith("sa_") or name.endswith(("_ws_bytes", "_rows", "_state")):
                return fn

            def wrapped(*args):
                marker()
                fl, by, label = bench._algorithmic(name, args)
                self.calls.append({"call": name, "label": label, "flops": fl, "bytes": by})
                return fn(*args)
            return wrapped

    all_passes = []
    for p in range(2 + passes):                      # two warm passes, same shape (caches, workspaces)
        proxy = Proxy()
        native._LIB = proxy
        lu.PLAN_LOG = []
        try:
            marker(5)
            net(x)
            marker()                                 # closes the last call's group
            torch.cuda.synchronize()
        finally:
            native._LIB = real
        log, lu.PLAN_LOG = lu.PLAN_LOG, None
        # executed flops of the grouped-MLP calls, from the plan headers (granules x rows per granule x MACs per row)
        ex = {}
        for (b, m, ns, macs, plan) in log:
            h = plan[:4].cpu().tolist()
            ex.setdefault(m, [0.0, 0, 0])
            ex[m][0] += 2.0 * h[0] * (h[3] or 8) * macs
            ex[m][1] += h[0] * (h[3] or 8)
            ex[m][2] += h[2]
        for c in proxy.calls:
            if c["call"] in ("sa_group_mlp_max_layer", "sa_group_mlp_max"):
                m = int(c["label"].split("m=")[1].split()[0])
                if m in ex:
                    c["flops_executed"], c["rows_evaluated"], c["rows_distinct"] = ex[m]
        all_passes.append(proxy.calls)
    json.dump({"frames_per_launch": frames, "data": data, "warm_passes": 2, "passes": passes, "calls_per_pass": all_passes[-1],
               "marker_kernel": "vote_translate_kernel"}, open(os.path.join(out_dir, "calls.json"), "w"), indent=1)
    print("prof_128f: %d passes of %d calls, data=%s" % (passes, len(all_passes[-1]), data))


main()
