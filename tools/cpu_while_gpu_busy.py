"""How much HOST CPU does this process burn while the GPU works and the Python thread sleeps?  (under a
CPU quota a process that spins is frozen for the rest of the 100 ms period.)  Enqueues `passes` eager backbone passes
over 128 frames (~10 ms each), then sleeps while they run and reads the process's CPU time and per-thread CPU times.
    python tools/cpu_while_gpu_busy.py [eager|graph] [passes]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda m: importlib.import_module("3dssd_amd." + m)


def thread_cpu():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception:  # noqa: BLE001
            pass
    return out


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE, True, None,
                                     dfps_side_stream=5)
    x = torch.from_numpy(np.stack([syn.frame_of("default", f, 16384) for f in range(128)])).to(dev)
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    g = None
    if mode == "graph":
        st = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            net(x)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        g.replay() if g is not None else net(x)
    t_issue = time.perf_counter() - t0
    done = torch.cuda.Event()
    done.record()
    th0, c0, w0 = thread_cpu(), time.process_time(), time.perf_counter()
    n = 0
    while not done.query():          # the Python thread sleeps; query() is a host-side check
        time.sleep(0.005)
        n += 1
    c1, w1, th1 = time.process_time(), time.perf_counter(), thread_cpu()
    busy = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1], th1[t][0], t) for t in th1), reverse=True)[:4]
    print("%s: %d passes issued in %.1f ms; GPU busy %.1f ms more while the Python thread slept (%d naps): process CPU %.1f ms = %.2f cores; busiest threads %s"
          % (mode, passes, t_issue * 1e3, (w1 - w0) * 1e3, n, (c1 - c0) * 1e3, (c1 - c0) / (w1 - w0),
             ", ".join("%s[%d] %.0f ms" % (c, t, d * 1e3) for d, c, t in busy)))


main()
