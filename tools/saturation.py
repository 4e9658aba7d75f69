"""Service demand of every C-ABI call of a step at saturation.

With 16 steps in flight the chip is a shared first-come-first-served server (tools/microbench/dispatch_contention.hip:
a tiny kernel waits for the backlog of the kernels dispatched before it), so the throughput of the bench is bounded by
the SUM over the calls of a step of the chip time each one needs when the chip is busy -- not by their latencies alone.
This tool measures that per call: the call is re-issued (same arguments, all intermediates of one eager step kept
alive) R times in a graph, and the graph is replayed (a) on one stream and (b) on 16 streams at once:
    alone us      = time per launch on one stream
    saturated us  = wall time / (16 x R) with 16 streams replaying the same call concurrently
saturated << alone: the call is a latency chain that overlaps with itself (FPS); saturated ~ alone: it fills the chip.
The sum of the saturated column estimates the floor of ms_per_step at the bench's 16 streams.

    python tools/saturation.py > gpurun_out/saturation.txt
"""
import importlib
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pkg(name):
    return importlib.import_module("3dssd_amd." + name)


class Recorder:
    def __init__(self, real):
        self._real = real
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("sa_") or name.endswith("_ws_bytes"):
            return fn

        def wrapped(*args):
            self.calls.append((name, fn, args))
            return fn(*args)
        return wrapped


def main():
    dev = torch.device("cuda:0")
    nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cfgs, syn, native = pkg("configs"), pkg("synthetic"), pkg("utils._native")
    bench = importlib.import_module("bench")
    arch = cfgs.KITTI_3DSSD_ARCH
    net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)
    pts = torch.from_numpy(syn.kitti_like_batch(8, n=16384)).to(dev)
    net(pts)
    torch.cuda.synchronize()
    # one eager step with every allocation kept alive: the recorded pointers stay valid and hold this step's data
    keep, real_empty = [], torch.empty

    def empty_keep(*a, **k):
        t = real_empty(*a, **k)
        keep.append(t)
        return t
    rec = Recorder(native.lib())
    native._LIB = rec
    torch.empty = empty_keep
    try:
        out = net(pts)
        torch.cuda.synchronize()
    finally:
        torch.empty = real_empty
        native._LIB = rec._real
    keep.append(out)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    print("%-3s %-58s %10s %12s %8s" % ("#", "call", "alone us", "saturated us", "overlap"))
    tot_a = tot_s = 0.0
    for ci, (name, fn, args) in enumerate(rec.calls):
        label = bench._algorithmic(name, args)[2] if hasattr(bench, "_algorithmic") else name
        # alone, eager, to size R
        s0 = streams[0]
        with torch.cuda.stream(s0):
            fn(*args[:-1], s0.cuda_stream)
        s0.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s0):
            fn(*args[:-1], s0.cuda_stream)
        s0.synchronize()
        rough = (time.perf_counter() - t0) * 1e6
        R = int(min(40, max(2, 1500.0 / max(rough, 1.0))))
        graphs = []
        for s in streams:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(R):
                    fn(*args[:-1], torch.cuda.current_stream().cuda_stream)
            graphs.append(g)
        for s, g in zip(streams, graphs):
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()

        def timed(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for rep in range(3):
                for s, g in zip(streams[:k], graphs[:k]):
                    with torch.cuda.stream(s):
                        g.replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / (3 * k * R)
        alone = timed(1)
        sat = timed(nstreams)
        tot_a += alone
        tot_s += sat
        print("%-3d %-58s %10.1f %12.1f %8.1f" % (ci + 1, label[:58], alone, sat, alone / sat))
        del graphs
    print("sum: alone %.3f ms, saturated %.3f ms per step" % (tot_a / 1e3, tot_s / 1e3))


if __name__ == "__main__":
    main()
