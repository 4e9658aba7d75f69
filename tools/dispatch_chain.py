"""A captured chain of K tiny dependent kernels replayed on S streams at once: time per kernel boundary vs S
(what a many-kernel step pays per launch when 16 graphs are in flight)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
N = importlib.import_module("3dssd_amd.utils._native")
S_ = importlib.import_module("3dssd_amd.utils.tf_ops.sampling.tf_sampling")
K = 50
x = torch.rand((8, 512, 3), device="cuda")
idx = torch.randint(0, 512, (8, 256), dtype=torch.int32, device="cuda")
for S in [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,2,4,8,16'.split(','))]:
    streams = [torch.cuda.Stream() for _ in range(S)]
    graphs = []
    for st in streams:
        with torch.cuda.stream(st):
            S_.gather_point(x, idx)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(K):
                y = S_.gather_point(x, idx)
        graphs.append(g)
    torch.cuda.synchronize()
    def go():
        evs = []
        for r in range(6):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); g.replay(); b.record()
                    evs.append((a, b))
        torch.cuda.synchronize()
        return evs
    go()
    d = [a.elapsed_time(b) for a, b in go()]
    print("streams %2d: chain of %d kernels %.3f ms mean -> %.2f us per kernel" % (S, K, sum(d) / len(d), 1e3 * sum(d) / len(d) / K))
