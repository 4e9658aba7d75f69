"""Layer-1 D-FPS launches on S streams at once (8 frames each): time per launch (events on each stream) and WALL time of
the S launches vs S.  The first separates a slow-down of the kernel itself (clock, CU sharing) from queueing; the second
shows how many of the kernels really run side by side (S x 3 ms / wall).  argv[1] = torch | raw: streams from torch's
pool, or created with hipStreamCreateWithFlags and wrapped as torch.cuda.ExternalStream."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
kind = sys.argv[1] if len(sys.argv) > 1 else "torch"
N = importlib.import_module("3dssd_amd.utils._native"); syn = importlib.import_module("3dssd_amd.synthetic")
pts = syn.kitti_like_batch(8)[:, :, :3].copy()
t = torch.from_numpy(pts).cuda()
lib = N.lib()
if kind == "raw":
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    all_streams = []
    for _ in range(24):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0      # hipStreamNonBlocking
        all_streams.append(torch.cuda.ExternalStream(h.value))
else:
    all_streams = [torch.cuda.Stream() for _ in range(24)]
for S in (1, 8, 12, 16, 24):
    streams = all_streams[:S]
    outs = [torch.empty((8, 4096), dtype=torch.int32, device="cuda") for _ in range(S)]
    def go(reps):
        evs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            for s, o in zip(streams, outs):
                with torch.cuda.stream(s):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    assert lib.sa_fps_bucket_ex(8, 16384, 4096, t.data_ptr(), o.data_ptr(), 4096, 0, N.current_stream()) == 0
                    b.record()
                    evs.append((a, b))
        torch.cuda.synchronize()
        return evs, (time.perf_counter() - t0) * 1e3
    go(1)
    evs, wall = go(2)
    d = [a.elapsed_time(b) for a, b in evs]
    print("%s streams %2d: per-launch %.3f ms mean (%.3f .. %.3f) | wall of 2 rounds %.2f ms -> %.1f kernels side by side"
          % (kind, S, sum(d) / len(d), min(d), max(d), wall, 2 * S * min(d) / wall))
