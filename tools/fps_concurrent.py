"""Layer-1 D-FPS launches on S streams at once (8 frames each): per-launch time vs S.  Separates slow-down of the
kernel itself (clock, CU sharing) from per-dispatch overheads of a many-kernel step."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
N = importlib.import_module("3dssd_amd.utils._native"); syn = importlib.import_module("3dssd_amd.synthetic")
pts = syn.kitti_like_batch(8)[:, :, :3].copy()
t = torch.from_numpy(pts).cuda()
lib = N.lib()
for S in (1, 2, 4, 8, 16, 24):
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [torch.empty((8, 4096), dtype=torch.int32, device="cuda") for _ in range(S)]
    reps = 4
    def go():
        evs = []
        for r in range(reps):
            for s, o in zip(streams, outs):
                with torch.cuda.stream(s):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    assert lib.sa_fps_bucket_ex(8, 16384, 4096, t.data_ptr(), o.data_ptr(), 4096, 0, N.current_stream()) == 0
                    b.record()
                    evs.append((a, b))
        torch.cuda.synchronize()
        return evs
    go()
    evs = go()
    d = [a.elapsed_time(b) for a, b in evs]
    print("streams %2d: per-launch %.3f ms mean, %.3f min, %.3f max" % (S, sum(d) / len(d), min(d), max(d)))
