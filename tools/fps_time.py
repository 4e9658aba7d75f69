"""Times the layer-1 D-FPS kernel (8 frames, 16384 -> 4096) and checks it against the oracle on one frame.
usage: python tools/fps_time.py [reps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
N = importlib.import_module("3dssd_amd.utils._native"); syn = importlib.import_module("3dssd_amd.synthetic")
from oracle import sa_oracle as O
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pts = syn.kitti_like_batch(8)[:, :, :3].copy()
t = torch.from_numpy(pts).cuda()
out = torch.empty((8, 4096), dtype=torch.int32, device="cuda")
lib = N.lib()
def run():
    assert lib.sa_fps_bucket_ex(8, 16384, 4096, t.data_ptr(), out.data_ptr(), 4096, 0, N.current_stream()) == 0
run(); torch.cuda.synchronize()
ref = O.farthest_point_sample(4096, pts[:2])
ok = np.array_equal(out[:2].cpu().numpy(), ref)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps): run()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / reps
print("%.4f ms per launch, %.4f us per pick, oracle match (2 frames): %s" % (ms, ms * 1e3 / 4095, ok))
