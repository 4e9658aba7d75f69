"""Phase clocks of the wave-bucket D-FPS kernel (debug library built with -DSA_FPSB_PROF, pointed to by SA3D_LIB)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
N = importlib.import_module("3dssd_amd.utils._native")
syn = importlib.import_module("3dssd_amd.synthetic")
lib = ctypes.CDLL(os.environ["SA3D_LIB"])
dev = torch.device("cuda:0")
pts = torch.from_numpy(np.ascontiguousarray(syn.kitti_like_batch(1)[:, :, :3])).to(dev)
m = 4096
out = torch.empty((1, m), dtype=torch.int32, device=dev)
h = (ctypes.c_ulonglong * 16)()
lib.sa_debug_fpsb_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.sa_fps_bucket_ex2.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p] + [ctypes.c_int] * 2 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
for rep in range(2):
    lib.sa_debug_fpsb_prof(None, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    lib.sa_fps_bucket_ex2(1, 16384, m, pts.data_ptr(), 0, out.data_ptr(), m, 0, None, 0, None)
    e.record(); torch.cuda.synchronize()
    lib.sa_debug_fpsb_prof(h, 0)
    v = list(h)
    it = m - 1
    print("ms %.3f | wave0 cycles/iter: test %.0f process %.0f carry %.0f barrier %.0f select %.0f | active buckets/iter %.2f, bucket evaluations/iter %.2f, busy waves/iter %.2f | all waves: process %.0f barrier-wait %.0f cycles/iter/wave"
          % (s.elapsed_time(e), v[0] / it, v[1] / it, v[2] / it, v[3] / it, v[4] / it, v[5] / it, v[6] / it, v[7] / it, v[8] / it / 8, v[9] / it / 8))
