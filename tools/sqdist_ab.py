"""Time sa_calc_square_dist_split at the layer-2 F-FPS shape for every variant library given on the command line."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda:0")
b, n, c0, c1 = 8, 4096, 3, 64
xyz = torch.randn(b, n, c0, device=dev); feat = torch.randn(b, n, c1, device=dev)
out = torch.empty(b, n, n, device=dev)
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    f = lib.sa_calc_square_dist_split
    f.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 5 + [ctypes.c_void_p]
    def run():
        return f(b, n, n, c0, c1, xyz.data_ptr(), feat.data_ptr(), xyz.data_ptr(), feat.data_ptr(), out.data_ptr(), None)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    print("%-40s %.3f ms" % (os.path.basename(path), s.elapsed_time(e) / 10))
    if hasattr(lib, "sa_debug_sq_prof"):
        h = (ctypes.c_ulonglong * 8)()
        lib.sa_debug_sq_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.sa_debug_sq_prof(None, 1); run(); torch.cuda.synchronize(); lib.sa_debug_sq_prof(h, 0)
        v = list(h); wg = max(v[7], 1)
        print("   per workgroup (thread 0) cycles: stage loads+LDS %d | barrier %d | norms %d | k-loop %d | barrier+ %d... epilogue %d  (workgroups %d)" % (
            v[0] // wg, v[1] // wg, v[2] // wg, v[3] // wg, 0, v[4] // wg, wg))
