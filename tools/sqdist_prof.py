"""Phase clocks of the F-FPS matrix kernel (sqdist_mfma3_kernel) at the layer-2 shape: library built with
-DSA_SQ_TIMING (tools/build_variant.sh sqt sqdist "-DSA_SQ_TIMING"), pointed to by SA3D_LIB.
   SA3D_LIB=3dssd_amd/csrc/variants/lib_sqt.so python tools/sqdist_prof.py [frames]"""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.environ.get("SA3D_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3dssd_amd", "csrc", "lib3dssd_sa.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.sa_calc_square_dist_ws_bytes.argtypes = [ci] * 5
lib.sa_calc_square_dist_ws_bytes.restype = ctypes.c_size_t
lib.sa_calc_square_dist_self_ws.argtypes = [ci] * 4 + [vp, ci, vp, ci, vp, vp, vp]
timing = hasattr(lib, "sa_debug_sq_prof")
if timing:
    lib.sa_debug_sq_prof.argtypes = [vp, ci]
dev = torch.device("cuda:0")
for frames in ([int(a) for a in sys.argv[1:]] or [32, 128]):
    n, c1 = 4096, 64
    g = torch.Generator(device="cpu").manual_seed(1)
    xyz = (torch.rand((frames, n, 3), generator=g) * 40).to(dev)
    feat = torch.randn((frames, n, c1), generator=g).to(dev)
    dist = torch.empty((frames, n, n), dtype=torch.float32, device=dev)
    ws = torch.empty((lib.sa_calc_square_dist_ws_bytes(frames, n, n, 3 + c1, 1) + 3) // 4, dtype=torch.float32, device=dev)
    h = (ctypes.c_ulonglong * 8)()
    def call():
        return lib.sa_calc_square_dist_self_ws(frames, n, 3, c1, xyz.data_ptr(), n, feat.data_ptr(), n, dist.data_ptr(), ws.data_ptr(), None)
    for _ in range(max(20, 6400 // frames)):      # ~0.3 s of the kernel itself: clocks and power state settled
        call()
    torch.cuda.synchronize()
    times = []
    best = 1e9
    for rep in range(25):
        if timing:
            lib.sa_debug_sq_prof(None, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        st = lib.sa_calc_square_dist_self_ws(frames, n, 3, c1, xyz.data_ptr(), n, feat.data_ptr(), n, dist.data_ptr(), ws.data_ptr(), None)
        e.record(); torch.cuda.synchronize()
        assert st == 0, st
        best = min(best, s.elapsed_time(e))
        times.append(s.elapsed_time(e))
    gb = frames * n * n * 4 / 1e9
    times.sort()
    line = "frames %d: min %.3f median %.3f max %.3f ms, %.2f TB/s written (median)" % (frames, best, times[len(times) // 2], times[-1], gb / times[len(times) // 2])
    if timing:
        lib.sa_debug_sq_prof(h, 0)
        v = list(h); wg = max(1, v[7])
        names = ["load wait + staging", "matrix loop", "norms + barriers", "epilogue issue", "store drain"]
        tot = sum(v[:5])
        line += " | per workgroup (wave 0) clocks: " + ", ".join("%s %.0f" % (nm, v[i] / wg) for i, nm in enumerate(names)) + \
                " | total %.0f clocks = %.2f us of real time -> shader clock %.2f GHz, %.2f workgroups resident per CU" % (
                    tot / wg, v[5] / wg / 100.0, tot / max(1, v[5]) / 10.0, v[5] / 100.0 / (best * 1e3) / 256)
    print(line, flush=True)
    print("  checksum %.6e" % float(dist[0, 17, :64].sum()), flush=True)
