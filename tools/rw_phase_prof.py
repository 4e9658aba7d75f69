"""Per-phase clock breakdown of the row-wave MLP kernel (needs the SA_RW_TIMING build: variants/lib_rwtiming.so)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SA3D_LIB"] = os.path.join(ROOT, "3dssd_amd", "csrc", "variants", "lib_rwtiming.so")
import numpy as np, torch
native = importlib.import_module("3dssd_amd.utils._native")
cfgs = importlib.import_module("3dssd_amd.configs"); syn = importlib.import_module("3dssd_amd.synthetic")
real = native.lib()
raw = ctypes.CDLL(native.LIB_PATH)
raw.sa_debug_rw_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 9)()

class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name == "sa_group_mlp_max_layer":
            def wrapped_layer(*a):
                torch.cuda.synchronize(); raw.sa_debug_rw_prof(None, 1)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); st = fn(*a); e.record(); torch.cuda.synchronize()
                raw.sa_debug_rw_prof(buf, 0)
                v = list(buf)
                waves, tiles = max(v[8], 1), max(v[7], 1)
                print("layer call b=%d m=%d: %.3f ms | instrumented waves %d tiles/wave %.2f | per wave: prologue %d | per tile: [1] %d [2] %d [3] %d [4] %d [5] %d [6] %d | total/wave %d cycles" % (
                    a[1], a[3], s.elapsed_time(e), waves, tiles / waves, v[0] // waves, v[1] // tiles, v[2] // tiles, v[3] // tiles,
                    v[4] // tiles, v[5] // tiles, v[6] // tiles, sum(v[:7]) // waves))
                return st
            return wrapped_layer
        if name != "sa_group_mlp_max":
            return fn
        def wrapped(*a):
            torch.cuda.synchronize(); raw.sa_debug_rw_prof(None, 1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); st = fn(*a); e.record(); torch.cuda.synchronize()
            raw.sa_debug_rw_prof(buf, 0)
            v = list(buf)
            nl = a[10]; dims = [a[11][i] for i in range(nl + 1)]
            waves, tiles = max(v[8], 1), max(v[7], 1)
            if v[7]:
                if dims[0] > 100:    # streamed-weight kernel: phases are (prologue, compute, barrier wait, stage store, chunk issue, write)
                    print("m=%d ns=%d %s: %.3f ms | waves %d tiles/wave %.1f | per wave: prologue %d | per tile: convert %d hidden0 %d hidden1 %d gather-issue+last %d write %d | total/wave %d cycles" % (
                        a[2], a[3], "-".join(map(str, dims)), s.elapsed_time(e), waves, tiles / waves, v[0] // waves,
                        v[1] // tiles, v[2] // tiles, v[3] // tiles, v[4] // tiles, v[5] // tiles, sum(v[:7]) // waves))
                else:
                    print("m=%d ns=%d %s: %.3f ms | waves %d tiles/wave %.1f | per wave: weight copy %d | per tile: convert(+wait) %d issue-loads %d hidden0 %d hidden1 %d last+pool %d write %d | total/wave %d cycles" % (
                        a[2], a[3], "-".join(map(str, dims)), s.elapsed_time(e), waves, tiles / waves, v[0] // waves,
                        v[1] // tiles, v[2] // tiles, v[3] // tiles, v[4] // tiles, v[5] // tiles, v[6] // tiles, sum(v[:7]) // waves))
            return st
        return wrapped
native._LIB = Proxy()
dev = torch.device("cuda:0")
arch = cfgs.KITTI_3DSSD_ARCH
net = importlib.import_module("3dssd_amd.backbone").SABackbone(arch, syn.random_backbone_params(arch), dev)
pts = torch.from_numpy(syn.kitti_like_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 8)).to(dev)
for rep in range(2):
    print("--- rep", rep)
    net(pts)
