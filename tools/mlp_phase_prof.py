"""Per-phase clock breakdown of the fused MLP kernel (needs the SA_MLP_TIMING build: variants/lib_timing.so)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SA3D_LIB"] = os.path.join(ROOT, "3dssd_amd", "csrc", "variants", "lib_timing.so")
import numpy as np, torch
native = importlib.import_module("3dssd_amd.utils._native")
cfgs = importlib.import_module("3dssd_amd.configs"); syn = importlib.import_module("3dssd_amd.synthetic")
real = native.lib()
raw = ctypes.CDLL(native.LIB_PATH)
raw.sa_debug_mlp_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()

class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name != "sa_group_mlp_max":
            return fn
        def wrapped(*a):
            torch.cuda.synchronize(); raw.sa_debug_mlp_prof(None, 1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); st = fn(*a); e.record(); torch.cuda.synchronize()
            raw.sa_debug_mlp_prof(buf, 0)
            v = list(buf)
            nl = a[10]; dims = [a[11][i] for i in range(nl + 1)]
            tot = sum(v[:4]) or 1
            print("m=%d ns=%d %s: %.3f ms | items %d passes %d | per-pass cycles: gather %d hidden %d last+pool %d write %d | frac %s" % (
                a[2], a[3], "-".join(map(str, dims)), s.elapsed_time(e), v[4], v[5],
                v[0] // max(v[5], 1), v[1] // max(v[5], 1), v[2] // max(v[5], 1), v[3] // max(v[4], 1),
                " ".join("%.2f" % (x / tot) for x in v[:4])))
            return st
        return wrapped
native._LIB = Proxy()
dev = torch.device("cuda:0")
arch = cfgs.KITTI_3DSSD_ARCH
net = importlib.import_module("3dssd_amd.backbone").SABackbone(arch, syn.random_backbone_params(arch), dev)
pts = torch.from_numpy(syn.kitti_like_batch(8)).to(dev)
for rep in range(2):
    print("--- rep", rep)
    net(pts)
