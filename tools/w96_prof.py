"""Per-phase cycle breakdown of the 96-row fused MLP kernel (needs: tools/build_variant.sh w96prof mlp_wide128 -DSA_W96_PROF)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SA3D_LIB", os.path.join(ROOT, "3dssd_amd", "csrc", "variants", "lib_w96prof.so"))
import numpy as np, torch
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
cfgs, syn = pkg("configs"), pkg("synthetic")
net = pkg("backbone").SABackbone(cfgs.KITTI_3DSSD_ARCH, syn.random_backbone_params(cfgs.KITTI_3DSSD_ARCH), "cuda:0")
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pts = torch.from_numpy(np.stack([syn.frame_of("default", f, 16384) for f in range(NF)])).cuda()
lib = pkg("utils._native").lib()
lib.sa_debug_w96_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3):
    net(pts)
torch.cuda.synchronize()
lib.sa_debug_w96_prof(None, 1)
R = 10
for _ in range(R):
    net(pts)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.sa_debug_w96_prof(buf, 0)
a = np.array(list(buf), dtype=np.float64)
items = max(a[5], 1)
print("%d frames: items per forward %.0f; cycles per item: gather %.0f  hidden1 %.0f  hidden2 %.0f  last-layer loops %.0f  pool/write %.0f  total %.0f"
      % (NF, items / R, a[0] / items, a[1] / items, a[2] / items, a[3] / items, a[4] / items, a[:5].sum() / items))
print("wait at the top-of-item barrier per wave (cycles per item):", " ".join("%.0f" % (x / items) for x in a[8:16]))
