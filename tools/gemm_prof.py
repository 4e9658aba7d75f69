"""Per-phase cycle breakdown of the GEMM-chain kernels (needs the SA_GEMM_PROF build: tools/build_variant.sh gprof mlp_gemm
-DSA_GEMM_PROF).  usage (GPU box): python tools/gemm_prof.py"""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SA3D_LIB"] = os.path.join(ROOT, "3dssd_amd", "csrc", "variants", "lib_gprof.so")
import numpy as np, torch
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
cfgs, syn = pkg("configs"), pkg("synthetic")
net = pkg("backbone").SABackbone(cfgs.KITTI_3DSSD_ARCH, syn.random_backbone_params(cfgs.KITTI_3DSSD_ARCH), "cuda:0")
pts = torch.from_numpy(syn.kitti_like_batch(8)).cuda()
lib = pkg("utils._native").lib()
lib.sa_debug_gemm_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3):
    net(pts)
torch.cuda.synchronize()
lib.sa_debug_gemm_prof(None, 1)
R = 10
for _ in range(R):
    net(pts)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib.sa_debug_gemm_prof(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(4, 8)
for ph in (1, 2, 3):
    r = a[ph]
    works, stages, wgs = max(r[4], 1), max(r[5], 1), max(r[7], 1)
    print("phase %d: per launch %d work items on %d workgroups; per work item: setup %.0f  fill %.0f  loop %.0f (%.0f per stage, %.1f stages)  epilogue %.0f cycles; kernel span of a working workgroup %.0f"
          % (ph, works / R, wgs / R, r[0] / works, r[1] / works, r[2] / works, r[2] / stages, stages / works, r[3] / works, r[6] / wgs))
