"""How often does the matrix sampler find the row of its pick held ahead?  Library built with -DSA_FFPS_SPEC_STATS
(tools/build_variant.sh fspec fps "-DSA_FFPS_SPEC_STATS"), SA3D_LIB pointing at it."""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
cfgs, syn, N = pkg("configs"), pkg("synthetic"), pkg("utils._native")
lib = N.lib()
dbg = ctypes.CDLL(os.environ["SA3D_LIB"]).sa_debug_ffps_spec
dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
arch = cfgs.KITTI_3DSSD_ARCH
net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)
for variant in sys.argv[1:] or ["default"]:
    x = torch.from_numpy(np.stack([syn.frame_of(variant, f, 16384) for f in range(32)])).to(dev)
    net(x); torch.cuda.synchronize()
    dbg(None, 1)
    net(x); torch.cuda.synchronize()
    h = (ctypes.c_ulonglong * 4)()
    dbg(h, 0)
    v = list(h)
    print(variant, "picks", v[0], "not held", v[1], "= %.3f" % (v[1] / max(1, v[0])), "| clocks per pick %.0f" % (v[2] / max(1, v[0])), "workgroups", v[3])
