"""Debug: the backbone over nf 65536-point frames in one pass against the same frames in passes of 16."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
syn, cfgs = pkg("synthetic"), pkg("configs")
dev = torch.device("cuda:0")
nf, n = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 65536
arch = cfgs.KITTI_3DSSD_ARCH
net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)
pts = torch.from_numpy(np.stack([syn.frame_of("default", 900 + i, n) for i in range(nf)])).to(dev)
parts = [net(pts[i:i + 16].contiguous()) for i in range(0, nf, 16)]
torch.cuda.synchronize()
whole = net(pts)
torch.cuda.synchronize()
names = ["xyz", "feat", "idx"]
for li in range(len(whole[0])):
    for k in range(3):
        w = whole[k][li]
        if w is None:
            continue
        ref = torch.cat([p[k][li] for p in parts])
        if not torch.equal(w, ref):
            bad = (w != ref).flatten(1).any(1).nonzero().flatten().tolist()
            print("list index %d %s differs in frames %s" % (li, names[k], bad[:20]))
print("done")
