import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
cfgs, syn, M = pkg("configs"), pkg("synthetic"), pkg("utils.model_util")
dev = torch.device("cuda:0")
arch = cfgs.KITTI_3DSSD_ARCH
net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)
pts = torch.from_numpy(np.stack([syn.frame_of("default", 40 + f, 16384) for f in range(1)])).to(dev)
xl, fl, il = net(pts)
x, f = xl[1][:, :4096], fl[1][:, :4096]
cat = torch.cat([x, f], -1).contiguous()
D = M.calc_square_dist(cat, cat, norm=False).cpu().numpy()[0]
print("features: abs max %.3g mean %.3g; xyz abs max %.3g" % (f.abs().max().item(), f.abs().mean().item(), x.abs().max().item()))
print("D: min %.3g median %.3g max %.3g finite %s" % (D.min(), np.median(D), D.max(), np.isfinite(D).all()))
td = np.full(4096, 1e38, np.float32); old = 0; picks = [0]; vals = []
for it in range(1, 512):
    td = np.minimum(td, D[old]); old = int(np.argmax(td)); picks.append(old); vals.append(float(td[old]))
print("picks[:40]", picks[:40])
print("pick values[:10]", ["%.4g" % v for v in vals[:10]], "... [100:105]", ["%.4g" % v for v in vals[100:105]])
d = np.diff(picks)
print("index difference of consecutive picks: median %d, |d| < 64: %.3f" % (np.median(np.abs(d)), np.mean(np.abs(d) < 64)))
srt = np.sort(td)[::-1]
print("running distance after 512 picks: top values", srt[:8])
print("F-FPS index list of the kernel [:20]:", il[2][0, :20].tolist() if il[2] is not None else None)
