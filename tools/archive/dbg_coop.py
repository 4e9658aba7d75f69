"""Debug: multi-launch path of the cooperative / plain multi-workgroup FPS (csrc/fps_coop.hip) at 65536-point frames."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
S, syn = pkg("utils.tf_ops.sampling.tf_sampling"), pkg("synthetic")
dev = torch.device("cuda:0")
nf, n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 65536, 1024
pts = torch.from_numpy(np.stack([syn.frame_of("default", 500 + i, n)[:, :3] for i in range(nf)])).to(dev)
ref = torch.cat([S.farthest_point_sample(m, pts[i:i + 8].contiguous()) for i in range(0, nf, 8)])
torch.cuda.synchronize()
allf = S.farthest_point_sample(m, pts)
torch.cuda.synchronize()
bad = (allf != ref).any(1).nonzero().flatten().tolist()
print("eager: frames differing from the 8-frame calls:", bad)
st = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    out = S.farthest_point_sample(m, pts)
for rep in range(3):
    g.replay()
    torch.cuda.synchronize()
    bad = (out != ref).any(1).nonzero().flatten().tolist()
    print("graph replay %d: frames differing:" % rep, bad)
