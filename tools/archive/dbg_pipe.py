"""Debug: staged pipeline at 65536-point frames, coalesce C, against eager per batch; prints the first differing list."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
syn, cfgs = pkg("synthetic"), pkg("configs")
dev = torch.device("cuda:0")
C, n, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 2 * C
arch = cfgs.KITTI_3DSSD_ARCH
pipe = pkg("pipeline").SAPipeline(arch, syn.random_backbone_params(arch), dev, batch=B, points=n, streams=4, coalesce=C,
                                  max_translate_range=cfgs.KITTI_MAX_TRANSLATE_RANGE, mode="staged")
batches = [torch.from_numpy(np.stack([syn.frame_of("default", 100 + B * i + j, n) for j in range(B)])).to(dev) for i in range(nb)]
eager = []
for t in batches:
    xl, fl, il = pipe.forward_eager(t)
    eager.append(([None if v is None else v.clone() for v in xl], [None if v is None else v.clone() for v in fl], [None if v is None else v.clone() for v in il]))
torch.cuda.synchronize()
for rnd in range(3):
    tk = [pipe.submit(t, sync_source=False) for t in batches]
    pipe.flush()
    for i, t in enumerate(tk):
        got = t.all_outputs()
        for k, nm in enumerate(("xyz", "feat", "idx")):
            for li in range(len(got[k])):
                a, b = got[k][li], eager[i][k][li]
                if a is None:
                    continue
                if not torch.equal(a, b):
                    bad = (a != b).flatten(1).any(1).nonzero().flatten().tolist()
                    print("round %d batch %d (slot part %d): list %d %s differs in frames %s" % (rnd, i, i % C, li, nm, bad))
                    break
print("done")
