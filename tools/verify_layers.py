"""Layer-by-layer comparison of the staged executor (captured graphs) with eager passes of the same batches:
   python tools/verify_layers.py [points] [batch] [coalesce] [streams] [rounds] [data]
Prints, per round, the first list entry (layer, kind) of any batch that differs and how many elements do."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda m: importlib.import_module("3dssd_amd." + m)


def main():
    a = sys.argv[1:]
    points, batch, coalesce, streams, rounds = (int(a[i]) if len(a) > i else d for i, d in enumerate((65536, 16, 2, 4, 6)))
    data = a[5] if len(a) > 5 else "default"
    dev = torch.device("cuda:0")
    cfgs, syn, P = pkg("configs"), pkg("synthetic"), pkg("pipeline")
    arch = cfgs.KITTI_3DSSD_ARCH
    params = syn.random_backbone_params(arch)
    if os.environ.get("VL_PAD_WS"):                     # debugging: a ball-query workspace with slack behind it
        lib = pkg("utils._native").lib()
        real = lib.sa_query_ball_point_grid_ws_bytes
        lib.sa_query_ball_point_grid_ws_bytes = lambda b, n, m: real(b, n, m) + (int(os.environ["VL_PAD_WS"]) << 20)
    pipe = P.SAPipeline(arch, params, dev, batch=batch, points=points, streams=streams, coalesce=coalesce, mode="staged", graphs=not os.environ.get("VL_EAGER"),
                        max_translate_range=cfgs.KITTI_MAX_TRANSLATE_RANGE)
    nb = streams * coalesce
    pool = [torch.from_numpy(np.stack([syn.frame_of(data, 1000 * r + f, points) for f in range(batch)])).to(dev) for r in range(nb)]
    eager = []
    for x in pool:
        xl, fl, il = pipe.net(x)
        eager.append(([t.clone() for t in xl], [t.clone() for t in fl], [None if t is None else t.clone() for t in il]))
    torch.cuda.synchronize()
    bad_total = 0
    for rd in range(rounds):
        tickets = [pipe.submit(x, sync_source=False) for x in pool]
        pipe.flush()
        for i, t in enumerate(tickets):
            t.wait()
            r = t._round
            xl, fl, il = r.slot.lists[r.size]
            lo, hi = t._part * batch, (t._part + 1) * batch
            for kind, got, ref in (("xyz", xl, eager[i][0]), ("feat", fl, eager[i][1]), ("idx", il, eager[i][2])):
                for layer, (g, e) in enumerate(zip(got, ref)):
                    if g is None or e is None:
                        continue
                    gg = g[lo:hi]
                    if not torch.equal(gg, e):
                        d = gg != e
                        fr = d.reshape(batch, -1).any(1).nonzero().flatten().tolist()
                        firsts = [int(d[f].reshape(d.shape[1], -1).any(1).nonzero()[0]) for f in fr]
                        print("round %d batch %d: %s[%d] differs in %d elements, frames %s, first differing row per frame %s" % (rd, i, kind, layer, int(d.sum()), fr, firsts))
                        f0, r0 = fr[0], firsts[0]
                        print("    got %s\n    ref %s" % (gg[f0, r0:r0 + 3].tolist(), e[f0, r0:r0 + 3].tolist()))
                        bad_total += 1
                        break
                else:
                    continue
                break
    print("rounds %d, batches per round %d: %d differing batches" % (rounds, nb, bad_total))
    N = pkg("utils._native")
    print("sticky sampler error word: %d" % N.lib().sa_coop_error_state(0))


if __name__ == "__main__":
    main()
