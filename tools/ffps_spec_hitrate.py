"""Would a speculative row fetch help the matrix sampler (fpsdist_reg_body, csrc/fps.hip)?  F-FPS waits one HBM round trip per
pick for the row of the point just picked.  The runner-up of pick t (the second-largest running distance BEFORE row(p_t) is
applied) could be requested together with row(p_t); if it turns out to be pick t+1, that pick needs no memory wait.  This
tool replays the layer-2 / layer-3 F-FPS of real backbone passes on the host and counts how often pick t+1 equals
  (a) the exact runner-up, (b) the best of the OTHER 15 waves' maxima (what the kernel has at hand for free: thread k mod
  1024 owns point k, wave = (k mod 1024) / 64), (c) either of the two best other-wave maxima.
   python tools/ffps_spec_hitrate.py [variant ...]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
cfgs, syn, M = pkg("configs"), pkg("synthetic"), pkg("utils.model_util")
dev = torch.device("cuda:0")
arch = cfgs.KITTI_3DSSD_ARCH
net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)


def replay(D, m):
    n = D.shape[0]
    td = np.full(n, 1e38, np.float32)
    wave = (np.arange(n) % 1024) // 64
    old, hits = 0, np.zeros(7, int)
    where = np.zeros(3, int)                   # the next pick relative to the current one: same thread / same wave, other lane / other wave
    prev = None
    spec = None
    two_ahead = {}                             # static protocol: at pick t the row of c2 is requested for pick t + 2
    age = {}                                   # two-slot cache of the kernel: point -> iterations since its row was requested
    for it in range(1, m):
        if prev is not None:
            where[0 if old % 1024 == prev % 1024 else (1 if wave[old] == wave[prev] else 2)] += 1
        prev = old
        if spec is not None:
            hits += [old == spec[0], old == spec[1], old in spec[1:3], old in age, old in age and age[old] >= 2,
                     two_ahead.get(it) == old, old == spec[3]]
        # runner-up candidates BEFORE row(old) is applied (what is known when row(old) is requested)
        order = np.argsort(-td, kind="stable")
        exact = int(order[1]) if order[0] == old else int(order[0])        # best point other than `old` itself
        wmax = np.full(16, -1.0, np.float32); widx = np.zeros(16, int)
        for w in range(16):
            sel = np.nonzero(wave == w)[0]
            sel = sel[sel != old]
            if len(sel) == 0:
                continue
            j = sel[np.argmax(td[sel])]
            wmax[w], widx[w] = td[j], j
        wo = np.argsort(-wmax, kind="stable")
        ow = [w for w in wo if w != wave[old]]            # what the kernel sees: one value per wave, the winner's wave excluded
        spec = (exact, int(widx[wo[0]]), int(widx[wo[1]]), int(widx[ow[0]]))
        two_ahead[it + 2] = spec[2]
        # the kernel's cache: keep what is still one of the two candidates, request the others (two slots)
        age = {k: v + 1 for k, v in age.items() if k in spec[1:3] and k != old}
        for c in spec[1:3]:
            if c not in age and len(age) < 2:
                age[c] = 0
        td = np.minimum(td, D[old])
        old = int(np.argmax(td))
    print('      next pick vs current pick: same thread %.3f, same wave other lane %.3f, other wave %.3f; index difference median %d' % (where[0] / where.sum(), where[1] / where.sum(), where[2] / where.sum(), 0))
    return hits, m - 2


for variant in (sys.argv[1:] or ["default", "rings64"]):
    pts = torch.from_numpy(np.stack([syn.frame_of(variant, 40 + f, 16384) for f in range(2)])).to(dev)
    xl, fl, il = net(pts)
    torch.cuda.synchronize()
    for lvl, (nF, mF) in ((1, (4096, 512)), (2, (512, 256))):
        # the 'FS' rows sample F-FPS over ALL points of the level with xyz || features (layers_util.py:93-98)
        x, f = xl[lvl][:, :nF], fl[lvl][:, :nF]
        cat = torch.cat([x, f], -1).contiguous()
        D = M.calc_square_dist(cat, cat, norm=False).cpu().numpy()
        tot = np.zeros(7, int); cnt = 0
        for b in range(D.shape[0]):
            h, c = replay(D[b], mF)
            tot += h; cnt += c
        print("%s level %d (n=%d -> %d, c=%d): next pick == exact runner-up %.3f | best other-wave maximum %.3f | one of the two best other-wave maxima %.3f"
              " | in the two-slot cache %.3f, requested >= 3 picks earlier %.3f | static protocol (second candidate of pick t requested for pick t + 2) %.3f | best maximum of the waves other than the winner's %.3f"
              % (variant, lvl + 1, nF, mF, cat.shape[2], tot[0] / cnt, tot[1] / cnt, tot[2] / cnt, tot[3] / cnt, tot[4] / cnt, tot[5] / cnt, tot[6] / cnt), flush=True)
