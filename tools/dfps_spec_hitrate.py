"""Host simulation for "two picks per barrier" in the layer-1 D-FPS (csrc/fps_bucket.hip).

The kernel's pick loop: box test of the new pick p against the 64 Morton buckets -> cull set S(p) (buckets whose
lower-bound distance to p is below their current maximum) -> owners re-evaluate the buckets of S(p) -> barrier -> arg-max
over the 64-entry table.  Speculation: the entries of the buckets NOT in S(p) cannot change, so their arg-max u is known
before any distance is evaluated; if the true next pick is u (it lies in an untouched bucket), u's updates could have been
applied in the same round.  This script replays exact FPS on the generator's frames with the kernel's bucket structure
and counts how often that guess is right -- the number the round-2 kernel variant (HISTORY.md section 6: built, 93 %
hits, 4.43 ms against 2.98 ms) and any future attempt should be judged by.

    python tools/dfps_spec_hitrate.py [frames per variant]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

syn = importlib.import_module("3dssd_amd.synthetic")


def morton_buckets(xyz, bucket=256):
    x, z = xyz[:, 0], xyz[:, 2]
    qx = np.clip(((x - x.min()) * (511.0 / max(x.max() - x.min(), 1e-20))).astype(np.int64), 0, 511)
    qz = np.clip(((z - z.min()) * (511.0 / max(z.max() - z.min(), 1e-20))).astype(np.int64), 0, 511)

    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    key = ((spread(qx) | (spread(qz) << 1)) << 20) | np.arange(len(xyz))
    order = np.argsort(key, kind="stable")
    return order.reshape(-1, bucket)


def replay(xyz, m):
    xyz = xyz.astype(np.float32)
    bk = morton_buckets(xyz)                                   # [nb, 256] original indices
    nb = bk.shape[0]
    P = xyz[bk]                                                # [nb, 256, 3]
    lo, hi = P.min(1), P.max(1)
    td = np.full(bk.shape, 1e38, np.float32)
    tiekey = ((bk & 1023) << 16) | (bk >> 10)                  # the reference's (k mod 1024, k) order
    cur = xyz[0]
    hits = touched = 0
    spec_in_cull_of_spec = 0
    for _ in range(1, m):
        e = np.maximum(np.maximum(lo - cur, cur - hi), 0.0).astype(np.float32)
        lb = (e * e).sum(1)
        bmax = td.max(1)
        S = lb * np.float32(1.0 - 1e-5) < bmax                 # buckets the kernel re-evaluates
        touched += int(S.sum())
        # the guess: arg-max over the untouched entries (value, then minimum tie key)
        guess = None
        if (~S).any():
            v = np.where(~S, bmax, -np.inf)
            b = np.flatnonzero(v == v.max())
            cand = [(int(tiekey[bi][td[bi] == bmax[bi]].min()), bi) for bi in b]
            guess = min(cand)
        d = ((P[S] - cur) ** 2).sum(-1).astype(np.float32)
        td[S] = np.minimum(td[S], d)
        bmax = td.max(1)
        b = np.flatnonzero(bmax == bmax.max())
        cand = [(int(tiekey[bi][td[bi] == bmax[bi]].min()), bi) for bi in b]
        true = min(cand)
        if guess is not None and guess == true:
            hits += 1
        kk = true[0]
        k = ((kk >> 16) & 1023) | ((kk & 0xFFFF) << 10)
        cur = xyz[k]
    return hits / (m - 1), touched / (m - 1) / nb


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for variant, n in (("default", 16384), ("dup10", 16384), ("rings64", 16384), ("default", 65536)):
        hr, tf = [], []
        for f in range(nf):
            pts = syn.frame_of(variant, 500 + f, n)
            h, t = replay(pts[:, :3], 4096)
            hr.append(h)
            tf.append(t)
        print("%-8s n=%-6d frames %d: next pick is the best UNTOUCHED entry in %.1f %% of the picks (min %.1f); buckets touched per pick %.2f of %d"
              % (variant, n, nf, 100 * np.mean(hr), 100 * min(hr), np.mean(tf) * (n // 256), n // 256))


main()
