"""Randomised parity sweep: every operator of the C ABI on random small/ragged shapes against the CPU oracle
(bit-exact for index/byte results, 1e-3 relative for the MLP).  Test infrastructure (uses oracle/), meant for the GPU
box:  python tests/fuzz_ops.py [seconds] [seed] [big]   (big: mid-size shapes of the hot-path kernels) -> prints one line per failure and a summary; exit code 1 on failure.
"""
import ctypes
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import interp_oracle as IO  # noqa: E402
from oracle import sa_oracle as O  # noqa: E402


def pkg(name):
    return importlib.import_module("3dssd_amd." + name)


S, G = pkg("utils.tf_ops.sampling.tf_sampling"), pkg("utils.tf_ops.grouping.tf_grouping")
I, M = pkg("utils.tf_ops.interpolation.tf_interpolate"), pkg("utils.model_util")
N, Wt = pkg("utils._native"), pkg("utils.weights")
dev = torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def cloud(rng, b, n, c=3, dup=0.0, lattice=False):
    p = rng.normal(0, 2, (b, n, c)).astype(np.float32)
    if lattice:
        p = np.round(p * 2) / 2
    if dup > 0 and n > 2:
        k = max(1, int(n * dup))
        for i in range(b):
            p[i, rng.integers(0, n, k)] = p[i, rng.integers(0, n, k)]
    return p


def eq(name, got, ref, info):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    if got.shape != ref.shape or not np.array_equal(got, ref):
        bad = int((got != ref).sum()) if got.shape == ref.shape else -1
        return "%s MISMATCH %s (%d elements)" % (name, info, bad)
    return None


def case_fps(rng):
    b, c = int(rng.integers(1, 4)), int(rng.choice([3, 3, 3, 1, 2, 5, 16, 67]))
    n = int(rng.choice([1, 2, 63, 64, 65, 500, 1023, 1024, 1025, 2047, 3000, 4097, 9000]))
    if c == 67:
        n = min(n, 3000)
    m = int(rng.integers(1, min(n, 300) + 1))
    p = cloud(rng, b, n, c, dup=float(rng.choice([0, 0.1, 0.5])), lattice=bool(rng.integers(0, 2)))
    return eq("farthest_point_sample", S.farthest_point_sample(m, t(p)), O.farthest_point_sample(m, p), (b, n, c, m))


def case_fps_dist(rng):
    b, n = int(rng.integers(1, 3)), int(rng.choice([1, 5, 64, 100, 513, 1024, 1500]))
    m = int(rng.integers(1, min(n, 200) + 1))
    d = rng.uniform(-1, 5, (b, n, n)).astype(np.float32)
    if rng.integers(0, 2):
        d = np.round(d)
    return eq("fps_with_distance", S.farthest_point_sample_with_distance(m, t(d)),
              O.farthest_point_sample_with_distance(m, d), (b, n, m))


def case_fps_preidx(rng):
    b, c, n = int(rng.integers(1, 3)), int(rng.choice([3, 3, 7])), int(rng.choice([3, 70, 1024, 1100, 2500]))
    m, m1 = int(rng.integers(1, min(n, 60) + 1)), int(rng.integers(0, 9))
    p = cloud(rng, b, n, c, dup=float(rng.choice([0, 0.3])), lattice=bool(rng.integers(0, 2)))
    pre = rng.integers(0, n, (b, m1)).astype(np.int32)
    return eq("fps_with_preidx", S.farthest_point_sample_with_preidx(m, t(p), t(pre)),
              O.farthest_point_sample_with_preidx(m, p, pre), (b, n, c, m, m1))


def case_gather(rng):
    b, n, c, m = int(rng.integers(1, 4)), int(rng.integers(1, 900)), int(rng.choice([1, 3, 4, 7, 64, 130])), int(rng.integers(1, 300))
    p = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m)).astype(np.int32)
    e = eq("gather_point", S.gather_point(t(p), t(idx)), O.gather_point(p, idx), (b, n, c, m))
    if e:
        return e
    ns = int(rng.integers(1, 40))
    gi = rng.integers(-1, n, (b, m, ns)).astype(np.int32)
    return eq("group_point", G.group_point(t(p), t(gi)), O.group_point(p, gi), (b, n, c, m, ns))


def case_ball(rng):
    b, n = int(rng.integers(1, 3)), int(rng.choice([1, 30, 64, 65, 511, 512, 513, 1024, 1025, 2000, 2048, 2049, 5000]))
    m, ns = int(rng.integers(1, 200)), int(rng.choice([1, 2, 16, 32, 33, 64, 65, 100, 300]))
    xyz = cloud(rng, b, n, 3, dup=float(rng.choice([0, 0.2])), lattice=bool(rng.integers(0, 2)))
    ctr = xyz[:, rng.integers(0, n, m)] + (rng.normal(0, 0.2, (b, m, 3)).astype(np.float32) if rng.integers(0, 2) else 0)
    ctr = np.ascontiguousarray(ctr, np.float32)
    r0, r1 = sorted(rng.choice([0.0, 0.25, 0.5, 1.0, 1.5, 3.0, 50.0], 2, replace=False).tolist())
    i1, c1 = G.query_ball_point_dilated(r0, r1, ns, t(xyz), t(ctr))
    ri, rc = O.query_ball_point_dilated(r0, r1, ns, xyz, ctr)
    e = eq("query_ball_point_dilated idx", i1, ri, (b, n, m, ns, r0, r1)) or eq("query_ball_point_dilated cnt", c1, rc, (b, n, m, ns, r0, r1))
    if e:
        return e
    i2, c2 = G.query_ball_point(r1, ns, t(xyz), t(ctr))
    ri, rc = O.query_ball_point(r1, ns, xyz, ctr)
    e = eq("query_ball_point idx", i2, ri, (b, n, m, ns, r1)) or eq("query_ball_point cnt", c2, rc, (b, n, m, ns, r1))
    if e or n > 600:
        return e
    order = np.stack([np.stack([rng.permutation(n) for _ in range(m)]) for _ in range(b)]).astype(np.int32)
    i3, c3 = G.query_ball_point_withidx(r1, ns, t(xyz), t(ctr), t(order))
    ri, rc = O.query_ball_point_withidx(r1, ns, xyz, ctr, order)
    return eq("withidx idx", i3, ri, (b, n, m, ns, r1)) or eq("withidx cnt", c3, rc, (b, n, m, ns, r1))


def case_sqdist(rng):
    b, n, m = int(rng.integers(1, 3)), int(rng.choice([1, 31, 128, 129, 300, 512])), int(rng.choice([1, 64, 127, 300, 512]))
    c = int(rng.choice([1, 3, 4, 35, 67, 68, 131]))
    a = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    bb = a if (n == m and rng.integers(0, 2)) else rng.normal(0, 1, (b, m, c)).astype(np.float32)
    ta = t(a)
    tb = ta if bb is a else t(bb)
    return eq("calc_square_dist", M.calc_square_dist(ta, tb, norm=False), O.calc_square_dist(a, bb), (b, n, m, c, bb is a))


def case_mlp(rng):
    b, n, m = int(rng.integers(1, 3)), int(rng.integers(4, 400)), int(rng.integers(1, 120))
    c = int(rng.choice([0, 1, 1, 5, 8, 64, 64, 128, 256]))
    ns = int(rng.choice([1, 3, 8, 16, 20, 32, 48, 64, 70]))
    if rng.integers(0, 8) == 0:                      # several 4096-ball chunks of the row plan (round 3)
        b, m, c, ns = 2, int(rng.integers(2100, 6000)), int(rng.choice([1, 8, 64])), int(rng.choice([3, 8, 20]))
    nl = int(rng.integers(1, 4))
    wide = int(rng.choice([16, 32, 64, 128, 256])) if c < 128 else int(rng.choice([128, 256]))
    dims = [int(rng.choice([wide, wide, wide // 2 + 4, wide + 8])) for _ in range(nl)]
    if rng.integers(0, 3) == 0 and nl == 3:
        dims = {1: [16, 16, 32], 64: [64, 64, 128], 128: [128, 192, 256], 256: [256, 256, 512]}.get(c, dims)
    if rng.integers(0, 6) == 0 and c >= 8:           # shapes of the 96-row kernel: first hidden width 256, ragged others
        nl, dims = 3, [256, int(rng.choice([128, 160, 256, 384, 512])), int(rng.choice([512, 544, 1024, 800]))]
    xyz = cloud(rng, b, n)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32) if c else None
    new_xyz = np.ascontiguousarray(xyz[:, rng.integers(0, n, m)] + rng.normal(0, 0.1, (b, m, 3)), np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    dense = int(rng.integers(0, 2))                  # 1: all nsample rows count; 0: ball-query format (rows padded
    if not dense:                                    # with the first hit), only the distinct rows are evaluated
        idx = np.where(np.arange(ns)[None, None, :] >= np.maximum(cnt, 1)[:, :, None], idx[:, :, :1], idx)
    cin = [c + 3] + dims[:-1]
    ws = [rng.normal(0, 1.0 / np.sqrt(k), (k, o)).astype(np.float32) for k, o in zip(cin, dims)]
    bs = [rng.normal(0, 0.1, o).astype(np.float32) for o in dims]
    layers = Wt.pack_scale(ws, bs, dev) if rng.integers(0, 2) else [Wt.PackedLayer(w, x, dev) for w, x in zip(ws, bs)]
    out = torch.empty((b, m, dims[-1]), dtype=torch.float32, device=dev)
    dm = (ctypes.c_int * (nl + 1))(*([c + 3] + dims))
    tx, tn, ti, tc = t(xyz), t(new_xyz), t(idx), t(cnt)
    tf = t(feat) if c else None
    chain = int(rng.integers(0, 2)) * 16             # opt-in GEMM chain (taken only by eligible fp16 scales with the big scratch)
    chain |= int(rng.integers(0, 2)) * 32            # force the 96-row kernel wherever it supports the shape (csrc/mlp_wide128.hip)
    plan, plan_bytes = N.mlp_plan_ws(b, m, ns, dev, c, [c + 3] + dims) if chain else N.mlp_plan_ws(b, m, ns, dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    st = N.lib().sa_group_mlp_max(b, n, m, ns, c, tx.data_ptr(), tf.data_ptr() if c else None, tn.data_ptr(), ti.data_ptr(),
                                  tc.data_ptr(), nl, dm, (ctypes.c_void_p * nl)(*[l.w.data_ptr() for l in layers]),
                                  (ctypes.c_void_p * nl)(*[l.bias.data_ptr() for l in layers]), out.data_ptr(), dims[-1], 0,
                                  plan.data_ptr(), plan_bytes, dense | chain | Wt.scale_flags(layers), ovf.data_ptr(), N.current_stream())
    if st != 0:
        return "group_mlp_max status %d %s" % (st, (b, n, m, c, ns, dims))
    torch.cuda.synchronize()
    if int(ovf.item()) != 0:
        return "group_mlp_max raised the fp16 range flag on in-range data %s" % ((b, n, m, c, ns, dims),)
    ref = O.group_mlp_max(xyz, feat, new_xyz, idx, cnt, ws, bs)
    got = out.cpu().numpy()
    err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
    if not np.isfinite(got).all() or err > 1e-3 or (got[cnt == 0] != 0).any():
        return "group_mlp_max rel err %.3g %s" % (err, (b, n, m, c, ns, dims))
    return None


def case_interp(rng):
    b, n, m, c = int(rng.integers(1, 3)), int(rng.integers(1, 700)), int(rng.choice([1, 2, 3, 50, 2047, 2049, 5000])), int(rng.choice([1, 7, 64]))
    x1 = cloud(rng, b, n, lattice=bool(rng.integers(0, 2)))
    x2 = cloud(rng, b, m, lattice=bool(rng.integers(0, 2)))
    d, ix = I.three_nn(t(x1), t(x2))
    rd, ri = IO.three_nn(x1, x2)
    e = eq("three_nn dist", d, rd, (b, n, m)) or eq("three_nn idx", ix, ri, (b, n, m))
    if e:
        return e
    pts = rng.normal(0, 1, (b, m, c)).astype(np.float32)
    w = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
    return eq("three_interpolate", I.three_interpolate(t(pts), t(ri), t(w)), IO.three_interpolate(pts, ri, w), (b, n, m, c))


def case_boxes(rng):
    b, n, m, ns = int(rng.integers(1, 3)), int(rng.choice([1, 63, 64, 65, 1000, 3000])), int(rng.integers(1, 40)), int(rng.choice([1, 5, 64, 200]))
    xyz = cloud(rng, b, n, lattice=bool(rng.integers(0, 2)))
    boxes = np.zeros((b, m, 7), np.float32)
    ctr = xyz[:, rng.integers(0, n, m)]
    boxes[..., :3] = ctr + rng.normal(0, 0.5, (b, m, 3))
    boxes[..., 3:6] = rng.uniform(0, 5, (b, m, 3))
    boxes[..., 6] = rng.choice([0, np.pi / 2, np.pi, -np.pi / 2, 0.3, 1.7, -2.9], (b, m))
    if rng.integers(0, 2):
        boxes = np.round(boxes * 2) / 2
    boxes = boxes.astype(np.float32)
    i1, c1 = G.query_boxes_3d_points(ns, t(xyz), t(boxes))
    ri, rc = O.query_boxes_3d_points(ns, xyz, boxes)
    e = eq("boxes_points idx", i1, ri, (b, n, m, ns)) or eq("boxes_points cnt", c1, rc, (b, n, m, ns))
    e = e or eq("boxes_mask", G.query_boxes_3d_mask(t(xyz), t(boxes)), O.query_boxes_3d_mask(xyz, boxes), (b, n, m))
    if e:
        return e
    g = int(rng.integers(1, min(m, 5) + 1))
    gt = boxes[:, :g].copy()
    iou = rng.uniform(0, 0.004, (b, m, g)).astype(np.float32)
    return eq("points_iou", G.query_points_iou(t(xyz), t(boxes), t(gt), t(iou)), O.query_points_iou(xyz, boxes, gt, iou), (b, n, m, g))


def case_misc(rng):
    b, n, c, pn = int(rng.integers(1, 4)), int(rng.choice([1, 255, 256, 257, 1000])), int(rng.choice([1, 3, 64])), int(rng.choice([1, 7, 300]))
    inp = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    mask = ((rng.uniform(0, 1, (b, n)) < rng.choice([0.0, 0.02, 0.5, 1.0])) * rng.choice([1.0, -3.0, 0.7, 1.2], (b, n))).astype(np.float32)
    e = eq("gather_by_mask", S.gather_by_mask(pn, t(inp), t(mask)), O.gather_by_mask(pn, inp, mask), (b, n, c, pn))
    if e:
        return e
    m, k = int(rng.integers(1, 20)), int(rng.choice([1, 3, 16, 400]))
    d = (rng.integers(0, 30, (b, m, n)) * 0.5).astype(np.float32)
    oi, o = G.select_top_k(k, t(d))
    roi, ro = O.select_top_k(k, d)
    e = eq("select_top_k val", o, ro, (b, m, n, k)) or eq("select_top_k idx", oi, roi, (b, m, n, k))
    if e:
        return e
    gi = rng.integers(-1, n, (b, m, 6)).astype(np.int32)
    gg = rng.integers(-4, 5, (b, m, 6, c)).astype(np.float32)
    return eq("group_point_grad", G.group_point_grad(t(inp), t(gi), t(gg)), O.group_point_grad(inp, gi, gg), (b, n, c, m))


def case_vote_tail(rng):
    # sa_vote_tail against the three launches it replaces: the same bits
    rows, K, H = int(rng.choice([1, 31, 32, 33, 500, 2048])), int(rng.choice([1, 16, 100, 256, 300])), int(rng.choice([1, 3, 32, 40, 64, 128]))
    x = rng.normal(0, 1, (rows, K)).astype(np.float32)
    xyz = rng.uniform(-30, 30, (rows, 3)).astype(np.float32)
    L1 = Wt.PackedLayer(rng.normal(0, 1 / np.sqrt(K), (K, H)).astype(np.float32), rng.normal(0, 0.2, H).astype(np.float32), dev)
    L2 = Wt.PackedLayer(rng.normal(0, 3 / np.sqrt(H), (H, 3)).astype(np.float32), rng.normal(0, 0.5, 3).astype(np.float32), dev)
    tx, tp, lo = t(x), t(xyz), (-3.0, -2.0, -3.0)
    lib, st = N.lib(), N.current_stream()
    new = lambda *sh: torch.full(sh, -5.0, dtype=torch.float32, device=dev)
    h_a, o_a, out_a, h_b, o_b, out_b = new(rows, H), new(rows, 3), new(rows, 3), new(rows, H), new(rows, 3), new(rows, 3)
    ok = lib.sa_dense(rows, K, H, tx.data_ptr(), L1.w.data_ptr(), L1.bias.data_ptr(), 1, h_a.data_ptr(), st) == 0
    ok = ok and lib.sa_dense(rows, H, 3, h_a.data_ptr(), L2.w.data_ptr(), L2.bias.data_ptr(), 0, o_a.data_ptr(), st) == 0
    ok = ok and lib.sa_vote_translate(rows, tp.data_ptr(), o_a.data_ptr(), *lo, out_a.data_ptr(), st) == 0
    ok = ok and lib.sa_vote_tail(rows, K, H, tx.data_ptr(), L1.w.data_ptr(), L1.bias.data_ptr(), L2.w.data_ptr(), L2.bias.data_ptr(),
                                 h_b.data_ptr(), o_b.data_ptr(), tp.data_ptr(), *lo, out_b.data_ptr(), st) == 0
    if not ok:
        return "vote_tail: a launch failed %r" % ((rows, K, H),)
    return (eq("vote_tail hidden", h_b, h_a.cpu().numpy(), (rows, K, H)) or eq("vote_tail offsets", o_b, o_a.cpu().numpy(), (rows, K, H))
            or eq("vote_tail out", out_b, out_a.cpu().numpy(), (rows, K, H)))


def case_fps_big(rng):
    """layer-1 shapes: the wave-bucket kernel (8192 <= n <= 16384, m >= 64), ragged n, clustered / duplicated / flat data"""
    b, n = int(rng.integers(1, 4)), int(rng.integers(8192, 16385))
    m = int(rng.integers(64, 700))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        p = pkg("synthetic").kitti_like_batch(b, n=n)[:, :, :3].copy()
    elif kind == 1:
        p = cloud(rng, b, n, 3, dup=0.3, lattice=True)
    elif kind == 2:                                       # a few tight clusters + a far outlier
        p = (rng.normal(0, 0.01, (b, n, 3)) + rng.integers(0, 5, (b, n, 1)) * 3.0).astype(np.float32)
        p[:, n // 2] = 1000.0
    else:
        p = cloud(rng, b, n, 3)
        p[:, :, 1] = 0.0                                  # flat
    return eq("farthest_point_sample(big)", S.farthest_point_sample(m, t(p)), O.farthest_point_sample(m, p), (b, n, m, kind))


def case_ball_multi(rng):
    """the fused per-layer call: 2-4 bands in one pass, grid and scan kernels"""
    b, n, m = int(rng.integers(1, 3)), int(rng.choice([300, 512, 1024, 4096, 12000])), int(rng.integers(1, 600))
    nb = int(rng.integers(2, 5))
    xyz = pkg("synthetic").kitti_like_batch(b, n=n)[:, :, :3].copy() if rng.integers(0, 2) else cloud(rng, b, n, 3, dup=0.1)
    ctr = np.ascontiguousarray(xyz[:, rng.integers(0, n, m)])
    edges = np.sort(rng.choice([0.1, 0.2, 0.4, 0.8, 1.6, 3.2, 4.8, 6.4], nb, replace=False)).astype(np.float32)
    dil = bool(rng.integers(0, 2))
    rmin = [0.0] + [float(e) for e in edges[:-1]]
    rmax = [float(e) for e in edges]
    ns = [int(rng.choice([8, 16, 32, 64])) for _ in range(nb)]
    idx = [torch.empty((b, m, k), dtype=torch.int32, device=dev) for k in ns]
    cnt = [torch.empty((b, m), dtype=torch.int32, device=dev) for _ in ns]
    lib = N.lib()
    args = (b, n, m, nb, (ctypes.c_float * nb)(*rmin), (ctypes.c_float * nb)(*rmax), (ctypes.c_int * nb)(*ns), 1 if dil else 0)
    tx, tc = t(xyz), t(ctr)
    ptrs = ((ctypes.c_void_p * nb)(*[x.data_ptr() for x in idx]), (ctypes.c_void_p * nb)(*[x.data_ptr() for x in cnt]))
    if rng.integers(0, 2):
        ws = torch.empty((lib.sa_query_ball_point_grid_ws_bytes(b, n, m) + 3) // 4, dtype=torch.int32, device=dev)
        st = lib.sa_query_ball_point_grid(*args, tx.data_ptr(), tc.data_ptr(), *ptrs, ws.data_ptr(), N.current_stream())
        which = "grid"
    else:
        st = lib.sa_query_ball_point_multi(*args, tx.data_ptr(), tc.data_ptr(), *ptrs, N.current_stream())
        which = "scan"
    if st != 0:
        return "ball multi status %d" % st
    for i in range(nb):
        ri, rc = (O.query_ball_point_dilated(rmin[i], rmax[i], ns[i], xyz, ctr) if dil else O.query_ball_point(rmax[i], ns[i], xyz, ctr))
        e = eq("ball multi idx[%d] %s" % (i, which), idx[i], ri, (b, n, m, nb, dil)) or eq("ball multi cnt[%d] %s" % (i, which), cnt[i], rc, (b, n, m, nb, dil))
        if e:
            return e
    return None


def case_sqdist_big(rng):
    b, n = int(rng.integers(1, 3)), int(rng.choice([512, 640, 1000, 1024, 2048]))
    c = int(rng.choice([67, 131, 35, 7, 64]))
    a = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    ta = t(a)
    e = eq("calc_square_dist(sym big)", M.calc_square_dist(ta, ta, norm=False), O.calc_square_dist(a, a), (b, n, c))
    if e:
        return e
    m = int(rng.choice([300, 512, 1111]))
    bb = rng.normal(0, 1, (b, m, c)).astype(np.float32)
    return eq("calc_square_dist(big)", M.calc_square_dist(ta, t(bb), norm=False), O.calc_square_dist(a, bb), (b, n, m, c))


def case_mlp_big(rng):
    """the backbone's own scales with enough balls for several tiles per workgroup"""
    c, ns, dims = [(1, 32, [16, 16, 32]), (1, 64, [32, 32, 64]), (64, 32, [64, 64, 128]), (64, 64, [64, 96, 128]),
                   (128, 32, [128, 128, 256]), (128, 32, [128, 192, 256]), (128, 32, [128, 256, 256]),
                   (256, 16, [256, 256, 512]), (256, 32, [256, 512, 1024])][int(rng.integers(0, 9))]
    b, n, m = int(rng.integers(1, 3)), int(rng.integers(300, 1500)), int(rng.integers(200, 1500))
    xyz = cloud(rng, b, n)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, rng.integers(0, n, m)], np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    dense = int(rng.integers(0, 2))                  # 1: all nsample rows count; 0: ball-query format (rows padded
    if not dense:                                    # with the first hit), only the distinct rows are evaluated
        idx = np.where(np.arange(ns)[None, None, :] >= np.maximum(cnt, 1)[:, :, None], idx[:, :, :1], idx)
    cin = [c + 3] + dims[:-1]
    ws = [rng.normal(0, 1.0 / np.sqrt(k), (k, o)).astype(np.float32) for k, o in zip(cin, dims)]
    bs = [rng.normal(0, 0.1, o).astype(np.float32) for o in dims]
    layers = Wt.pack_scale(ws, bs, dev)
    nl = 3
    out = torch.empty((b, m, dims[-1]), dtype=torch.float32, device=dev)
    tx, tn, ti, tc, tf = t(xyz), t(new_xyz), t(idx), t(cnt), t(feat)
    chain = 0 if dense else int(rng.integers(0, 2)) * 16
    plan, plan_bytes = N.mlp_plan_ws(b, m, ns, dev, c, [c + 3] + dims) if chain else N.mlp_plan_ws(b, m, ns, dev)
    st = N.lib().sa_group_mlp_max(b, n, m, ns, c, tx.data_ptr(), tf.data_ptr(), tn.data_ptr(), ti.data_ptr(), tc.data_ptr(), nl,
                                  (ctypes.c_int * 4)(*([c + 3] + dims)), (ctypes.c_void_p * nl)(*[l.w.data_ptr() for l in layers]),
                                  (ctypes.c_void_p * nl)(*[l.bias.data_ptr() for l in layers]), out.data_ptr(), dims[-1], 0,
                                  plan.data_ptr(), plan_bytes, dense | chain | Wt.scale_flags(layers), None, N.current_stream())
    if st != 0:
        return "group_mlp_max(big) status %d" % st
    torch.cuda.synchronize()
    ref = O.group_mlp_max(xyz, feat, new_xyz, idx, cnt, ws, bs)
    got = out.cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    if not np.isfinite(got).all() or err > 1e-3 or (got[cnt == 0] != 0).any():
        return "group_mlp_max(big) rel err %.3g %s" % (err, (b, n, m, c, ns, dims))
    return None


def case_dense_blocks(rng):
    """sa_dense on the 128-row block kernel (K a multiple of 192, N of 128, >= 192 blocks) against the 32-row kernel on
    pieces of the same rows (identical bits) and against the oracle on a sample of rows"""
    K, N_ = [(384, 128), (768, 256), (1536, 512), (192, 128), (576, 384), (128, 64), (256, 192)][int(rng.integers(0, 7))]
    wide = N_ % 128 == 0
    blocks = (int(rng.integers(192, 260)) // (N_ // 128) + 1) if wide else (int(rng.integers(512, 600)) // (N_ // 64) + 1)
    rows = blocks * 128 - int(rng.integers(0, 128))
    relu = int(rng.integers(0, 2))
    x = rng.normal(0, 1, (rows, K)).astype(np.float32)
    w = rng.normal(0, 1 / np.sqrt(K), (K, N_)).astype(np.float32)
    bias = rng.normal(0, 0.2, N_).astype(np.float32)
    L = Wt.PackedLayer(w, bias, dev)
    tx = t(x)
    lib, st = N.lib(), N.current_stream()
    y = torch.full((rows + 1, N_), -5.0, dtype=torch.float32, device=dev)
    y2 = torch.full((rows + 1, N_), -5.0, dtype=torch.float32, device=dev)
    ok = lib.sa_dense(rows, K, N_, tx.data_ptr(), L.w.data_ptr(), L.bias.data_ptr(), relu, y.data_ptr(), st) == 0
    piece = (191 // (N_ // 128)) * 128 if wide else (511 // (N_ // 64)) * 128
    for a in range(0, rows, piece):
        n = min(piece, rows - a)
        ok = ok and lib.sa_dense(n, K, N_, tx.data_ptr() + 4 * K * a, L.w.data_ptr(), L.bias.data_ptr(), relu,
                                 y2.data_ptr() + 4 * N_ * a, st) == 0
    torch.cuda.synchronize()
    if not ok:
        return "dense blocks: status"
    if not torch.equal(y, y2) or not bool((y[rows] == -5.0).all()):
        return "dense blocks: 128-row kernel differs from the 32-row kernel %s" % ((rows, K, N_, relu),)
    sample = np.unique(np.r_[rng.integers(0, rows, 200), rows - 1, 0])
    ref = O.dense(x[sample], w, bias, bool(relu))
    err = np.abs(y.cpu().numpy()[sample] - ref).max() / max(np.abs(ref).max(), 1e-30)
    return None if err < 1e-3 else "dense blocks: rel err %.3g %s" % (err, (rows, K, N_))


def case_mlp_replay(rng):
    """shapes of a coalesced replay: plans of several 4096-ball chunks (two-launch plan), the eight-wave streamed kernel of
    259-256-256-512 (b * m * ns / 32 >= 2048); the whole call against the same frames in two halves (identical bits),
    two frames against the oracle"""
    c, ns, dims, m = [(256, 16, [256, 256, 512], 256), (1, 32, [16, 16, 32], 2048), (64, 32, [64, 64, 128], 1024),
                      (128, 32, [128, 192, 256], 512), (256, 32, [256, 512, 1024], 256)][int(rng.integers(0, 5))]
    b, n = 2 * int(rng.integers(4, 11)), int(rng.integers(400, 900))
    xyz = cloud(rng, b, n)
    feat = rng.normal(0, 1, (b, n, c)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, rng.integers(0, n, m)], np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    idx = np.where(np.arange(ns)[None, None, :] >= np.maximum(cnt, 1)[:, :, None], idx[:, :, :1], idx)
    cin = [c + 3] + dims[:-1]
    ws = [rng.normal(0, 1.0 / np.sqrt(k), (k, o)).astype(np.float32) for k, o in zip(cin, dims)]
    bs = [rng.normal(0, 0.1, o).astype(np.float32) for o in dims]
    layers = Wt.pack_scale(ws, bs, dev)

    def run(sl):
        bb = sl.stop - sl.start
        out = torch.empty((bb, m, dims[-1]), dtype=torch.float32, device=dev)
        tx, tn, ti, tc, tf = t(xyz[sl]), t(new_xyz[sl]), t(idx[sl]), t(cnt[sl]), t(feat[sl])
        plan, plan_bytes = N.mlp_plan_ws(bb, m, ns, dev)
        st = N.lib().sa_group_mlp_max(bb, n, m, ns, c, tx.data_ptr(), tf.data_ptr(), tn.data_ptr(), ti.data_ptr(), tc.data_ptr(), 3,
                                      (ctypes.c_int * 4)(*([c + 3] + dims)), (ctypes.c_void_p * 3)(*[l.w.data_ptr() for l in layers]),
                                      (ctypes.c_void_p * 3)(*[l.bias.data_ptr() for l in layers]), out.data_ptr(), dims[-1], 0,
                                      plan.data_ptr(), plan_bytes, Wt.scale_flags(layers), None, N.current_stream())
        torch.cuda.synchronize()
        return st, out

    st, whole = run(slice(0, b))
    st1, lo = run(slice(0, b // 2))
    st2, hi = run(slice(b // 2, b))
    if st or st1 or st2:
        return "mlp replay: status %d %d %d" % (st, st1, st2)
    if not torch.equal(whole[:b // 2], lo) or not torch.equal(whole[b // 2:], hi):
        return "mlp replay: %d frames at once differ from the two halves %s" % (b, (c, ns, dims, m, n))
    ref = O.group_mlp_max(xyz[:2], feat[:2], new_xyz[:2], idx[:2], cnt[:2], ws, bs)
    got = whole[:2].cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    if not np.isfinite(got).all() or err > 1e-3 or (got[cnt[:2] == 0] != 0).any():
        return "mlp replay: rel err %.3g %s" % (err, (b, n, m, c, ns, dims))
    return None


BIG = [case_fps_big, case_ball_multi, case_sqdist_big, case_mlp_big, case_dense_blocks, case_mlp_replay, case_mlp_replay]

def case_pooling(rng):
    P = pkg("utils.tf_ops.points_pooling.points_pooling")
    bs, pn, pts, c = int(rng.integers(1, 3)), int(rng.integers(1, 20)), int(rng.choice([1, 63, 64, 65, 300])), int(rng.choice([1, 3, 16]))
    l, h, w, sn = int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.choice([1, 2, 8, 35]))
    box = np.concatenate([rng.normal(0, 3, (bs, pn, 3)), rng.uniform(0.5, 5, (bs, pn, 3))], -1).astype(np.float32)
    if rng.integers(0, 2):
        box = np.round(box * 2) / 2 + np.float32(0.5) * (box[..., :1] * 0 + np.array([0, 0, 0, 1, 1, 1], np.float32))
        box = box.astype(np.float32)
    ctr = box[..., :3].copy()
    ctr[..., 1] -= box[..., 4] / 2
    loc = (ctr[:, :, None, :] + rng.uniform(-0.7, 0.7, (bs, pn, pts, 3)) * box[:, :, None, 3:6]).astype(np.float32)
    if rng.integers(0, 2):
        loc = (np.round(loc * 4) / 4).astype(np.float32)
    pc = rng.normal(0, 1, (bs, pn, pts, c)).astype(np.float32)
    got = P.points_pooling(t(pc), t(box), t(loc), l=l, h=h, w=w, sample_num=sn)
    ref = O.points_pooling(pc, box, loc, l=l, h=h, w=w, sample_num=sn)
    for nm, a, b in zip(("features", "idx", "num", "pillars"), got, ref):
        e = eq("points_pooling " + nm, a, b, (bs, pn, pts, c, l, h, w, sn))
        if e:
            return e
    return None


def case_ffps_fly(rng):
    # F-FPS without the matrix (csrc/ffps_fly.hip) against the HIP matrix path on the same rows (itself fuzzed against the
    # oracle by case_sqdist / case_fps_dist): random range inside a larger tensor, lattice features (ties), duplicates
    n = int(rng.choice([1024, 2048, 4096]))
    b, pad = int(rng.integers(1, 5)), int(rng.integers(0, 300))
    m = int(rng.integers(1, 200))
    xyz = cloud(rng, b, n + pad, 3, dup=float(rng.choice([0, 0.2])))
    feat = rng.normal(0, 1, (b, n + pad, 64)).astype(np.float32)
    if rng.integers(0, 2):
        feat = np.round(feat * 2) / 2
        xyz = np.round(xyz * 2) / 2
    start = int(rng.integers(0, pad + 1))
    tx, tf = t(xyz), t(feat)
    out = torch.full((b, m), -1, dtype=torch.int32, device=dev)
    ws = torch.empty((int(N.lib().sa_ffps_fly_ws_bytes(b, n)) + 7) // 8, dtype=torch.int64, device=dev)
    st = N.lib().sa_ffps_fly_ex(b, n, 64, m, tx.data_ptr() + 12 * start, 3 * (n + pad), tf.data_ptr() + 256 * start, 64 * (n + pad),
                                ws.data_ptr(), out.data_ptr(), m, start, None, 0, N.current_stream())
    if st != 0:
        return "ffps_fly status %d" % st
    cat = torch.cat([tx[:, start:start + n], tf[:, start:start + n]], 2).contiguous()
    ref = S.farthest_point_sample_with_distance(m, M.calc_square_dist(cat, cat, norm=False)) + start
    return eq("ffps_fly", out, ref.cpu().numpy(), (b, n, m, start))


def case_prob_iou(rng):
    E = pkg("utils.tf_ops.evaluation.tf_evaluate")
    k = int(rng.integers(1, 200))
    gt = np.concatenate([rng.normal(0, 4, (k, 3)), rng.uniform(0.3, 5, (k, 3)), rng.uniform(-4, 4, (k, 1))], -1).astype(np.float32)
    det = (gt + rng.normal(0, 0.4, (k, 7)) * (rng.uniform(0, 1, (k, 1)) < 0.8)).astype(np.float32)
    det[:, 3:6] = np.abs(det[:, 3:6])
    b1, t1 = E.calc_iou_match(t(det), t(gt))
    rb, rt = O.calc_iou_match(det, gt)
    if np.abs(b1.cpu().numpy() - rb).max() > 5e-6 or np.abs(t1.cpu().numpy() - rt).max() > 5e-6:
        return "calc_iou_match differs by %.3g / %.3g (k = %d)" % (np.abs(b1.cpu().numpy() - rb).max(), np.abs(t1.cpu().numpy() - rt).max(), k)
    return None


CASES = [case_fps, case_fps, case_fps_dist, case_fps_preidx, case_gather, case_ball, case_ball, case_sqdist, case_mlp, case_mlp,
         case_mlp, case_interp, case_boxes, case_misc, case_pooling, case_prob_iou, case_vote_tail, case_ffps_fly]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cases = BIG if (len(sys.argv) > 3 and sys.argv[3] == "big") else CASES
    rng = np.random.default_rng(seed)
    t0, runs, fails = time.time(), {}, []
    while time.time() - t0 < budget:
        fn = cases[int(rng.integers(0, len(cases)))]
        sub = np.random.default_rng(int(rng.integers(0, 2 ** 31)))
        try:
            e = fn(sub)
        except Exception as ex:  # noqa: BLE001
            e = "%s raised %r" % (fn.__name__, ex)
        runs[fn.__name__] = runs.get(fn.__name__, 0) + 1
        if e:
            fails.append(e)
            print("FAIL", e, flush=True)
            if len(fails) > 20:
                break
    print("fuzz: %d cases in %.0f s, %d failures; per op: %s" % (sum(runs.values()), time.time() - t0, len(fails), runs))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
