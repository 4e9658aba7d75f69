"""SA-layer functions with the reference's signatures (lib/utils/layers_util.py): vote_layer (:12-24)
and pointnet_sa_module_msg (:59-189), inference mode, on torch-ROCm tensors.

The control flow (range slicing, FS = [F-FPS idx | D-FPS idx], index offsets, scale loop, aggregation)
follows the reference line by line; what differs is what runs underneath:
  * the radius bands of a layer go through ONE fused ball-query pass (csrc/ballquery.hip);
  * group_point x2 + concat + conv2d x3 + reduce_max + mask is ONE fused MFMA kernel per scale
    (csrc/mlp.hip) that writes straight into the concatenated [B,m,sum C] tensor;
  * the F-FPS matrix is built from (xyz, features) without materialising their concat.
Weights come from a VariableStore (utils/weights.py), the stand-in for TF variable scopes.
Arguments that only matter for training (is_training, bn_decay) are accepted and ignored; BatchNorm is
always the inference form.  use_attention (query_ball_point_withidx) is not on the 3DSSD path and is
rejected.
"""
import ctypes

import torch

from . import _native as N
from . import weights as W
from .tf_ops import _tensor as T
from .tf_ops.sampling.tf_sampling import farthest_point_sample, farthest_point_sample_with_distance, gather_point  # noqa: F401
from .tf_ops.grouping.tf_grouping import query_ball_point, query_ball_point_dilated, group_point  # noqa: F401

# cfg.MODEL.NETWORK.AGGREGATION_SA_FEATURE and cfg.MODEL.MAX_TRANSLATE_RANGE of the reference's global
# config (configs/kitti/3dssd/3dssd.yaml:39,44): the DEFAULTS of the keyword arguments `aggregation_sa_feature` /
# `max_translate_range` below.  SABackbone passes its own values per call (two backbones with different settings
# in one process do not share them); a caller that uses the reference's signatures only gets these.
AGGREGATION_SA_FEATURE = True
# frames with at least this many points go through the grid ball query (csrc/ballquery_grid.hip)
# sa_group_mlp_max flags: 0 = evaluate only the distinct rows of every ball (default), 1 = all nsample rows (A/B)
# Where the D-FPS half of an 'FS' layer (and of a two-range layer) runs, SA_DFPS_SIDE_STREAM:
#   2            on a helper stream, forked AFTER the F-FPS chain was enqueued and joined right behind it.  The two
#                halves still run back to back, but in a captured graph the D-FPS node sits on its own branch; with 16
#                graphs in flight this form measured 8.3k frames/s against 6.5k with everything on one stream (0, 3).
#   1            forked BEFORE the F-FPS chain: the two serial chains overlap, one batch's latency drops from 5.25 to
#                4.9 ms, but the throughput with 16 graphs in flight HALVES (4.2k frames/s: parallel branches of many
#                graphs compete for the hardware queues).  The right choice for single-frame latency.
#   0 / 3        no helper stream (D-FPS first / F-FPS first);   4: every FPS kernel on the helper stream (= 2).
#   5 / 6        the matrix sampler and the coordinate sampler of a layer in ONE launch (sa_fps_dual_ex), on the issuing
#                stream / on the helper stream between a fork and a join event.  6 is the DEFAULT: the two chains really
#                run side by side (layer2: 0.54 ms instead of 0.54 + 0.29; one batch 4.6 ms instead of 5.0) without a
#                second concurrent queue, throughput as mode 2 (10.4 k frames/s), +5 % in a 20-step run; the same launch
#                on the issuing stream (5) gives 7.9 k -- a captured graph WITHOUT a helper-stream branch loses a quarter
#                of the 16-stream throughput (also seen with modes 0 / 3), for reasons inside the graph executor.
# These are plain module attributes: nothing in the package reads the environment (experiments set them from tools/).
DFPS_SIDE_STREAM = 6
PLAN_LOG = None     # set to a list to collect (b, m, nsample, MACs per row, plan tensor) of every fused-MLP call
CONCAT_LOG = None   # set to a list to collect (scope, pooled concat tensor [B,m,sum N], offsets, widths, precisions) per SA layer
MLP_PLAN_FLAGS = 0  # sa_group_mlp_max flag bit 0 (all nsample rows instead of the distinct ones), A/B measurements
# Row plans in granules of 4 rows for the scales the row-wave kernels take (csrc/mlp_plan.h; flag bit 6): True (every such
# scale), a set of npoint values (= layers), or {npoint: True | tuple of scale indices}.  Built in round 5 and left off then:
# the 8-phase next-fit scan of its plans cost what the saved rows gave.  Round 6 packs plans tight with a plain prefix sum (the
# plan's cost no longer depends on the granule size) and the one-launch layer kernels take a granule size PER SCALE.  Per 128
# frames (tools/stages_at.py), default / rings64 / dense(32):
#   layer 2 (npoint 1024), all scales: 499 -> 396-403 us | 945 -> 910 | a wash            -> ON
#   layer 1 (npoint 4096), scales 0 / 1 (the inner bands: 1-2 points per ball, 65 536 rows evaluated for 9 010 distinct at
#   8 rows): 325 -> 283 | 930 -> 930 | 394 -> 403                                         -> ON; with scale 2 as well: 261 | 987: off
#   layer 3 (npoint 512), scale 0: 622 -> 600 | 1 007 -> 1 018; all scales 590 | 1 050    -> off; layer 4: never taken
# {npoint: {scale index: 4 | 2}} asks for granules of TWO rows on a scale (flag bit 8; sixteen entries per tile): built and
# bit-identical, 41 % fewer rows on default frames for 0.6 % of the time, +80 us on rings64 (profiles/r06_granule2_ab.txt): off.
# The keys are the npoint values of configs/kitti/3dssd/3dssd.yaml; other networks keep 8 rows unless told otherwise.
MLP_GRANULE4 = {1024: True, 4096: (0, 1)}
GRID_BALL_QUERY_MIN_N = 1024   # round 5: the 1024-point frames of layer 3 through the grid too (120 -> 77 us per 128 frames; 512-point frames are faster brute force: 28 vs 42 us)
MLP_GEMM_CHAIN = False  # True: eligible fp16 scales (layer4) run as three large-tile GEMM launches (flags bit 4); measured slower
MAX_TRANSLATE_RANGE = (-3.0, -2.0, -3.0)
# F-FPS without the distance matrix (csrc/ffps_fly.hip) where the shape allows it (64 feature channels, 1024 / 2048 /
# 4096 points: layer 2 of 3dssd.yaml).  Several workgroups share a frame, so every call of a process must be on one
# stream at a time.  Opt-in for DIRECT single-stream callers only (SABackbone(ffps_fly=True)); the executor of pipeline.py
# runs that layer's sampler on alternating main streams and refuses a network built with it.
FFPS_FLY = False


_UNSUPPORTED = -3


def _dense(x, layer, relu):
    """tf_util.conv1d 1x1 (+ folded BN) (+ ReLU) on [..., K] -> [..., N]."""
    rows = x.numel() // layer.K
    y = torch.empty(x.shape[:-1] + (layer.N,), dtype=torch.float32, device=x.device)
    st = N.lib().sa_dense(rows, layer.K, layer.N, x.data_ptr(), layer.w.data_ptr(), layer.bias.data_ptr(),
                          1 if relu else 0, y.data_ptr(), N.current_stream())
    N.check(st, "dense")
    return y


def vote_layer(xyz, points, mlp_list, is_training, bn_decay, bn, scope, variables=None, max_translate_range=None):
    """layers_util.py:12-24.  Returns (xyz + clipped offsets, features, raw offsets).  max_translate_range: the
    reference's cfg.MODEL.MAX_TRANSLATE_RANGE (None: the module default above)."""
    vs = variables or W.default_variables()
    xyz = T.f32_cuda(xyz, "xyz")
    points = T.f32_cuda(points, "points")
    hidden = [vs.layer("%s/vote_layer_%d" % (scope, i), bn) for i, _channel in enumerate(mlp_list)]
    last = vs.layer(scope + "/vote_offsets", False)
    out = torch.empty_like(xyz)
    lo = MAX_TRANSLATE_RANGE if max_translate_range is None else tuple(max_translate_range)
    for layer in hidden[:-1]:
        points = _dense(points, layer, relu=True)
    if hidden:
        # the last hidden layer, the offset layer and the translation in one launch (three before)
        h = hidden[-1]
        rows = points.numel() // h.K
        feats = torch.empty(points.shape[:-1] + (h.N,), dtype=torch.float32, device=points.device)
        ctr_offsets = torch.empty(points.shape[:-1] + (3,), dtype=torch.float32, device=points.device)
        st = N.lib().sa_vote_tail(rows, h.K, h.N, points.data_ptr(), h.w.data_ptr(), h.bias.data_ptr(), last.w.data_ptr(),
                                  last.bias.data_ptr(), feats.data_ptr(), ctr_offsets.data_ptr(), xyz.data_ptr(),
                                  float(lo[0]), float(lo[1]), float(lo[2]), out.data_ptr(), N.current_stream())
        if st != _UNSUPPORTED:
            N.check(st, "vote_tail")
            return out, feats, ctr_offsets
    if hidden:
        points = _dense(points, hidden[-1], relu=True)
    ctr_offsets = _dense(points, last, relu=False)
    st = N.lib().sa_vote_translate(xyz.numel() // 3, xyz.data_ptr(), ctr_offsets.data_ptr(), float(lo[0]),
                                   float(lo[1]), float(lo[2]), out.data_ptr(), N.current_stream())
    N.check(st, "vote_translate")
    return out, points, ctr_offsets




def _ffps_into(npoint, xyz, points, start, end, out, col, ctr, matrix_only=False):
    """F-FPS on rows [start, end) of every frame: calc_square_dist(concat([xyz, feat])) +
    farthest_point_sample_with_distance (layers_util.py:94-96,102-104), written into out[:, col:col+npoint] with
    `start` added.  The range is read in place (no slice copy) and, when ctr = (tensor [b, total, 3]) is given, the
    picked points go straight into ctr[:, col:col+npoint] (the gather_point of :116-119).  Returns True when the
    centres were written."""
    b, n_all, _ = xyz.shape
    n = end - start
    c1 = points.shape[2]
    dev = xyz.device
    dist = torch.empty((b, n, n), dtype=torch.float32, device=dev)
    lib = N.lib()
    xp = xyz.data_ptr() + 4 * 3 * start
    pp = points.data_ptr() + 4 * c1 * start
    # packed form: the operand is laid out once in the matrix kernel's LDS image, tiles are staged by plain copies
    ws = torch.empty((lib.sa_calc_square_dist_ws_bytes(b, n, n, 3 + c1, 1) + 3) // 4, dtype=torch.float32, device=dev)
    keep = None
    st = lib.sa_calc_square_dist_self_ws(b, n, 3, c1, xp, n_all, pp, n_all, dist.data_ptr(), ws.data_ptr(),
                                         N.current_stream())
    if st == _UNSUPPORTED:                          # strided sources need the packed form: copy the slice instead
        keep = (xyz[:, start:end].contiguous(), points[:, start:end].contiguous())
        st = lib.sa_calc_square_dist_split_ws(b, n, n, 3, c1, keep[0].data_ptr(), keep[1].data_ptr(),
                                              keep[0].data_ptr(), keep[1].data_ptr(), dist.data_ptr(),
                                              ws.data_ptr(), N.current_stream())
    N.check(st, "calc_square_dist")
    if matrix_only:
        return dist
    temp = torch.empty((b, n), dtype=torch.float32, device=dev) if n > 16384 else None
    done = [False]

    def chain():
        tp = temp.data_ptr() if temp is not None else None
        if ctr is not None:
            st = lib.sa_fps_with_distance_ex2(b, n, npoint, dist.data_ptr(), tp, out.data_ptr() + 4 * col, out.shape[1],
                                              start, xp, 3 * n_all, ctr.data_ptr() + 4 * 3 * col, 3 * ctr.shape[1],
                                              N.current_stream())
            if st != _UNSUPPORTED:
                N.check(st, "farthest_point_sample_with_distance")
                done[0] = True
                return
        st = lib.sa_fps_with_distance_ex(b, n, npoint, dist.data_ptr(), tp, out.data_ptr() + 4 * col, out.shape[1],
                                         start, N.current_stream())
        N.check(st, "farthest_point_sample_with_distance")
    _run_chain(chain)
    return done[0]


def _run_chain(fn):
    """A long serial FPS kernel.  Mode 4: on the helper stream between two events (fork / join) instead of in line."""
    if DFPS_SIDE_STREAM != 4:
        return fn()
    main = torch.cuda.current_stream()
    side = _side_stream(main)
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        fn()
    ev2 = torch.cuda.Event()
    ev2.record(side)
    main.wait_event(ev2)


_SIDE_STREAMS = {}
_IDENTITY_IDX = {}


def _identity_idx(bs, start, cnt, dev):
    """[bs, cnt] int32 tensor of start .. start+cnt-1 in every row.  READ-ONLY: the tensor is shared between calls (it
    is the layer's `fps_idx` output when sampling is the identity, layers_util.py:92,100); a caller that wants to edit it
    must clone it.  Never cached from inside a hipGraph capture: a tensor created there lives in that graph's private
    pool and holds garbage until that graph replays."""
    key = (bs, start, cnt, str(dev))
    if key not in _IDENTITY_IDX:
        t = torch.arange(start, start + cnt, dtype=torch.int32, device=dev)[None].repeat(bs, 1).contiguous()
        if torch.cuda.is_current_stream_capturing():
            return t
        _IDENTITY_IDX[key] = t
    return _IDENTITY_IDX[key]


def _side_stream(main, which=0):
    """Helper streams per issuing stream: D-FPS runs there while the F-FPS chain (distance matrix +
    FPS on it) runs on the issuing stream -- the two halves of 'FS' / of a two-range layer are
    independent (layers_util.py:93-106)."""
    key = (main.device, main.cuda_stream, which)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=main.device)
    return _SIDE_STREAMS[key]


def _dfps_into(npoint, xyz, start, end, out, col, ctr, coop_capture=False):
    """D-FPS on rows [start, end) of every frame (layers_util.py:97,106), read in place; indices (+ start) into
    out[:, col:col+npoint], the picked points into ctr[:, col:col+npoint] when given.  Returns True when the centres
    were written.  coop_capture: sa_fps_ex3 flag bit 0 -- frames beyond one workgroup's capacity may launch the
    multi-workgroup sampler plainly under graph capture (the caller keeps all such launches on one stream)."""
    b, n_all, c = xyz.shape
    n = end - start
    dev = xyz.device
    done = [False]

    def chain():
        lib = N.lib()
        xp = xyz.data_ptr() + 4 * c * start
        if c == 3 and n <= 16384:                   # the register-resident kernels: no scratch, range read in place
            st = lib.sa_fps_ex2(b, n, c, npoint, xp, c * n_all, None, out.data_ptr() + 4 * col, out.shape[1], start,
                                ctr.data_ptr() + 4 * 3 * col if ctr is not None else None,
                                3 * ctr.shape[1] if ctr is not None else 0, N.current_stream())
            if st != _UNSUPPORTED:
                N.check(st, "farthest_point_sample")
                done[0] = ctr is not None
                return
        # frames that do not fit the register-resident kernels: dense copy of the range, scratch, separate gather
        src = xyz if (start == 0 and end == n_all) else xyz[:, start:end].contiguous()
        temp = torch.empty((b, n), dtype=torch.float32, device=dev)
        st = lib.sa_fps_ex3(b, n, c, npoint, src.data_ptr(), 0, temp.data_ptr(), out.data_ptr() + 4 * col, out.shape[1],
                            start, None, 0, 1 if coop_capture else 0, N.current_stream())
        N.check(st, "farthest_point_sample")
    _run_chain(chain)
    return done[0]


def sample_layer(xyz, points, fps_sample_range_list, fps_method_list, npoint_list, former_fps_idx, vote_ctr, radius_list,
                 side_mode=None, ffps_fly=None, coop_capture=False):
    """The sampling half of pointnet_sa_module_msg (layers_util.py:84-119): range slicing, D-FPS / F-FPS / FS / identity
    per range, index offsets, the centres.  Returns (fps_idx [B,m] int32, new_xyz [B,m,3], sliced_points or None).
    side_mode: DFPS_SIDE_STREAM for this call (None: the module default)."""
    side_mode = DFPS_SIDE_STREAM if side_mode is None else side_mode
    ffps_fly = FFPS_FLY if ffps_fly is None else ffps_fly
    bs, n_all, _ = xyz.shape
    dev = xyz.device

    # ---- sampling plan (layers_util.py:84-111): (kind, start, end, npoint) per non-empty range
    plan, last, total = [], 0, 0
    for fps_sample_range, fps_method, npoint in zip(fps_sample_range_list, fps_method_list, npoint_list):
        end = n_all if fps_sample_range == -1 else last + fps_sample_range   # tf.slice size -1
        if npoint == 0:                                                      # :87-89
            last += fps_sample_range
            continue
        rng_n = end - last
        if vote_ctr is not None:                                             # :90-92
            kind, cnt = "identity", vote_ctr.shape[1]
        elif fps_method == "FS":                                             # :93-98
            kind, cnt = "FS", 2 * npoint
        elif npoint == rng_n:                                                # :99-100
            kind, cnt = "identity", npoint
        elif fps_method == "F-FPS":                                          # :101-104
            kind, cnt = "F-FPS", npoint
        else:                                                                # :105-106
            kind, cnt = "D-FPS", npoint
        plan.append((kind, last, end, cnt))
        total += cnt
        last += fps_sample_range
    former_n = former_fps_idx.shape[1] if former_fps_idx is not None else 0
    only_identity = len(plan) == 1 and plan[0][0] == "identity" and former_n == 0
    if only_identity:
        # identity sampling (npoint == range size / vote centres): a cached constant index tensor, no kernel
        fps_idx = _identity_idx(bs, plan[0][1], plan[0][3], dev)
        plan_iter = []
    else:
        fps_idx = torch.empty((bs, total + former_n), dtype=torch.int32, device=dev)
        plan_iter = plan
    col = 0
    has_f = any(k in ("FS", "F-FPS") for k, _s, _e, _c in plan)
    has_d = any(k in ("FS", "D-FPS") for k, _s, _e, _c in plan)
    main = torch.cuda.current_stream()
    side = _side_stream(main) if (has_f and has_d and side_mode in (1, 2)) else None
    # The samplers read their range of xyz / points in place (frame stride = the whole tensor's) and write the picked
    # points themselves: no slice copies, no separate gather_point launch (:116-119) -- each launch of a step costs
    # 25-50 us of its span when 16 steps are in flight (tools/dispatch_chain.py).
    ctr_src = T.f32_cuda(vote_ctr, "vote_ctr") if vote_ctr is not None else xyz
    fuse_ctr = (vote_ctr is None and former_n == 0 and not only_identity and xyz.shape[2] == 3)
    new_xyz = torch.empty((bs, total, 3), dtype=torch.float32, device=dev) if fuse_ctr else None
    centres_ok = fuse_ctr
    work = []
    for kind, start, end, cnt in plan_iter:
        if kind == "identity":
            fps_idx[:, col:col + cnt] = torch.arange(start, start + cnt, dtype=torch.int32, device=dev)[None]
            if fuse_ctr:
                new_xyz[:, col:col + cnt] = xyz[:, start:start + cnt]
        else:
            work.append((kind, start, end, cnt, col))
        col += cnt

    def ffps_all():
        ok = True
        for kind, start, end, cnt, c0 in work:
            if kind in ("FS", "F-FPS"):
                ok = _ffps_into(cnt // 2 if kind == "FS" else cnt, xyz, points, start, end, fps_idx, c0, new_xyz) and ok
        return ok

    # ---- one matrix sampler + one coordinate sampler (an 'FS' range, or an F-FPS range and a D-FPS range): ONE launch
    #      for both (sa_fps_dual_ex, modes 5 / 6); anything else takes the per-sampler path below
    dual_done = f_handled = False
    if side_mode in (5, 6):
        fparts = [(s0, e0, (c // 2 if k == "FS" else c), c0) for k, s0, e0, c, c0 in work if k in ("FS", "F-FPS")]
        dparts = [(s0, e0, (c // 2 if k == "FS" else c), (c0 + c // 2 if k == "FS" else c0)) for k, s0, e0, c, c0 in work
                  if k in ("FS", "D-FPS")]
        if len(fparts) == 1 and len(dparts) == 1 and xyz.shape[2] == 3:
            (fs, fe, fm, fc), (ds, de, dm, dc) = fparts[0], dparts[0]
            lib = N.lib()
            cptr = (lambda col_: new_xyz.data_ptr() + 12 * col_) if new_xyz is not None else (lambda col_: None)
            cstr = 3 * new_xyz.shape[1] if new_xyz is not None else 0
            if ffps_fly and points.shape[2] == 64 and (fe - fs) in (1024, 2048, 4096):
                # the matrix is never built: every row a pick needs is computed on the fly (same picks, bit for bit)
                ws = torch.empty((int(lib.sa_ffps_fly_ws_bytes(bs, fe - fs)) + 7) // 8, dtype=torch.int64, device=dev)
                st = lib.sa_ffps_fly_ex(bs, fe - fs, 64, fm, xyz.data_ptr() + 12 * fs, 3 * n_all,
                                        points.data_ptr() + 4 * 64 * fs, 64 * n_all, ws.data_ptr(),
                                        fps_idx.data_ptr() + 4 * fc, fps_idx.shape[1], fs, cptr(fc), cstr, N.current_stream())
                if st != _UNSUPPORTED:
                    N.check(st, "ffps_fly")
                    f_handled = dual_done = True
                    ok_d = _dfps_into(dm, xyz, ds, de, fps_idx, dc, new_xyz, coop_capture)
                    centres_ok = centres_ok and new_xyz is not None and ok_d
                    work = []
        if not dual_done and len(fparts) == 1 and len(dparts) == 1 and xyz.shape[2] == 3:
            (fs, fe, fm, fc), (ds, de, dm, dc) = fparts[0], dparts[0]
            dist = _ffps_into(fm, xyz, points, fs, fe, fps_idx, fc, new_xyz, matrix_only=True)

            def dual():
                return lib.sa_fps_dual_ex(bs, fe - fs, fm, dist.data_ptr(), fps_idx.data_ptr() + 4 * fc, fps_idx.shape[1], fs,
                                          xyz.data_ptr() + 12 * fs, 3 * n_all, cptr(fc), cstr, de - ds, dm,
                                          xyz.data_ptr() + 12 * ds, 3 * n_all, fps_idx.data_ptr() + 4 * dc, fps_idx.shape[1],
                                          ds, cptr(dc), cstr, N.current_stream())
            if side_mode == 6:                       # on the helper stream between a fork and a join event
                hs = _side_stream(main)
                ev = torch.cuda.Event()
                ev.record(main)
                hs.wait_event(ev)
                with torch.cuda.stream(hs):
                    st = dual()
                ev2 = torch.cuda.Event()
                ev2.record(hs)
                main.wait_event(ev2)
            else:
                st = dual()
            f_handled = True
            if st != _UNSUPPORTED:
                N.check(st, "fps_dual")
                dual_done = True
                centres_ok = centres_ok and new_xyz is not None
                work = []
            else:                                           # sizes the dual kernel does not take: sampler by sampler
                tmp = torch.empty((bs, fe - fs), dtype=torch.float32, device=dev) if fe - fs > 16384 else None
                st = lib.sa_fps_with_distance_ex(bs, fe - fs, fm, dist.data_ptr(), tmp.data_ptr() if tmp is not None else None,
                                                 fps_idx.data_ptr() + 4 * fc, fps_idx.shape[1], fs, N.current_stream())
                N.check(st, "farthest_point_sample_with_distance")
                centres_ok = False
                work = [wk for wk in work if wk[0] == "D-FPS"] + [("D-FPS", s0, e0, c // 2, c0 + c // 2)
                                                                   for k, s0, e0, c, c0 in work if k == "FS"]
    if dual_done:
        side = None
    if side_mode in (2, 3):
        centres_ok = ffps_all() and centres_ok
    if side is not None:
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
    for kind, start, end, cnt, c0 in work:                                  # D-FPS parts (:97,106)
        if kind in ("FS", "D-FPS"):
            d_n = cnt // 2 if kind == "FS" else cnt
            d_col = c0 + d_n if kind == "FS" else c0                        # 'FS': [F-FPS idx || D-FPS idx] (:96-98)
            if side is not None:
                with torch.cuda.stream(side):
                    centres_ok = _dfps_into(d_n, xyz, start, end, fps_idx, d_col, new_xyz, coop_capture) and centres_ok
            else:
                centres_ok = _dfps_into(d_n, xyz, start, end, fps_idx, d_col, new_xyz, coop_capture) and centres_ok
    if side_mode not in (2, 3) and not f_handled:                    # F-FPS parts (:94-96,102-104)
        centres_ok = ffps_all() and centres_ok
    if side is not None:
        ev = torch.cuda.Event()
        ev.record(side)
        main.wait_event(ev)
    if former_fps_idx is not None:                                          # :112-113
        fps_idx[:, col:] = T.i32_cuda(former_fps_idx, "former_fps_idx")

    sliced_points = None
    if centres_ok:
        pass                                                                # written by the samplers
    elif only_identity and plan[0][1] == 0 and plan[0][3] == ctr_src.shape[1]:
        new_xyz = ctr_src                                                   # gather with the identity: the tensor itself
    elif only_identity and ctr_src.shape[2] == 3:
        # identity sampling of a range: the gathers of :116-119 (and :186-187 when the layer has no radius scales) are
        # block copies -- one launch for both
        s0, cnt0 = plan[0][1], plan[0][3]
        new_xyz = torch.empty((bs, cnt0, 3), dtype=torch.float32, device=dev)
        jobs = [(ctr_src[:, s0:s0 + cnt0], new_xyz, bs, cnt0, 3)]
        if len(radius_list) == 0:
            sliced_points = torch.empty((bs, cnt0, points.shape[2]), dtype=torch.float32, device=dev)
            jobs.append((points[:, s0:s0 + cnt0], sliced_points, bs, cnt0, points.shape[2]))
        N.copy_blocks(jobs)
    else:
        new_xyz = gather_point(ctr_src, fps_idx)                            # :116-119
    return fps_idx, new_xyz, sliced_points


def pointnet_sa_module_msg(xyz, points, radius_list, nsample_list, mlp_list, is_training, bn_decay, bn,
                           fps_sample_range_list, fps_method_list, npoint_list, former_fps_idx,
                           use_attention, scope, dilated_group, vote_ctr=None, aggregation_channel=None,
                           debugging=False, epsilon=1e-5, variables=None, aggregation_sa_feature=None, presampled=None,
                           dfps_side_stream=None, ffps_fly=None, coop_capture=False):
    """layers_util.py:59-189.  xyz (B,n,3), points (B,n,C) -> new_xyz (B,m,3), new_points (B,m,C'),
    fps_idx (B,m) int32.  aggregation_sa_feature: cfg.MODEL.NETWORK.AGGREGATION_SA_FEATURE (None: the module default);
    presampled = (fps_idx, new_xyz, sliced_points) of an earlier `sample_layer` call with the same arguments (the staged
    executor of pipeline.py runs the sampling half on its own stream)."""
    T.require(not use_attention, "use_attention (query_ball_point_withidx) is outside the 3DSSD SA path")
    vs = variables or W.default_variables()
    xyz = T.f32_cuda(xyz, "xyz")
    points = T.f32_cuda(points, "points")
    bs, n_all, _ = xyz.shape
    dev = xyz.device

    # ---- sampling (layers_util.py:84-119)
    if presampled is not None:
        fps_idx, new_xyz, sliced_points = presampled
    else:
        fps_idx, new_xyz, sliced_points = sample_layer(xyz, points, fps_sample_range_list, fps_method_list, npoint_list,
                                                       former_fps_idx, vote_ctr, radius_list, dfps_side_stream, ffps_fly,
                                                       coop_capture)
    m = new_xyz.shape[1]
    lib = N.lib()
    stream = N.current_stream()

    nscale = len(radius_list)
    if nscale > 0:
        # ---- all bands of the layer in one ball-query pass (:134-147)
        idx_list = [torch.empty((bs, m, int(ns)), dtype=torch.int32, device=dev) for ns in nsample_list]
        cnt_list = [torch.empty((bs, m), dtype=torch.int32, device=dev) for _ in nsample_list]
        rmax = (ctypes.c_float * nscale)(*[float(r) for r in radius_list])
        rmin = (ctypes.c_float * nscale)(*[0.0 if i == 0 else float(radius_list[i - 1]) for i in range(nscale)])
        nsa = (ctypes.c_int * nscale)(*[int(v) for v in nsample_list])
        idxp = (ctypes.c_void_p * nscale)(*[t.data_ptr() for t in idx_list])
        cntp = (ctypes.c_void_p * nscale)(*[t.data_ptr() for t in cnt_list])
        if n_all >= GRID_BALL_QUERY_MIN_N and nscale <= 4:
            # large frames: per-frame x-z grid, candidates from the 3 x 3 cells around each centre (same outputs)
            ws = torch.empty((lib.sa_query_ball_point_grid_ws_bytes(bs, n_all, m) + 3) // 4, dtype=torch.int32, device=dev)
            st = lib.sa_query_ball_point_grid(bs, n_all, m, nscale, rmin, rmax, nsa, 1 if dilated_group else 0,
                                              xyz.data_ptr(), new_xyz.data_ptr(), idxp, cntp, ws.data_ptr(), stream)
        else:
            st = lib.sa_query_ball_point_multi(bs, n_all, m, nscale, rmin, rmax, nsa, 1 if dilated_group else 0,
                                               xyz.data_ptr(), new_xyz.data_ptr(), idxp, cntp, stream)
        N.check(st, "query_ball_point")
        # ---- per scale: fused mask/group/concat/MLP/max/mask (:157-181) into the concat buffer (:183)
        layers = [vs.scale(["%s/conv%d_%d" % (scope, i, j) for j in range(len(mlp_list[i]))], bn)
                  for i in range(nscale)]
        ctot = sum(ls[-1].N for ls in layers)
        new_points_concat = torch.empty((bs, m, ctot), dtype=torch.float32, device=dev)
        c_feat = points.shape[2]
        # row plans of all scales in one launch: only the distinct rows of every ball are evaluated (mlp_plan.h)
        # (+ the packed hidden activations of the opt-in GEMM chain, csrc/mlp_gemm.hip, when MLP_GEMM_CHAIN is set)
        plans = [N.mlp_plan_ws(bs, m, int(ns), dev, c_feat, [c_feat + 3] + [l.N for l in layers[i]]
                               if (MLP_GEMM_CHAIN and W.scale_flags(layers[i])) else None)
                 for i, ns in enumerate(nsample_list)]
        offs, acc = [], 0
        for ls in layers:
            offs.append(acc)
            acc += ls[-1].N
        have_plans = nscale <= 4
        base_flags = [MLP_PLAN_FLAGS | (16 if MLP_GEMM_CHAIN else 0) | (2 if have_plans else 0) | W.scale_flags(ls) for ls in layers]
        g4 = MLP_GRANULE4
        g4_rows = None                                       # {scale index: rows per granule (4 or 2)}; None: 4 rows on every scale
        if isinstance(g4, dict):
            v = g4.get(m)
            g4 = v is not None
            if isinstance(v, dict):
                g4_rows = dict(v)
            elif v is not None and v is not True:
                g4_rows = {int(i): 4 for i in v}
        if have_plans and (g4 is True or (g4 and m in g4)):      # True, a set of npoint values (layers), or {npoint: True | scale indices | {scale: rows}}
            # granule size per scale where a row-wave kernel will take the scale (the library says which), else 8
            for i, ls in enumerate(layers):
                if g4_rows is not None and i not in g4_rows:
                    continue
                d_ = (ctypes.c_int * (len(ls) + 1))(*([c_feat + 3] + [l.N for l in ls]))
                wp_ = (ctypes.c_void_p * len(ls))(*[l.w.data_ptr() for l in ls])
                if lib.sa_group_mlp_granule_rows(bs, n_all, m, int(nsample_list[i]), c_feat, len(ls), d_, wp_, plans[i][1], base_flags[i]) == 4:
                    base_flags[i] |= 256 if (g4_rows is not None and g4_rows[i] == 2) else 64
            if nscale == 3 and g4_rows is None and len({f & 64 for f in base_flags}) != 1:      # "all scales" asked for, not all possible: none
                base_flags = [f & ~64 for f in base_flags]
        if have_plans:
            st = lib.sa_group_mlp_plan2(bs, m, nscale, nsa, cntp, (ctypes.c_void_p * nscale)(*[p[0].data_ptr() for p in plans]),
                                        new_points_concat.data_ptr(), ctot, (ctypes.c_int * nscale)(*offs),
                                        (ctypes.c_int * nscale)(*[ls[-1].N for ls in layers]), MLP_PLAN_FLAGS,
                                        (ctypes.c_int * nscale)(*base_flags), stream)
            N.check(st, "group_mlp_plan")
        # ---- the grouped MLPs of all scales: ONE C-ABI call per layer (one launch for the three-scale layers of the
        #      reference configuration: the scales are independent, every launch costs ~2 us of throughput)
        nls = {len(ls) for ls in layers}
        live = list(range(nscale))
        if len(nls) == 1:
            nl = nls.pop()
            k = len(live)
            dims_a = (ctypes.c_int * (k * (nl + 1)))(*[v for i in live for v in ([c_feat + 3] + [l.N for l in layers[i]])])
            wp = (ctypes.c_void_p * (k * nl))(*[l.w.data_ptr() for i in live for l in layers[i]])
            bp = (ctypes.c_void_p * (k * nl))(*[l.bias.data_ptr() for i in live for l in layers[i]])
            st = lib.sa_group_mlp_max_layer(
                k, bs, n_all, m, (ctypes.c_int * k)(*[int(nsample_list[i]) for i in live]), c_feat, xyz.data_ptr(),
                points.data_ptr(), new_xyz.data_ptr(), (ctypes.c_void_p * k)(*[idx_list[i].data_ptr() for i in live]),
                (ctypes.c_void_p * k)(*[cnt_list[i].data_ptr() for i in live]), nl, dims_a, wp, bp,
                new_points_concat.data_ptr(), ctot, (ctypes.c_int * k)(*[offs[i] for i in live]),
                (ctypes.c_void_p * k)(*[plans[i][0].data_ptr() for i in live]),
                (ctypes.c_ulong * k)(*[plans[i][1] for i in live]),
                (ctypes.c_int * k)(*[base_flags[i] for i in live]),
                vs.overflow.data_ptr(), stream)
            N.check(st, "group_mlp_max_layer")
        else:                                                               # scales of different depth: one by one
            for i in live:
                ls = layers[i]
                nl = len(ls)
                dims = (ctypes.c_int * (nl + 1))(*([c_feat + 3] + [l.N for l in ls]))
                wp = (ctypes.c_void_p * nl)(*[l.w.data_ptr() for l in ls])
                bp = (ctypes.c_void_p * nl)(*[l.bias.data_ptr() for l in ls])
                st = lib.sa_group_mlp_max(bs, n_all, m, int(nsample_list[i]), c_feat, xyz.data_ptr(),
                                          points.data_ptr(), new_xyz.data_ptr(), idx_list[i].data_ptr(),
                                          cnt_list[i].data_ptr(), nl, dims, wp, bp,
                                          new_points_concat.data_ptr(), ctot, offs[i], plans[i][0].data_ptr(), plans[i][1],
                                          base_flags[i], vs.overflow.data_ptr(), stream)
                N.check(st, "group_mlp_max")
        if PLAN_LOG is not None:                                            # bench.py: rows evaluated per scale
            for i in live:
                nl = len(layers[i])
                d_ = [c_feat + 3] + [l.N for l in layers[i]]
                PLAN_LOG.append((bs, m, int(nsample_list[i]), sum(d_[j] * d_[j + 1] for j in range(nl)), plans[i][0]))
        if CONCAT_LOG is not None:                                          # tests: the pooled outputs of the scales before the aggregation layer
            CONCAT_LOG.append((scope, new_points_concat, list(offs), [ls[-1].N for ls in layers], [ls[0].precision for ls in layers]))
        if (AGGREGATION_SA_FEATURE if aggregation_sa_feature is None else aggregation_sa_feature):   # :184-185
            agg = vs.layer(scope + "/ensemble", bn)
            T.require(agg.N == aggregation_channel, "aggregation_channel does not match the ensemble weights")
            new_points_concat = _dense(new_points_concat, agg, relu=True)
    else:
        new_points_concat = sliced_points if sliced_points is not None else gather_point(points, fps_idx)   # :186-187
    return new_xyz, new_points_concat, fps_idx
