"""Seeded synthetic inputs and weights for the SA backbone.

There is no dataset and no checkpoint in this environment, so the benchmark and the parity tests
use KITTI-shape synthetic frames and random-init weights.  The generator is fixed here (seeds are
part of the contract) so that the CPU oracle and the HIP path see identical tensors.
"""
import numpy as np

FRAME_SEED = 20260925
WEIGHT_SEED = 1234


def kitti_like_frame(frame_id, n=16384, dup_fraction=0.0):
    """One [n,4] fp32 frame: LiDAR-like polar density inside POINT_CLOUD_RANGE
    (configs/kitti/3dssd/3dssd.yaml:3), channel 3 = intensity in [0,1].

    dup_fraction > 0 overwrites that fraction of the tail rows with copies of earlier rows, which
    mirrors the with-replacement padding of lib/dataset/dataloader/kitti_dataloader.py:142-147 and
    exercises the FPS tie-break.
    """
    rng = np.random.default_rng(FRAME_SEED + int(frame_id))
    m = int(1.25 * n) + 64
    r = rng.uniform(2.0, 70.0, m).astype(np.float32)
    th = rng.uniform(-np.pi / 4, np.pi / 4, m).astype(np.float32)
    x = (r * np.sin(th)).astype(np.float32)
    z = (r * np.cos(th)).astype(np.float32)
    y = np.clip(1.6 - np.abs(rng.normal(0.0, 0.5, m)), -5.0, 3.0).astype(np.float32)
    inten = rng.uniform(0.0, 1.0, m).astype(np.float32)
    keep = np.abs(x) <= 40.0
    pts = np.stack([x, y, z, inten], 1)[keep][:n]
    assert pts.shape[0] == n, "synthetic frame generator ran short"
    if dup_fraction > 0:
        k = int(n * dup_fraction)
        src = rng.integers(0, n - k, k)
        pts[n - k:] = pts[src]
    return np.ascontiguousarray(pts, np.float32)


def kitti_like_batch(batch, n=16384, first_frame=0, dup_fraction=0.0):
    return np.stack([kitti_like_frame(first_frame + i, n, dup_fraction) for i in range(batch)])


# Uniform-density WORST case of the distinct-row plan and of the culled samplers (bench.py --data dense): n points drawn
# uniformly from a 3 m x 1 m x 3 m box = 1 820 points per cubic metre at n = 16384, i.e. ~61 points inside a radius-0.2
# ball, so that every ball of every layer is full (nsample distinct rows, evaluated_frac -> 1), every ball-query band
# overflows its candidate lists, and a D-FPS pick touches many buckets.  Not a LiDAR frame: a sensitivity bound.
DENSE_BOX = ((-1.5, 1.5), (0.6, 1.6), (10.0, 13.0))


def dense_frame(frame_id, n=16384):
    rng = np.random.default_rng(FRAME_SEED + 7919 + int(frame_id))
    pts = np.empty((n, 4), np.float32)
    for a, (lo, hi) in enumerate(DENSE_BOX):
        pts[:, a] = rng.uniform(lo, hi, n).astype(np.float32)
    pts[:, 3] = rng.uniform(0.0, 1.0, n).astype(np.float32)
    return pts


# A simulated 64-beam sweep (bench.py --data rings64; VERDICT r3: "the generator that decides the headline has no
# beam / ring structure").  Not real data -- there is none here -- but the structure real KITTI frames have and the
# polar generator lacks: points lie on 64 scan rings (HDL-64E elevations +2 .. -24.8 deg, 0.18 deg azimuth steps at 10 Hz),
# the rings of the downward beams are arcs on a ground plane 1.65 m below the sensor that crowd together with range
# (ring spacing ~ r^2 / h), vehicles and facades return dense vertical stripes, upward beams mostly return nothing, and
# the frame is then cropped to POINT_CLOUD_RANGE / the camera's field of view and re-sampled to n points exactly as
# the loader does (kitti_dataloader.py:137-151: choice without replacement, or all points + a with-replacement pad).
RINGS = 64
RINGS_ELEV_DEG = (2.0, -24.8)
RINGS_AZ_STEP_DEG = 0.18            # 64 beams x ~2000 azimuth steps per turn at 10 Hz (~1.3 M points/s)
RINGS_FOV_DEG = 41.0            # half angle of the image crop (get_point_filter_in_image, kitti_dataloader.py:192)
SENSOR_HEIGHT = 1.65            # camera-rect y of the ground plane (y points down)


def _ray_boxes(d, boxes):
    """Nearest hit of the rays (origin 0, directions d [R,3]) with axis-aligned boxes [K,6] (lo xyz, hi xyz): (t, k)."""
    t_best = np.full(d.shape[0], np.inf, np.float32)
    k_best = np.full(d.shape[0], -1, np.int32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (1.0 / d).astype(np.float32)
    for k, bx in enumerate(boxes):
        t0 = bx[None, :3] * inv
        t1 = bx[None, 3:] * inv
        tn = np.nanmax(np.minimum(t0, t1), 1)
        tf = np.nanmin(np.maximum(t0, t1), 1)
        hit = (tn <= tf) & (tn > 0.5) & (tn < t_best)
        t_best = np.where(hit, tn, t_best)
        k_best = np.where(hit, k, k_best)
    return t_best, k_best


def rings64_frame(frame_id, n=16384):
    rng = np.random.default_rng(FRAME_SEED + 104729 + int(frame_id))
    elev = np.deg2rad(np.linspace(RINGS_ELEV_DEG[0], RINGS_ELEV_DEG[1], RINGS)).astype(np.float32)
    az = np.deg2rad(np.arange(-RINGS_FOV_DEG, RINGS_FOV_DEG, RINGS_AZ_STEP_DEG)).astype(np.float32)
    e, a = np.meshgrid(elev, az, indexing="ij")
    e, a = e.ravel(), a.ravel() + rng.normal(0.0, 2e-4, e.size).astype(np.float32)        # encoder jitter
    d = np.stack([np.cos(e) * np.sin(a), -np.sin(e), np.cos(e) * np.cos(a)], 1).astype(np.float32)   # x right, y down, z forward
    # scene: ground plane, 8-20 vehicles, a few facades / fences along the road, some poles
    boxes = []
    for _ in range(int(rng.integers(8, 21))):
        l, w, h = rng.uniform(3.4, 4.8), rng.uniform(1.5, 1.9), rng.uniform(1.4, 1.8)
        if rng.uniform() < 0.3:
            l, w = w, l                                                                    # parked across
        cx, cz = rng.uniform(-18, 18), rng.uniform(6, 62)
        boxes.append([cx - w / 2, SENSOR_HEIGHT - h, cz - l / 2, cx + w / 2, SENSOR_HEIGHT, cz + l / 2])
    for _ in range(int(rng.integers(2, 7))):
        side = rng.choice([-1.0, 1.0])
        x0 = side * rng.uniform(7, 22)
        z0, ln, h = rng.uniform(0, 50), rng.uniform(8, 40), rng.uniform(1.2, 7.0)
        boxes.append([min(x0, x0 + side * 0.5), SENSOR_HEIGHT - h, z0, max(x0, x0 + side * 0.5), SENSOR_HEIGHT, z0 + ln])
    for _ in range(int(rng.integers(4, 12))):
        cx, cz, h = rng.uniform(-15, 15), rng.uniform(5, 60), rng.uniform(2.5, 6.0)
        boxes.append([cx - 0.12, SENSOR_HEIGHT - h, cz - 0.12, cx + 0.12, SENSOR_HEIGHT, cz + 0.12])
    boxes = np.asarray(boxes, np.float32)
    t_box, k_box = _ray_boxes(d, boxes)
    with np.errstate(divide="ignore"):
        t_gnd = np.where(d[:, 1] > 1e-4, SENSOR_HEIGHT / d[:, 1], np.inf).astype(np.float32)
    t = np.minimum(t_box, t_gnd)
    ok = np.isfinite(t) & (t < 80.0) & (rng.uniform(0, 1, t.size) > 0.08)                   # 8 % drop-outs
    t = (t + rng.normal(0.0, 0.02, t.size).astype(np.float32))[ok]                          # range noise
    on_box = (t_box <= t_gnd)[ok]
    pts = d[ok] * t[:, None]
    inten = np.where(on_box, rng.uniform(0.2, 0.9, t.size), rng.uniform(0.0, 0.4, t.size)).astype(np.float32)
    pts = np.concatenate([pts, inten[:, None]], 1).astype(np.float32)
    keep = (np.abs(pts[:, 0]) <= 40.0) & (pts[:, 1] >= -5.0) & (pts[:, 1] <= 3.0) & (pts[:, 2] >= 0.0) & (pts[:, 2] <= 70.0)
    pts = pts[keep]
    m = pts.shape[0]
    if m >= n:                                                                              # kitti_dataloader.py:140-141
        sel = rng.choice(m, n, replace=False)
    else:                                                                                   # :142-147
        sel = np.concatenate([rng.choice(m, m, replace=False), rng.choice(m, n - m, replace=True)])
    return np.ascontiguousarray(pts[sel], np.float32)


DATA_VARIANTS = ("default", "dup10", "dense", "rings64")


def frame_of(variant, frame_id, n=16384):
    """One frame of a bench.py --data variant: default = kitti_like_frame, dup10 = the same with 10 % of the rows
    duplicated (the loader's with-replacement padding, kitti_dataloader.py:142-147; SURVEY.md 8d "KITTI-padded"),
    dense = dense_frame, rings64 = rings64_frame (a simulated 64-beam sweep)."""
    if variant == "default":
        return kitti_like_frame(frame_id, n)
    if variant == "dup10":
        return kitti_like_frame(frame_id, n, dup_fraction=0.1)
    if variant == "dense":
        return dense_frame(frame_id, n)
    if variant == "rings64":
        return rings64_frame(frame_id, n)
    raise ValueError("unknown data variant %r (one of %s)" % (variant, ", ".join(DATA_VARIANTS)))


def _xavier(rng, cin, cout):
    lim = np.sqrt(6.0 / (cin + cout))  # tf.contrib.layers.xavier_initializer, tf_util.py:41
    return rng.uniform(-lim, lim, (cin, cout)).astype(np.float32)


def _bn(rng, c, p, scope):
    p[scope + "/bn/gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    p[scope + "/bn/beta"] = rng.normal(0.0, 0.1, c).astype(np.float32)
    p[scope + "/bn/moving_mean"] = rng.normal(0.0, 0.1, c).astype(np.float32)
    p[scope + "/bn/moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)


def random_backbone_params(arch, in_feature_channels=1, seed=WEIGHT_SEED, aggregation=True):
    """Random-init parameters for the SA_Layer / Vote_Layer rows of `arch`, keyed by the
    reference's TF variable names (lib/utils/layers_util.py:17-19,175,185;
    lib/utils/tf_util.py:96,111,439-442): <scope>/conv<i>_<j>/{weights,biases,bn/*},
    <scope>/ensemble/..., <scope>/vote_layer_<i>/..., <scope>/vote_offsets/...
    conv2d kernels are [1,1,cin,cout], conv1d kernels [1,cin,cout]; biases start at zero."""
    rng = np.random.default_rng(seed)
    p = {}
    feat_ch = [in_feature_channels]
    for row in arch:
        feature_index, radius_list, mlp_list, bn = row[1], row[2], row[4], row[5]
        layer_type, scope, agg = row[11], row[12], row[15]
        cin_feat = feat_ch[feature_index[0]]
        if layer_type == "SA_Layer":
            if radius_list:
                outs = 0
                for i, mlp in enumerate(mlp_list):
                    cin = cin_feat + 3
                    for j, cout in enumerate(mlp):
                        s = "%s/conv%d_%d" % (scope, i, j)
                        p[s + "/weights"] = _xavier(rng, cin, cout).reshape(1, 1, cin, cout)
                        p[s + "/biases"] = np.zeros(cout, np.float32)
                        if bn:
                            _bn(rng, cout, p, s)
                        cin = cout
                    outs += cin
                if aggregation:
                    s = scope + "/ensemble"
                    p[s + "/weights"] = _xavier(rng, outs, agg).reshape(1, outs, agg)
                    p[s + "/biases"] = np.zeros(agg, np.float32)
                    if bn:
                        _bn(rng, agg, p, s)
                    outs = agg
                feat_ch.append(outs)
            else:
                feat_ch.append(cin_feat)
        elif layer_type == "Vote_Layer":
            cin = cin_feat
            for i, cout in enumerate(mlp_list):
                s = "%s/vote_layer_%d" % (scope, i)
                p[s + "/weights"] = _xavier(rng, cin, cout).reshape(1, cin, cout)
                p[s + "/biases"] = np.zeros(cout, np.float32)
                if bn:
                    _bn(rng, cout, p, s)
                cin = cout
            s = scope + "/vote_offsets"
            p[s + "/weights"] = _xavier(rng, cin, 3).reshape(1, cin, 3)
            p[s + "/biases"] = np.zeros(3, np.float32)
            feat_ch.append(cin)
        else:
            raise NotImplementedError(layer_type)
    return p


def random_head_params(in_channels=512, cls_channel=1, angle_cls_num=12, mlp_list=(128,), scope="", seed=WEIGHT_SEED + 1,
                       params=None):
    """Random-init 'Det' head (lib/modeling/head_builder.py:97-98, lib/utils/head_util.py:32-38): conv1d_<i>,
    pred_cls_base, pred_cls, pred_reg_base, pred_reg under `scope` (empty for 3dssd.yaml:68)."""
    rng = np.random.default_rng(seed)
    p = {} if params is None else params
    pre = scope + "/" if scope else ""

    def conv(name, cin, cout, bn):
        p[pre + name + "/weights"] = _xavier(rng, cin, cout).reshape(1, cin, cout)
        p[pre + name + "/biases"] = rng.normal(0.0, 0.05, cout).astype(np.float32)
        if bn:
            _bn(rng, cout, p, pre + name)

    cin = in_channels
    for i, ch in enumerate(mlp_list):
        conv("conv1d_%d" % i, cin, ch, True)
        cin = ch
    conv("pred_cls_base", cin, 128, True)
    conv("pred_cls", 128, cls_channel, False)
    conv("pred_reg_base", cin, 128, True)
    conv("pred_reg", 128, 6 + 2 * angle_cls_num, False)
    return p
