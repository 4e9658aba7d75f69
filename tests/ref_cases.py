"""Case catalogue for pinning the oracle AND the HIP kernels to the reference's own device code
(oracle/_ref/libtf_ops_ref_fma.so, see tests/ref_gpu.py and oracle/Makefile `ref_gpu`).

Every case is (operator name, scalar args, numpy array args) in the reference's Python-API order, built from fixed
seeds.  Three consumers:
  * tests/test_ref_pin_gpu.py (-m gpu): reference library vs oracle vs HIP path, live, bit for bit;
  * tests/golden/make_golden_ref_gpu.py (run on the GPU box): writes the reference library's outputs to
    tests/golden/ref_gpu_pin.npz (SHA-1 of every output, the arrays themselves when small);
  * tests/test_ref_golden_cpu.py (CPU suite): the oracle must reproduce that committed fixture.
Shapes follow BASELINE.json configs[0] / configs[1] / configs[2] / configs[4] and the edge cases SURVEY.md 8c /
Appendix C name: 10 % duplicated rows, lattice points (ties, distances exactly on a radius), empty balls, points a
few ulps either side of a radius (where the FMA policy decides membership), idx == -1.
"""
import hashlib
import importlib

import numpy as np

f32 = np.float32


def _syn():
    return importlib.import_module("3dssd_amd.synthetic")


def sha1(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha1(a.tobytes() + str(a.dtype).encode() + str(a.shape).encode()).hexdigest()


# ------------------------------------------------------------------------------------------------ input builders
def lattice(n, step=0.5, seed=0, side=24):
    """n distinct points of a cubic lattice (exactly representable coordinates): every distance is exact, equal
    distances abound -> FPS picks are decided by the (k mod 1024, k) tie rule, radius tests by strictness."""
    rng = np.random.default_rng(seed)
    ids = rng.permutation(side ** 3)[:n]
    p = np.stack([ids % side, (ids // side) % side, ids // (side * side)], 1).astype(f32) * f32(step)
    return p[None]


def boundary_cloud(radii, q=96, per=40, seed=0):
    """q query points; for every radius, `per` dataset points per query placed at distance == radius up to the
    rounding of their coordinates, i.e. a few ulps either side of the threshold: here the three candidate
    arithmetics of tf_grouping_g.cu:243/336 (fused chain, no FMA, gfx950 packed form) disagree."""
    rng = np.random.default_rng(seed)
    ctr = np.stack([rng.uniform(-40, 40, q), rng.uniform(-3, 2, q), rng.uniform(0, 70, q)], 1).astype(f32)
    pts = [ctr.copy()]                                     # every centre is itself a dataset point (d == 0)
    for r in radii:
        u = rng.normal(0, 1, (q, per, 3))
        u /= np.linalg.norm(u, axis=2, keepdims=True)
        pts.append((ctr[:, None, :].astype(np.float64) + np.float64(f32(r)) * u).astype(f32).reshape(-1, 3))
    xyz = np.concatenate(pts, 0)
    xyz = xyz[rng.permutation(len(xyz))]
    return xyz[None].copy(), ctr[None].copy()


def _dups(p, frac, seed):
    rng = np.random.default_rng(seed)
    p = p.copy()
    n = p.shape[1]
    k = int(n * frac)
    for i in range(p.shape[0]):
        p[i, n - k:] = p[i, rng.integers(0, n - k, k)]
    return p


def _boxes_on_points(rng, xyz, m):
    b, n, _ = xyz.shape
    c = xyz[np.arange(b)[:, None], rng.integers(0, n, (b, m))]
    lhw = rng.uniform(0.5, 5.0, (b, m, 3))
    ry = rng.uniform(-np.pi, np.pi, (b, m, 1))
    boxes = np.concatenate([c[:, :, :1], c[:, :, 1:2] + lhw[:, :, 1:2] / 2, c[:, :, 2:3], lhw, ry], 2)
    return boxes.astype(f32)


# --------------------------------------------------------------------------------------------------- the catalogue
def sa_cases():
    """name -> (op, scalars, arrays)."""
    syn = _syn()
    rng = np.random.default_rng(77)
    C = {}
    # ---- a1 D-FPS / generic-channel FPS (tf_sampling_g.cu:123-178)
    C["fps_cfg0_4096_512"] = ("farthest_point_sample", (512,), (rng.uniform(0, 1, (1, 4096, 3)).astype(f32),))
    kit = syn.kitti_like_batch(2, n=16384)
    C["fps_cfg1_16384_4096"] = ("farthest_point_sample", (4096,), (kit[:, :, :3].copy(),))
    C["fps_cfg1_dup10"] = ("farthest_point_sample", (4096,),
                           (syn.kitti_like_batch(1, n=16384, first_frame=5, dup_fraction=0.1)[:, :, :3].copy(),))
    C["fps_lattice_ties"] = ("farthest_point_sample", (1024,), (lattice(8192, seed=1),))
    C["fps_all_identical"] = ("farthest_point_sample", (50,), (np.ones((2, 3000, 3), f32),))
    C["fps_small_ragged"] = ("farthest_point_sample", (5,), (rng.normal(0, 1, (3, 7, 3)).astype(f32),))
    C["fps_n_1000"] = ("farthest_point_sample", (1000,), (rng.normal(0, 3, (2, 1000, 3)).astype(f32),))
    C["fps_c67_4096_512"] = ("farthest_point_sample", (512,), (rng.normal(0, 1, (2, 4096, 67)).astype(f32),))
    C["fps_c131_512_256"] = ("farthest_point_sample", (256,), (rng.normal(0, 1, (2, 512, 131)).astype(f32),))
    fma = np.array([[[-0.6892402172088623, -0.11729754507541656, -0.16030718386173248],
                     [0.6565782427787781, 0.3208279013633728, -0.18391664326190948],
                     [-0.25111478567123413, -0.14090700447559357, 1.1855113506317139]],
                    [[0.707705020904541, -0.8554568886756897, 0.46334177255630493],
                     [-0.8842937350273132, 0.019086552783846855, 0.24701596796512604],
                     [1.5822484493255615, -1.0717827081680298, -1.1286571025848389]]], f32)
    C["fps_fma_sensitive"] = ("farthest_point_sample", (2,), (fma,))       # tests/test_oracle_kat.py:test_fps_fma_chain
    # ---- a2 F-FPS on a matrix (tf_sampling_g.cu:180-230)
    a = rng.normal(0, 1, (2, 300, 300)).astype(f32)                        # arbitrary (also negative) entries
    C["fpsdist_random_matrix"] = ("farthest_point_sample_with_distance", (100,), (a,))
    pts = rng.normal(0, 1, (1, 1024, 67)).astype(np.float64)
    d = ((pts[:, :, None, :] - pts[:, None, :, :]) ** 2).sum(-1).astype(f32)
    C["fpsdist_sq_matrix_1024"] = ("farthest_point_sample_with_distance", (256,), (d,))
    sym = np.round(rng.uniform(0, 4, (1, 640, 640))).astype(f32)            # few distinct values -> ties
    C["fpsdist_ties"] = ("farthest_point_sample_with_distance", (64,), (sym,))
    # ---- a4 gather_point (tf_sampling_g.cu:320-331)
    C["gather_c3"] = ("gather_point", (), (kit[:, :4096, :3].copy(), rng.integers(0, 4096, (2, 777)).astype(np.int32)))
    C["gather_c256"] = ("gather_point", (), (rng.normal(0, 1, (2, 512, 256)).astype(f32),
                                             rng.integers(0, 512, (2, 256)).astype(np.int32)))
    # ---- a5 query_ball_point_dilated (tf_grouping_g.cu:308-357): the bands of 3dssd.yaml:46-55
    x1 = kit[:1, :, :3].copy()
    ctr1 = x1[:, ::4].copy()
    for lo, hi, ns in ((0.0, 0.2, 32), (0.2, 0.4, 32), (0.4, 0.8, 64)):
        C["ballD_L1_%g_%g" % (lo, hi)] = ("query_ball_point_dilated", (lo, hi, ns), (x1, ctr1))
    x2 = kit[1:2, ::4, :3].copy()
    ctr2 = x2[:, ::4].copy()
    for lo, hi, ns in ((0.0, 0.4, 32), (0.4, 0.8, 32), (0.8, 1.6, 64)):
        C["ballD_L2_%g_%g" % (lo, hi)] = ("query_ball_point_dilated", (lo, hi, ns), (x2, ctr2))
    x3 = kit[1:2, ::16, :3].copy()
    ctr3 = x3[:, ::2].copy()
    for lo, hi, ns in ((0.0, 1.6, 32), (1.6, 3.2, 32), (3.2, 4.8, 32)):
        C["ballD_L3_%g_%g" % (lo, hi)] = ("query_ball_point_dilated", (lo, hi, ns), (x3, ctr3))
    bx, bc = boundary_cloud((0.2, 0.4, 0.8), seed=3)
    C["ballD_boundary_0.2_0.4"] = ("query_ball_point_dilated", (0.2, 0.4, 48), (bx, bc))
    C["ballD_boundary_0.4_0.8"] = ("query_ball_point_dilated", (0.4, 0.8, 48), (bx, bc))
    C["ballD_boundary_0_0.2"] = ("query_ball_point_dilated", (0.0, 0.2, 48), (bx, bc))
    lat = lattice(6000, step=0.1, seed=2, side=20)
    C["ballD_lattice_0.2_0.4"] = ("query_ball_point_dilated", (0.2, 0.4, 64), (lat, lat[:, ::7].copy()))
    C["ballD_dup_points"] = ("query_ball_point_dilated", (0.0, 0.8, 16),
                             (_dups(x2, 0.2, 9), _dups(x2, 0.2, 9)[:, -600:].copy()))
    # ---- a6 query_ball_point (tf_grouping_g.cu:215-255): layer 4 (3dssd.yaml:64-66), empty balls possible
    x4 = kit[1:2, ::32, :3].copy()
    ctr4 = (x4[:, ::2] + rng.normal(0, 2.5, (1, 256, 3))).astype(f32)
    ctr4[0, :8] += f32(500.0)                                                # certainly empty
    C["ball_L4_4.8"] = ("query_ball_point", (4.8, 16), (x4, ctr4))
    C["ball_L4_6.4"] = ("query_ball_point", (6.4, 32), (x4, ctr4))
    C["ball_cfg0_0.2"] = ("query_ball_point", (0.2, 32), (C["fps_cfg0_4096_512"][2][0], C["fps_cfg0_4096_512"][2][0][:, :512].copy()))
    bx2, bc2 = boundary_cloud((4.8, 6.4), seed=4)
    C["ball_boundary_4.8"] = ("query_ball_point", (4.8, 64), (bx2, bc2))
    C["ball_boundary_6.4"] = ("query_ball_point", (6.4, 64), (bx2, bc2))
    C["ball_lattice_0.5"] = ("query_ball_point", (0.5, 8), (lattice(4000, seed=5), lattice(4000, seed=5)[:, :300].copy()))
    C["ball_nsample_300"] = ("query_ball_point", (6.0, 300), (x2, ctr2[:, :64].copy()))
    # ---- a7 group_point (tf_grouping_g.cu:362-379), idx == -1 rows give zeros
    gi = rng.integers(0, 1024, (2, 128, 32)).astype(np.int32)
    gi[0, 3, 5:] = -1
    gi[1, 100] = -1
    C["group_c67"] = ("group_point", (), (rng.normal(0, 1, (2, 1024, 67)).astype(f32), gi))
    C["group_c3"] = ("group_point", (), (rng.normal(0, 1, (2, 1024, 3)).astype(f32), gi))
    return C


def full_depth_cases():
    """BASELINE.json configs[2] and configs[4] at full depth, one frame each.  Slow on the
    CPU oracle (seconds to a minute); the reference library does them in well under a second."""
    syn = _syn()
    rng = np.random.default_rng(78)
    C = {}
    xyz = syn.kitti_like_batch(1, n=16384, first_frame=11)
    feat = rng.normal(0, 0.5, (1, 16384, 64)).astype(f32)
    C["fps_cfg2_16384x67_4096"] = ("farthest_point_sample", (4096,), (np.concatenate([xyz[:, :, :3], feat], 2),))
    big = syn.kitti_like_batch(1, n=65536, first_frame=12)[:, :, :3].copy()
    C["fps_cfg4_65536_4096"] = ("farthest_point_sample", (4096,), (big,))
    ctr = big[:, ::16].copy()
    for lo, hi, ns in ((0.0, 0.2, 32), (0.2, 0.4, 32), (0.4, 0.8, 64)):
        C["ballD_cfg4_%g_%g" % (lo, hi)] = ("query_ball_point_dilated", (lo, hi, ns), (big, ctr))
    return C


def f4_cases():
    """SURVEY.md 8f rank 4 operators whose results are deterministic in the reference (no float atomics)."""
    syn = _syn()
    rng = np.random.default_rng(79)
    kit = syn.kitti_like_batch(2, n=4096)[:, :, :3].copy()
    C = {}
    boxes = _boxes_on_points(rng, kit, 48)
    C["boxes_mask"] = ("query_boxes_3d_mask", (), (kit, boxes))
    C["boxes_points"] = ("query_boxes_3d_points", (64,), (kit, boxes))
    gt = _boxes_on_points(rng, kit, 6)
    C["points_iou"] = ("query_points_iou", (), (kit, boxes, gt, rng.uniform(0, 0.01, (2, 48, 6)).astype(f32)))
    latb = lattice(3000, seed=6)
    lb = np.array([[[4.0, 6.0, 5.0, 4.0, 3.0, 2.0, 0.0], [6.0, 8.0, 6.0, 2.0, 2.0, 2.0, np.pi / 2],
                    [3.0, 4.0, 3.0, 0.0, 1.0, 0.0, 0.3]]], f32)                # faces on lattice planes, zero-size box
    C["boxes_mask_lattice"] = ("query_boxes_3d_mask", (), (latb, lb))
    mask = (rng.uniform(0, 1, (3, 500)) < 0.1).astype(f32)
    mask[2] = 0
    mask[1, :4] = (0.9, -1.0, 2.0, 0.5)
    C["gather_by_mask"] = ("gather_by_mask", (64,), (rng.normal(0, 1, (3, 500, 9)).astype(f32), mask))
    xs = kit[:, :600].copy()
    qs = xs[:, ::9].copy()
    d = ((qs[:, :, None, :].astype(np.float64) - xs[:, None, :, :]) ** 2).sum(-1)
    C["ball_withidx"] = ("query_ball_point_withidx", (1.5, 16), (xs, qs, np.argsort(d, 2, kind="stable").astype(np.int32)))
    C["select_top_k"] = ("select_top_k", (16,), (rng.normal(0, 1, (2, 50, 1000)).astype(f32),))
    C["select_top_k_ties"] = ("select_top_k", (30,), (np.round(rng.uniform(0, 5, (1, 7, 65))).astype(f32),))
    C["fps_preidx"] = ("farthest_point_sample_with_preidx", (256,),
                       (_dups(kit, 0.1, 3), rng.integers(0, 4096, (2, 64)).astype(np.int32)))
    C["fps_preidx_c67"] = ("farthest_point_sample_with_preidx", (40,),
                           (rng.normal(0, 1, (1, 3000, 67)).astype(f32), rng.integers(0, 3000, (1, 10)).astype(np.int32)))
    C["three_nn"] = ("three_nn", (), (kit[:, :2048].copy(), kit[:, 2048:2560].copy()))
    C["three_nn_lattice"] = ("three_nn", (), (lattice(1500, seed=7), lattice(400, seed=8)))
    ti = rng.integers(0, 512, (2, 2048, 3)).astype(np.int32)
    tw = rng.uniform(0, 1, (2, 2048, 3)).astype(f32)
    C["three_interpolate"] = ("three_interpolate", (), (rng.normal(0, 1, (2, 512, 64)).astype(f32), ti, tw))
    ki = rng.integers(0, 512, (2, 700, 5)).astype(np.int32)
    kw = rng.uniform(0, 1, (2, 700, 5)).astype(f32)
    C["k_interpolate"] = ("k_interpolate", (), (rng.normal(0, 1, (2, 512, 33)).astype(f32), ki, kw))
    return C


# ------------------------------------------------------------------------------------------------------ runners
def as_tuple(out):
    return tuple(out) if isinstance(out, (tuple, list)) else (out,)


def run_numpy(api, case):
    """api: a module/object with the reference's operator names taking numpy arrays (the oracle)."""
    op, scalars, arrays = case
    return tuple(np.asarray(o) for o in as_tuple(getattr(api, op)(*scalars, *arrays)))


def run_torch(api, case, device):
    """api: RefOps or the HIP wrapper namespace, taking CUDA tensors."""
    import torch
    op, scalars, arrays = case
    ts = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in arrays]
    out = as_tuple(getattr(api, op)(*scalars, *ts))
    torch.cuda.synchronize()
    return tuple(o.cpu().numpy() for o in out)
