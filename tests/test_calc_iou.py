"""calc_iou / calc_iou_match (lib/utils/tf_ops/evaluation): analytic known answers and a Monte-Carlo cross-check of the
oracle on CPU; the HIP kernel (a different clipping algorithm) against the oracle on GPU, to rounding."""
import numpy as np
import pytest
import torch

from conftest import pkg


def _t(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def _box(x, y, z, l, h, w, ry):
    return [x, y, z, l, h, w, ry]


KATS = [
    # (det, gt, iou_bev, iou_3d)
    (_box(0, 0, 0, 2, 1, 2, 0), _box(0, 0, 0, 2, 1, 2, 0), 1.0, 1.0),                       # identical
    (_box(0, 0, 0, 2, 1, 2, 0), _box(10, 0, 0, 2, 1, 2, 0), 0.0, 0.0),                      # disjoint
    (_box(0, 0, 0, 2, 1, 2, 0), _box(1, 0, 0, 2, 1, 2, 0), 2.0 / 6.0, 2.0 / 6.0),           # half overlap along x
    (_box(0, 0, 0, 2, 1, 2, 0), _box(1, 0, 1, 2, 1, 2, 0), 1.0 / 7.0, 1.0 / 7.0),           # quarter overlap
    (_box(0, 0, 0, 4, 1, 2, 0), _box(0, 0, 0, 4, 1, 2, np.pi / 2), 4.0 / 12.0, 4.0 / 12.0),  # a cross: 2 x 2 core
    (_box(0, 0, 0, 2, 1, 2, 0), _box(0, 0, 0, 2, 1, 2, np.pi / 4), (8 * np.sqrt(2) - 8) / (16 - 8 * np.sqrt(2)),
     (8 * np.sqrt(2) - 8) / (16 - 8 * np.sqrt(2))),                                          # square vs itself turned 45 deg: octagon
    (_box(0, 0, 0, 2, 2, 2, 0), _box(0, -1, 0, 2, 2, 2, 0), 1.0, 4.0 / 12.0),               # same footprint, half the height shared
    (_box(0, 0, 0, 2, 1, 2, 0), _box(0, -1, 0, 2, 1, 2, 0), 1.0, 0.0),                       # stacked: touching in y only
    (_box(0, 0, 0, 2, 1, 2, 0), _box(0, 0, 0, 1, 1, 1, 0.3), 0.25, 0.25),                    # small turned box inside
    (_box(0, 0, 0, 2, 1, 2, 0), _box(2, 0, 0, 2, 1, 2, 0), 0.0, 0.0),                        # sharing an edge only
]


def test_oracle_iou_known_answers(oracle):
    d = np.array([[k[0] for k in KATS]], np.float32)
    g = np.array([[k[1] for k in KATS]], np.float32)
    bev, i3d = oracle.calc_iou_match(d[0], g[0])
    np.testing.assert_allclose(bev, [k[2] for k in KATS], rtol=0, atol=2e-6)
    np.testing.assert_allclose(i3d, [k[3] for k in KATS], rtol=0, atol=2e-6)
    full_bev, full_3d = oracle.calc_iou(d, g)
    assert full_bev.shape == (1, len(KATS), len(KATS))
    np.testing.assert_allclose(np.diagonal(full_bev[0]), bev, rtol=0, atol=0)
    self_bev, _ = oracle.calc_iou(g, g)
    np.testing.assert_allclose(self_bev[0], self_bev[0].T, rtol=0, atol=1e-6)                # IoU is symmetric
    np.testing.assert_allclose(np.diagonal(self_bev[0]), 1.0, rtol=0, atol=1e-6)


def test_oracle_iou_against_monte_carlo(oracle):
    rng = np.random.default_rng(11)
    n = 12
    d = np.concatenate([rng.normal(0, 1, (n, 3)), rng.uniform(1, 4, (n, 3)), rng.uniform(-np.pi, np.pi, (n, 1))], -1).astype(np.float32)
    g = d + np.concatenate([rng.normal(0, 0.7, (n, 3)), rng.normal(0, 0.3, (n, 3)), rng.normal(0, 0.6, (n, 1))], -1).astype(np.float32)
    g[:, 3:6] = np.abs(g[:, 3:6]) + 0.5
    bev, _ = oracle.calc_iou_match(d, g)

    def inside(b, px, pz):
        c, s = np.cos(b[6]), np.sin(b[6])
        dx, dz = px - b[0], pz - b[2]
        u, v = c * dx - s * dz, s * dx + c * dz                       # inverse of [[c, s], [-s, c]]
        return (np.abs(u) <= b[3] / 2) & (np.abs(v) <= b[5] / 2)

    for i in range(n):
        px, pz = rng.uniform(-8, 8, 400000), rng.uniform(-8, 8, 400000)
        a, b_ = inside(d[i].astype(np.float64), px, pz), inside(g[i].astype(np.float64), px, pz)
        mc = (a & b_).sum() / max((a | b_).sum(), 1)
        assert abs(mc - bev[i]) < 0.02, (i, mc, bev[i])


@pytest.mark.gpu
def test_hip_iou_known_answers_and_random_vs_oracle(gpu, oracle):
    E = pkg("utils.tf_ops.evaluation.tf_evaluate")
    d = np.array([k[0] for k in KATS], np.float32)
    g = np.array([k[1] for k in KATS], np.float32)
    bev, i3d = E.calc_iou_match(_t(d, gpu), _t(g, gpu))
    np.testing.assert_allclose(bev.cpu().numpy(), [k[2] for k in KATS], rtol=0, atol=2e-6)
    np.testing.assert_allclose(i3d.cpu().numpy(), [k[3] for k in KATS], rtol=0, atol=2e-6)
    rng = np.random.default_rng(5)
    bs, dn, gn = 3, 200, 17
    gt = np.concatenate([rng.normal(0, 6, (bs, gn, 3)), rng.uniform(0.5, 5, (bs, gn, 3)), rng.uniform(-np.pi, np.pi, (bs, gn, 1))], -1).astype(np.float32)
    det = gt[:, rng.integers(0, gn, dn)] + rng.normal(0, 0.5, (bs, dn, 7)).astype(np.float32)
    det[..., 3:6] = np.abs(det[..., 3:6]) + 0.1
    det[:, :5] = gt[:, :5]                                             # exact copies: IoU 1
    det[:, 5, 3:6] = 0.0                                               # a degenerate detection
    b1, t1 = E.calc_iou(_t(det, gpu), _t(gt, gpu))
    rb, rt = oracle.calc_iou(det, gt)
    assert tuple(b1.shape) == (bs, dn, gn)
    np.testing.assert_allclose(b1.cpu().numpy(), rb, rtol=0, atol=3e-6)
    np.testing.assert_allclose(t1.cpu().numpy(), rt, rtol=0, atol=3e-6)
    assert np.allclose(rb[:, np.arange(5), np.arange(5)], 1.0, atol=1e-6) and (rb[:, 5] == 0).all()
    assert (rb >= 0).all() and (rb <= 1 + 1e-6).all() and (rt <= rb + 1e-6).all() and (rb > 0.05).mean() > 0.02
    wb, wt = E.calc_iou_match_warper(_t(det[:, :gn], gpu), _t(gt, gpu))
    assert tuple(wb.shape) == (bs, gn)
    np.testing.assert_allclose(wb.cpu().numpy(), rb[:, np.arange(gn), np.arange(gn)], rtol=0, atol=3e-6)
    with pytest.raises(ValueError, match="detections shape"):
        E.calc_iou(torch.zeros(1, 3, 6, device=gpu), torch.zeros(1, 2, 7, device=gpu))
    with pytest.raises(ValueError, match="gt shape"):
        E.calc_iou_match(torch.zeros(3, 7, device=gpu), torch.zeros(2, 7, device=gpu))
