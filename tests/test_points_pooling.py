"""points_pooling (lib/utils/tf_ops/points_pooling): oracle known answers on CPU, HIP vs oracle bit for bit on GPU."""
import numpy as np
import pytest
import torch

from conftest import pkg


def _t(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def test_oracle_points_pooling_known_answers(oracle):
    # box: centre x 1, bottom y 2, centre z 3; l = 4 (x from -1 to 3), h = 2 (y from 0 to 2), w = 2 (z from 2 to 4); grid 2 x 1 x 2
    box = np.array([[[1, 2, 3, 4, 2, 2]]], np.float32)
    loc = np.array([[[[-0.5, 1, 2.5],      # voxel (0,0,0)
                      [2.9, 0.1, 3.9],     # voxel (1,0,1)
                      [0.9, 1.9, 2.1],     # voxel (0,0,0) second
                      [-7, 1, 9],          # outside: clamped to (0,0,1)
                      [0.0, 1, 2.0],       # voxel (0,0,0) third: dropped (sample_num 2)
                      [1.0, 1, 3.0]]]], np.float32)   # exactly on the faces x = 1, z = 3: voxel (1,0,1)
    pc = np.arange(6 * 2, dtype=np.float32).reshape(1, 1, 6, 2) + 1
    feats, idx, num, pillars = oracle.points_pooling(pc, box, loc, l=2, h=1, w=2, sample_num=2)
    assert num.reshape(-1).tolist() == [2, 1, 0, 2]                      # voxels (0,0,0) (0,0,1) (1,0,0) (1,0,1)
    assert idx.reshape(4, 2).tolist() == [[0, 2], [3, 0], [0, 0], [1, 5]]
    assert feats.reshape(4, 2, 2)[0].tolist() == [[1, 2], [5, 6]] and feats.reshape(4, 2, 2)[3].tolist() == [[3, 4], [11, 12]]
    assert (feats.reshape(4, 2, 2)[2] == 0).all()
    assert pillars.reshape(4, 3).tolist() == [[0, 1, 2.5], [0, 1, 3.5], [2, 1, 2.5], [2, 1, 3.5]]
    g = np.ones_like(feats)
    pg = oracle.points_pooling_grad(pc, idx, num, g)
    assert pg.reshape(6, 2)[:, 0].tolist() == [1, 1, 1, 1, 0, 1]        # the dropped point gets no gradient


@pytest.mark.gpu
@pytest.mark.parametrize("bs,pn,pts,c,l,h,w,sn", [(2, 64, 512, 16, 7, 7, 7, 35), (1, 5, 100, 3, 2, 3, 4, 2),
                                                  (2, 9, 65, 1, 1, 1, 1, 100), (1, 3, 700, 131, 12, 12, 12, 4)])
def test_points_pooling_matches_oracle(gpu, oracle, bs, pn, pts, c, l, h, w, sn):
    P = pkg("utils.tf_ops.points_pooling.points_pooling")
    rng = np.random.default_rng(pts + c)
    box = np.concatenate([rng.normal(0, 5, (bs, pn, 3)), rng.uniform(1, 5, (bs, pn, 3))], -1).astype(np.float32)
    ctr = box[..., :3].copy()
    ctr[..., 1] -= box[..., 4] / 2
    loc = (ctr[:, :, None, :] + rng.uniform(-0.6, 0.6, (bs, pn, pts, 3)) * box[:, :, None, 3:6]).astype(np.float32)
    if pts > 64:
        loc[:, :, 10:20] = np.round(loc[:, :, 10:20] * 2) / 2            # points on round coordinates
        loc[:, 0, 30:60] = loc[:, 0, 30:31]                               # 30 identical points: one crowded voxel
    pc = rng.normal(0, 1, (bs, pn, pts, c)).astype(np.float32)
    got = P.points_pooling(_t(pc, gpu), _t(box, gpu), _t(loc, gpu), l=l, h=h, w=w, sample_num=sn)
    ref = oracle.points_pooling(pc, box, loc, l=l, h=h, w=w, sample_num=sn)
    for name, a, b in zip(("features", "idx", "num", "pillars"), got, ref):
        assert tuple(a.shape) == b.shape and np.array_equal(a.cpu().numpy(), b), name
    assert ref[2].max() <= sn and ref[2].sum() > 0
    g = rng.integers(-4, 5, ref[0].shape).astype(np.float32)            # exact partial sums -> order-free
    pg = P.points_pooling_grad(_t(pc, gpu), got[1], got[2], _t(g, gpu)).cpu().numpy()
    assert np.array_equal(pg, oracle.points_pooling_grad(pc, ref[1], ref[2], g))


@pytest.mark.gpu
def test_points_pooling_argument_errors(gpu):
    P = pkg("utils.tf_ops.points_pooling.points_pooling")
    pc, box, loc = torch.zeros(1, 2, 8, 3, device=gpu), torch.ones(1, 2, 6, device=gpu), torch.zeros(1, 2, 8, 3, device=gpu)
    with pytest.raises(ValueError, match="positive length"):
        P.points_pooling(pc, box, loc, l=0)
    with pytest.raises(ValueError, match="proposal shape"):
        P.points_pooling(pc, torch.ones(1, 2, 7, device=gpu), loc)
    with pytest.raises(ValueError, match="pc_loc shape"):
        P.points_pooling(pc, box, torch.zeros(1, 2, 7, 3, device=gpu))
    with pytest.raises(RuntimeError, match="unsupported"):
        P.points_pooling(pc, box, loc, l=16, h=16, w=16)                 # 4096 voxels > the LDS table
