"""Size-independent properties of the HIP path at BASELINE.json's FULL sizes (configs[1]: 8 frames x 16384 points,
the whole backbone), where the CPU oracle would take minutes: things that must hold for any correct
implementation of the reference operators and that a wrong kernel would almost surely break.

  * D-FPS: first index 0, no repeated index, and the distance of every new pick to the already picked set is
    non-increasing (the defining property of farthest-point sampling), checked with torch on the GPU;
  * ball query: the first cnt entries of a row are strictly increasing point indices (the reference's scan
    order), every one of them lies in the band, the padding repeats the first hit, cnt <= nsample, and a row with
    cnt < nsample holds ALL points of the band (torch brute force on a sample of queries);
  * cooperative FPS (configs[2] c = 67 and configs[4] n = 65536, full size): the same FPS properties, and its
    first picks equal those of the independent single-workgroup kernel;
  * distance matrix: bitwise symmetric;
  * fused grouped MLP: invariant (bit for bit) under any permutation of the samples inside a ball;
  * backbone: batched == frame by frame, bit for bit (no cross-frame coupling);
  * determinism: two runs of the backbone give identical bits;
  * execution mode of bench.py: hipGraph replays on four concurrent streams reproduce the eager result bit for bit.
"""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames(gpu):
    syn = pkg("synthetic")
    return torch.from_numpy(syn.kitti_like_batch(8)).to(gpu)


def test_dfps_full_size_properties(gpu, frames):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    xyz = frames[:, :, :3].contiguous()
    m = 4096
    idx = S.farthest_point_sample(m, xyz)
    torch.cuda.synchronize()
    assert idx.shape == (8, m) and idx.dtype == torch.int32
    assert (idx[:, 0] == 0).all()
    for b in range(8):
        assert torch.unique(idx[b]).numel() == m          # distinct points -> distinct picks
    # distance of pick i to picks 0..i-1, in chunks (fp64 on the GPU): non-increasing
    for b in (0, 5):
        p = xyz[b, idx[b].long()].double()                # [m,3]
        prev = None
        mind = torch.full((m,), float("inf"), dtype=torch.float64, device=gpu)
        for s in range(0, m, 512):
            d = torch.cdist(p[s:s + 512], p)              # [512, m]
            j = torch.arange(m, device=gpu)[None, :]
            i = torch.arange(s, min(s + 512, m), device=gpu)[:, None]
            d = torch.where(j < i, d, torch.full_like(d, float("inf")))
            mind[s:s + 512] = d.min(1).values
        seq = mind[1:]
        assert (seq[1:] <= seq[:-1] * (1 + 1e-6)).all(), "FPS pick distances must be non-increasing"
        assert seq[-1] > 0


def _fps_sequence_is_farthest_first(xyz_b, idx_b, gpu):
    """distance of pick i to picks 0..i-1 is non-increasing in i (fp64 on the GPU); returns the last distance"""
    m = idx_b.numel()
    p = xyz_b[idx_b.long()].double()
    mind = torch.full((m,), float("inf"), dtype=torch.float64, device=gpu)
    for s in range(0, m, 512):
        d = torch.cdist(p[s:s + 512], p)
        j = torch.arange(m, device=gpu)[None, :]
        i = torch.arange(s, min(s + 512, m), device=gpu)[:, None]
        mind[s:s + 512] = torch.where(j < i, d, torch.full_like(d, float("inf"))).min(1).values
    seq = mind[1:]
    assert (seq[1:] <= seq[:-1] * (1 + 1e-6)).all(), "FPS pick distances must be non-increasing"
    return float(seq[-1])


@pytest.mark.parametrize("b,n,c,m", [(32, 16384, 67, 4096),     # BASELINE.json configs[2]: F-FPS isolated, 3 + 64 channels
                                     (16, 65536, 3, 4096)])     # configs[4]: 65536-point frames
def test_cooperative_fps_full_size_properties(gpu, b, n, c, m):
    """fps_coop.hip at the full sizes of configs[2] / configs[4]: FPS properties, and the first picks equal the
    single-workgroup global-scratch kernel's (an independent implementation; all m picks would take it seconds)."""
    import ctypes
    S, N = pkg("utils.tf_ops.sampling.tf_sampling"), pkg("utils._native")
    if c == 3:
        x = torch.from_numpy(pkg("synthetic").kitti_like_batch(b, n=n)).to(gpu)[:, :, :3].contiguous()
    else:
        x = torch.randn(b, n, c, device=gpu, generator=torch.Generator(device=gpu).manual_seed(5))
    idx = S.farthest_point_sample(m, x)
    torch.cuda.synchronize()
    assert idx.shape == (b, m) and (idx[:, 0] == 0).all()
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    for f in range(b):
        assert torch.unique(idx[f]).numel() == m
    for f in (0, b - 1):
        assert _fps_sequence_is_farthest_first(x[f], idx[f], gpu) > 0
    k = 48
    sub = x[b - 2:].contiguous()
    temp = torch.empty((2, n), dtype=torch.float32, device=gpu)
    out = torch.empty((2, k), dtype=torch.int32, device=gpu)
    assert N.lib().sa_fps_generic(2, n, c, k, sub.data_ptr(), temp.data_ptr(), out.data_ptr(), 0, N.current_stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, idx[b - 2:, :k])


def test_ball_query_full_size_properties(gpu, frames):
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    xyz = frames[:, :, :3].contiguous()
    ctr = S.gather_point(xyz, S.farthest_point_sample(4096, xyz))
    for rmin, rmax, ns in ((0.0, 0.2, 32), (0.4, 0.8, 64)):
        idx, cnt = G.query_ball_point_dilated(rmin, rmax, ns, xyz, ctr)
        torch.cuda.synchronize()
        assert idx.shape == (8, 4096, ns) and cnt.shape == (8, 4096)
        assert (cnt >= 1).all() and (cnt <= ns).all()       # a centre is a member of xyz: d == 0 always hits
        ar = torch.arange(ns, device=gpu)[None, None, :]
        valid = ar < cnt[:, :, None]
        # strictly increasing inside the valid prefix
        inc = (idx[:, :, 1:] > idx[:, :, :-1]) | ~valid[:, :, 1:]
        assert inc.all()
        # padding repeats the first hit
        assert ((idx == idx[:, :, :1]) | valid).all()
        # every listed point is in the band (sqrt form of the reference, tf_grouping_g.cu:336-346)
        g = torch.gather(xyz, 1, idx.reshape(8, -1, 1).expand(-1, -1, 3).long()).reshape(8, 4096, ns, 3)
        d = torch.sqrt(((g - ctr[:, :, None, :]) ** 2).sum(-1))
        ok = (d == 0) | ((d >= rmin) & (d < rmax))
        assert (ok | ~valid).float().mean() > 0.99999       # fp32 re-association at the band edge only
        # completeness on a sample of queries: cnt < ns  =>  cnt == number of band members
        for b in (0, 7):
            q = torch.arange(0, 4096, 37, device=gpu)
            dd = torch.cdist(ctr[b, q].double(), xyz[b].double())
            members = ((dd == 0) | ((dd >= rmin) & (dd < rmax))).sum(1)
            c = cnt[b, q].long()
            short = c < ns
            assert ((members[short] - c[short]).abs() <= 1).all()   # +-1: a point exactly on the band edge
            assert (members[~short] >= ns - 1).all()


def test_distance_matrix_bitwise_symmetric_full_size(gpu):
    M = pkg("utils.model_util")
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.randn(2, 4096, 67, generator=g).to(gpu)
    d = M.calc_square_dist(a, a, norm=False)
    torch.cuda.synchronize()
    assert torch.equal(d, d.transpose(1, 2))
    assert (torch.diagonal(d, dim1=1, dim2=2).abs() < 1e-3).all()


@pytest.mark.parametrize("c,ns,dims", [(1, 32, [16, 16, 32]), (64, 64, [64, 96, 128]), (128, 32, [128, 192, 256]),
                                       (256, 32, [256, 512, 1024])])
def test_grouped_mlp_is_permutation_invariant_inside_a_ball(gpu, c, ns, dims):
    import ctypes
    N, Wt = pkg("utils._native"), pkg("utils.weights")
    rng = np.random.default_rng(c + ns)
    b, n, m = 2, 2048, 300
    xyz = torch.from_numpy(rng.uniform(-5, 5, (b, n, 3)).astype(np.float32)).to(gpu)
    feat = torch.from_numpy(rng.normal(0, 1, (b, n, c)).astype(np.float32)).to(gpu)
    new_xyz = xyz[:, :m].contiguous()
    idx = torch.from_numpy(rng.integers(0, n, (b, m, ns)).astype(np.int32)).to(gpu)
    cnt = torch.full((b, m), ns, dtype=torch.int32, device=gpu)
    ws = [rng.normal(0, 1.0 / np.sqrt(k), (k, o)).astype(np.float32) for k, o in zip([c + 3] + dims[:-1], dims)]
    bs = [rng.normal(0, 0.1, o).astype(np.float32) for o in dims]
    layers = Wt.pack_scale(ws, bs, gpu)
    nl = len(layers)

    def run(ix):
        out = torch.empty((b, m, dims[-1]), dtype=torch.float32, device=gpu)
        dm = (ctypes.c_int * (nl + 1))(*([c + 3] + dims))
        plan, plan_bytes = N.mlp_plan_ws(b, m, ns, gpu)
        st = N.lib().sa_group_mlp_max(b, n, m, ns, c, xyz.data_ptr(), feat.data_ptr(), new_xyz.data_ptr(), ix.data_ptr(),
                                      cnt.data_ptr(), nl, dm, (ctypes.c_void_p * nl)(*[l.w.data_ptr() for l in layers]),
                                      (ctypes.c_void_p * nl)(*[l.bias.data_ptr() for l in layers]), out.data_ptr(),
                                      dims[-1], 0, plan.data_ptr(), plan_bytes, Wt.scale_flags(layers), None, N.current_stream())
        assert st == 0
        torch.cuda.synchronize()
        return out

    ref = run(idx)
    perm = torch.from_numpy(np.stack([rng.permutation(ns) for _ in range(b * m)]).reshape(b, m, ns)).to(gpu)
    shuffled = torch.gather(idx, 2, perm.long()).contiguous()
    assert torch.equal(run(shuffled), ref)


def test_backbone_batched_equals_per_frame_and_is_deterministic(gpu, frames):
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    params = syn.random_backbone_params(arch)
    net = pkg("backbone").SABackbone(arch, params, gpu, cfgs.KITTI_MAX_TRANSLATE_RANGE)
    xl, fl, il = net(frames)
    torch.cuda.synchronize()
    xl2, fl2, il2 = net(frames)
    torch.cuda.synchronize()
    for a, b_ in zip(xl + fl, xl2 + fl2):
        if a is not None:
            assert torch.equal(a, b_)
    assert xl[-1].shape == (8, 256, 3) and fl[-1].shape == (8, 256, 512)
    assert torch.isfinite(fl[-1]).all()
    for f in (0, 3, 7):
        x1, f1, i1 = net(frames[f:f + 1].contiguous())
        torch.cuda.synchronize()
        for li in range(1, len(xl)):
            assert torch.equal(x1[li][0], xl[li][f]), "centres of list index %d differ for frame %d" % (li, f)
            assert torch.equal(f1[li][0], fl[li][f]), "features of list index %d differ for frame %d" % (li, f)
            if il[li] is not None:
                assert torch.equal(i1[li][0], il[li][f])


def test_graph_replay_on_concurrent_streams_equals_eager(gpu, frames):
    """The execution mode of the headline number: one captured hipGraph per stream, replays of several streams in flight
    at once, EVERY STREAM ON A DIFFERENT BATCH (identical inputs would hide shared scratch: both streams would write the
    same bytes).  Every replay must reproduce the eager single-stream result of ITS batch bit for bit, three rounds with
    the batches rotated over the streams (tests/test_pipeline_gpu.py drives the packaged executor the same way)."""
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), gpu, cfgs.KITTI_MAX_TRANSLATE_RANGE)
    nstream = 4
    batches = [torch.from_numpy(syn.kitti_like_batch(4, first_frame=200 + 4 * i)).to(gpu) for i in range(nstream)]
    refs = []
    for t in batches:
        xl, fl, _ = net(t)
        refs.append((xl[-1].clone(), fl[-1].clone()))
    torch.cuda.synchronize()
    assert not torch.equal(refs[0][1], refs[1][1])
    streams = [torch.cuda.Stream(device=gpu) for _ in range(nstream)]
    graphs = []
    for st in streams:
        inp = torch.zeros_like(batches[0])
        inp.copy_(batches[0])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            gx, gf, _ = net(inp)
        graphs.append((g, inp, gx[-1], gf[-1]))
    torch.cuda.synchronize()
    for rnd in range(3):
        for (g, inp, ox, of) in graphs:
            of.fill_(-1.0)                                   # on the default stream, before the replays of this round
        torch.cuda.synchronize()
        for j, ((g, inp, _, _), st) in enumerate(zip(graphs, streams)):
            with torch.cuda.stream(st):
                inp.copy_(batches[(j + rnd) % nstream], non_blocking=True)
                g.replay()
        torch.cuda.synchronize()
        for j, (g, inp, ox, of) in enumerate(graphs):
            rx, rf = refs[(j + rnd) % nstream]
            assert torch.equal(ox, rx) and torch.equal(of, rf), "round %d stream %d" % (rnd, j)


def test_non_finite_inputs_do_not_hang_or_leave_the_index_range(gpu):
    # The library is built -fno-honor-nans and documents non-finite inputs as unsupported: picks involving
    # a NaN / Inf point are unspecified (the reference's comparisons simply never select them, tf_sampling_g.cu:151-157).
    # What IS guaranteed and checked here: the kernels terminate, every index stays inside the frame, finite frames of the
    # same batch are unaffected.
    S, G = pkg("utils.tf_ops.sampling.tf_sampling"), pkg("utils.tf_ops.grouping.tf_grouping")
    rng = np.random.default_rng(11)
    for n, m in ((3000, 200), (16384, 512)):
        p = rng.uniform(-10, 10, (2, n, 3)).astype(np.float32)
        clean = p.copy()
        p[0, 17] = np.nan
        p[0, 99, 1] = np.inf
        t = torch.from_numpy(p).to(gpu)
        idx = S.farthest_point_sample(m, t).cpu().numpy()
        assert idx.min() >= 0 and idx.max() < n
        ref = S.farthest_point_sample(m, torch.from_numpy(clean).to(gpu)).cpu().numpy()
        assert np.array_equal(idx[1], ref[1])                    # the finite frame of the batch is untouched
        ctr = S.gather_point(t, torch.from_numpy(ref).to(gpu))
        gi, gc = G.query_ball_point(1.0, 16, t, ctr)
        gi, gc = gi.cpu().numpy(), gc.cpu().numpy()
        assert gi.min() >= 0 and gi.max() < n and gc.min() >= 0 and gc.max() <= 16
