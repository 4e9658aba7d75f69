"""world_size-2 gloo test of the multi-GPU path (frame sharding, timing reduction, output gather)."""
import os
import sys

import torch
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    sh = importlib.import_module("3dssd_amd.sharding")
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = sh.init(backend="gloo")
    mine = sh.frames_of_rank(10, 7, r, w)
    sh.barrier()
    t, total = sh.reduce_timing(1.0 + r, len(mine))
    xyz = torch.full((len(mine), 4, 3), float(r))
    feat = torch.full((len(mine), 4, 8), float(r))
    # all_gather needs equal shapes: pad to the common local batch like the bench does (fixed per-rank batch)
    pad = 4 - len(mine)
    xyz = torch.cat([xyz, torch.zeros(pad, 4, 3)])
    feat = torch.cat([feat, torch.zeros(pad, 4, 8)])
    xs, fs = sh.gather_outputs(xyz, feat)
    chk = sh.gather_check(xyz, feat)
    q.put((r, mine, t, total, [float(x[0, 0, 0]) for x in xs], chk))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, t0, n0, g0, c0), (r1, f1, t1, n1, g1, c1) = res
    for c in (c0, c1):                                         # the "result gather" of configs[3], checked by digest
        assert c["ranks_seen"] == 2 and c["world"] == 2 and c["rank_order_ok"] and c["distinct_rank_digests"] == 2
        assert c["bytes_gathered"] == 2 * (4 * 4 * 3 + 4 * 4 * 8) * 4 and c["backend"] == "gloo"
    assert f0 == [10, 12, 14, 16] and f1 == [11, 13, 15]       # f mod 2 == rank, disjoint, complete
    assert t0 == t1 == 2.0                                     # max over ranks
    assert n0 == n1 == 7                                       # frames add up
    assert g0 == g1 == [0.0, 1.0]                              # gathered in rank order


def test_single_process_passthrough():
    sh = pkg("sharding")
    assert sh.frames_of_rank(0, 5, 0, 1) == [0, 1, 2, 3, 4]
    assert sh.reduce_timing(0.5, 8) == (0.5, 8)
    assert sh.gather_check(torch.zeros(1, 2, 3), torch.zeros(1, 2, 4))["ranks_seen"] == 1
