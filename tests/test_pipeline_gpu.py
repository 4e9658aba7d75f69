"""The executor of the headline number (3dssd_amd/pipeline.py, SAPipeline): N slots = N HIP streams x captured hipGraphs
with per-slot static input / output buffers.  VERDICT r2: concurrency was only ever proven on IDENTICAL inputs (two
streams that wrongly shared scratch would have written identical bytes).  Here every batch in flight is different and
every output is compared bit for bit with the eager single-stream result of the same batch."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _batches(variant, nbatch, batch, first=300, n=16384):
    syn = pkg("synthetic")
    return [np.stack([syn.frame_of(variant, first + i * batch + j, n) for j in range(batch)]) for i in range(nbatch)]


@pytest.fixture(scope="module")
def pipe6(gpu):
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    return pkg("pipeline").SAPipeline(arch, syn.random_backbone_params(arch), gpu, batch=2, points=16384, streams=6,
                                      max_translate_range=cfgs.KITTI_MAX_TRANSLATE_RANGE)


def test_forty_distinct_batches_through_the_pipeline_equal_eager(gpu, pipe6):
    pipe = pipe6
    host = _batches("default", 40, 2)
    dev = [torch.from_numpy(h).to(gpu) for h in host]
    eager = []
    for t in dev:
        xl, fl, il = pipe.forward_eager(t)
        eager.append((xl[-1].clone(), fl[-1].clone()))
    torch.cuda.synchronize()
    assert not torch.equal(eager[0][1], eager[1][1])            # the batches really differ
    # (a) results copied out by the pipeline itself (out=): all 40 submitted back to back, 6 in flight at any time
    outs = [(torch.empty_like(e[0]), torch.empty_like(e[1])) for e in eager]
    tickets = [pipe.submit(t, out=o) for t, o in zip(dev, outs)]
    for i, tk in enumerate(tickets):
        x, f = tk.result()
        assert torch.equal(x, eager[i][0]) and torch.equal(f, eager[i][1]), "batch %d differs from its eager result" % i
    # (b) the slots' static buffers, read before the slot is reused; a stale ticket must refuse
    first = [pipe.submit(t) for t in dev[:6]]
    for i, tk in enumerate(first):
        x, f = tk.result(copy=True)
        assert torch.equal(f, eager[i][1])
    later = [pipe.submit(t) for t in dev[6:12]]
    with pytest.raises(RuntimeError, match="reused"):
        first[0].result()
    for i, tk in enumerate(later):
        assert torch.equal(tk.result()[1], eager[6 + i][1])
    pipe.drain()


def test_pipeline_takes_pinned_host_batches_and_checks_shapes(gpu, pipe6):
    pipe = pipe6
    h = torch.from_numpy(_batches("dup10", 1, 2, first=900)[0]).pin_memory()
    x, f = pipe.submit(h).result(copy=True)
    xl, fl, _ = pipe.forward_eager(h.to(gpu))
    torch.cuda.synchronize()
    assert torch.equal(f, fl[-1]) and torch.equal(x, xl[-1])
    with pytest.raises(ValueError):
        pipe.submit(torch.zeros((3, 16384, 4), device=gpu))
    with pytest.raises(ValueError):
        pipe.submit(torch.zeros((2, 16384, 4), device=gpu, dtype=torch.float64))


@pytest.mark.parametrize("variant", ["dup10", "dense"])
def test_pipeline_on_the_other_data_variants(gpu, pipe6, variant):
    # the sensitivity variants of bench.py --data: duplicated rows (tie-breaks) and the uniform box where every ball is
    # full (candidate lists of the grid ball query overflow, row plans are dense)
    pipe = pipe6
    dev = [torch.from_numpy(h).to(gpu) for h in _batches(variant, 6, 2, first=50)]
    eager = [pipe.forward_eager(t)[1][-1].clone() for t in dev]
    torch.cuda.synchronize()
    tickets = [pipe.submit(t) for t in dev]
    for i, tk in enumerate(tickets):
        assert torch.equal(tk.result()[1], eager[i]), "%s batch %d" % (variant, i)
    assert torch.isfinite(eager[0]).all()


def test_eager_pipeline_mode_for_uncapturable_frames(gpu):
    # n > 16384: the layer-1 sampler is the cooperative multi-workgroup kernel, which cannot be captured -> graphs=False
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    pipe = pkg("pipeline").SAPipeline(arch, syn.random_backbone_params(arch), gpu, batch=1, points=20000, streams=3,
                                      graphs=False, max_translate_range=cfgs.KITTI_MAX_TRANSLATE_RANGE)
    dev = [torch.from_numpy(h).to(gpu) for h in _batches("default", 5, 1, first=10, n=20000)]
    eager = [pipe.forward_eager(t)[1][-1].clone() for t in dev]
    torch.cuda.synchronize()
    outs = [(torch.empty((1, 256, 3), device=gpu), torch.empty((1, 256, 512), device=gpu)) for _ in dev]
    tickets = [pipe.submit(t, out=o) for t, o in zip(dev, outs)]
    for i, tk in enumerate(tickets):
        assert torch.equal(tk.result()[1], eager[i])


def test_coalescing_slots_give_every_batch_its_own_eager_result(gpu):
    # coalesce=3: a slot takes three consecutive batches and runs the backbone over all six frames in one pass.  Frames
    # never interact, so every batch must come out exactly as it does alone (eager, its own two frames) -- whichever
    # batches it shared a replay with, and also from a slot that was launched only partly filled.
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    pipe = pkg("pipeline").SAPipeline(arch, syn.random_backbone_params(arch), gpu, batch=2, points=16384, streams=3,
                                      coalesce=3, max_translate_range=cfgs.KITTI_MAX_TRANSLATE_RANGE)
    dev = [torch.from_numpy(h).to(gpu) for h in _batches("default", 20, 2, first=700)]
    eager = []
    for t in dev:
        xl, fl, il = pipe.forward_eager(t)
        eager.append((xl[-1].clone(), fl[-1].clone(), [None if v is None else v.clone() for v in il]))
    torch.cuda.synchronize()
    # 20 batches = 6 full replays + one slot holding 2 of 3: result() launches it
    tickets = [pipe.submit(t) for t in dev[:9]]
    assert tickets[0].wait().done() and tickets[8].wait().done()
    for i, tk in enumerate(tickets):
        x, f = tk.result()
        assert x.shape == (2, 256, 3) and f.shape == (2, 256, 512)
        assert torch.equal(x, eager[i][0]) and torch.equal(f, eager[i][1]), "batch %d" % i
    outs = [(torch.empty_like(e[0]), torch.empty_like(e[1])) for e in eager]
    tickets = [pipe.submit(t, out=o) for t, o in zip(dev, outs)]
    assert not tickets[-1].done()                    # batches 18, 19 sit in a slot that still waits for a third
    x, f = tickets[-1].result()                      # ... and is launched by the first result() on it
    assert torch.equal(f, eager[19][1])
    for i, tk in enumerate(tickets):
        x, f = tk.result()
        assert torch.equal(x, eager[i][0]) and torch.equal(f, eager[i][1]), "batch %d (out=)" % i
    # per-batch views of every list the backbone returns
    tk = pipe.submit(dev[5])
    xl, fl, il = tk.all_outputs()
    assert torch.equal(fl[-1], eager[5][1]) and xl[1].shape[0] == 2
    for a, b in zip(il, eager[5][2]):
        assert (a is None and b is None) or torch.equal(a, b)
    # a stale ticket refuses: its slot has started a later round
    later = [pipe.submit(t) for t in dev[:9]]
    with pytest.raises(RuntimeError, match="reused"):
        tk.result()
    pipe.drain()
    assert torch.equal(later[8].result()[1], eager[8][1])
