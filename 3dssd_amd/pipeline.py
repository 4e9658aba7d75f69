"""Batches in flight: the executor that serves `SingleStageDetector.network_forward`-style callers
(lib/modeling/single_stage_detector.py:115-125 runs ONE batch per `sess.run`; the reference has no
overlap of batches at all, lib/core/evaluator.py:94-135).

Why it exists: the layer-1 D-FPS is a 4 095-step dependent chain that keeps ONE compute unit per frame busy for
~3 ms, so one batch of 8 frames alone uses 8 of the 256 CUs most of the time.  Throughput comes from overlapping
the chains of different batches: N HIP streams, each with its own captured hipGraph of the whole backbone, its own
STATIC input buffer and its own intermediate / output buffers.  `submit(batch)` copies the batch into the next
slot's input buffer (one `sa_copy_blocks` launch on the slot's stream) and replays that slot's graph; it returns
a `Ticket` whose `result()` waits for that batch only.

    pipe = SAPipeline(arch, params, "cuda:0", batch=8, points=16384)
    t = [pipe.submit(b) for b in batches]          # up to `streams` batches run concurrently
    xyz, feat = t[0].result(copy=True)             # [B,256,3], [B,256,512]

Slots are reused round-robin: the tensors a ticket hands out are the slot's static output buffers and stay valid
until `streams` further submits (use copy=True, or pass out=(xyz, feat) to `submit`, to keep them longer);
`result()` raises if the slot was already reused.  Nothing here is a collective: on a multi-GPU node every rank
owns one pipeline and its share of the frames (sharding.py).

`coalesce=C` (default 1): a slot takes C consecutive batches before it is launched -- its input buffer holds C x B
frames and its graph is the backbone over all of them, so C x B sampler chains (one CU each) share one trip through the
slot's hardware queue.  The 16 hardware queues bound the number of chains in flight, not the kernels: at B = 8 the
steady rate goes 10.7 -> 13.2 (C = 2) -> 14.9 k frames/s (C = 4) while a replay alone takes 4.4 -> 4.8 -> 5.7 ms
(DESIGN.md section 5).  Frames never interact (every kernel indexes its frame only), so a batch's result does not
depend on which batches it shares a replay with; a slot that is only partly filled is launched by `flush()`,
`drain()` or the first `result()` / `wait()` on one of its tickets, its unfilled parts computing on stale frames.
"""
import os

# The slots live on different HIP streams; the ROCm default of 4 hardware queues would serialise them (measured:
# 16 queues = 1.6x the throughput of 4).  Only effective when set before the HIP runtime starts, hence at import.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch

from .backbone import SABackbone
from .utils import _native as N
from .utils.tf_ops import _tensor as T

DEFAULT_STREAMS = 16


class Ticket:
    """One submitted batch.  `result()` blocks the host until THIS batch is complete (and launches its slot first
    when the slot is still waiting for further batches to coalesce)."""
    __slots__ = ("_slot", "_seq", "_part", "_out")

    def __init__(self, slot, seq, part, out):
        self._slot, self._seq, self._part, self._out = slot, seq, part, out

    def _launched(self):
        s = self._slot
        return s.seq > self._seq or (s.seq == self._seq and s.launched)

    def done(self):
        return self._launched() and (self._slot.seq > self._seq or self._slot.event.query())

    def wait(self):
        s = self._slot
        if s.seq == self._seq and not s.launched:
            s.pipe._launch(s)
        s.event.synchronize()        # (a slot that was reused since: its newer event is later on the same stream)
        return self

    def _views(self):
        s, B = self._slot, self._slot.pipe.batch
        return s.out_xyz[self._part * B:(self._part + 1) * B], s.out_feat[self._part * B:(self._part + 1) * B]

    def result(self, copy=False):
        """(new_xyz [B,m,3], features [B,m,C]) of the backbone's last row.  With `out=` given at submit time those
        tensors are returned; otherwise this batch's part of the slot's static buffers (copy=True: clones)."""
        self.wait()
        pipe = self._slot.pipe
        if pipe.check_overflow:
            # fp16 scales guard their operand range (csrc/mlp_act.h): one 4-byte read per batch, after completion
            pipe.net.raise_if_overflow()
        if self._out is not None:
            return self._out
        if self._slot.seq != self._seq:
            raise RuntimeError("this ticket's slot has been reused by a later submit (%d batches in flight at most): "
                               "call result() earlier, or submit(..., out=...) / result(copy=True)"
                               % (self._slot.pipe.nslots * self._slot.pipe.coalesce))
        xyz, feat = self._views()
        if copy:
            with torch.cuda.stream(self._slot.stream):
                xyz, feat = xyz.clone(), feat.clone()
            self._slot.stream.synchronize()
        return xyz, feat

    def all_outputs(self):
        """(xyz_list, feature_list, fps_idx_list) of this batch, as SABackbone.forward returns them (views of the slot's
        static buffers)."""
        self.wait()
        if self._slot.seq != self._seq:
            raise RuntimeError("this ticket's slot has been reused by a later submit")
        B, p = self._slot.pipe.batch, self._part
        cut = lambda t: None if t is None else t[p * B:(p + 1) * B]
        return tuple([cut(t) for t in lst] for lst in self._slot.lists)


class _Slot:
    __slots__ = ("pipe", "stream", "inp", "graph", "out_xyz", "out_feat", "lists", "seq", "event", "launched", "outs")


class SAPipeline:
    def __init__(self, arch, params, device="cuda:0", batch=8, points=16384, channels=4, streams=DEFAULT_STREAMS,
                 graphs=True, max_translate_range=(-3.0, -2.0, -3.0), aggregation_sa_feature=True, net=None,
                 precision=None, check_overflow=True, coalesce=1):
        """arch / params as for SABackbone.  `streams` slots, each a HIP stream + (graphs=True) a captured hipGraph of
        net(slot input); a slot's input holds `coalesce` batches (module docstring).  graphs=False launches eagerly on the slot's stream (frames whose layer-1 sampler is the
        cooperative multi-workgroup kernel -- n > 16384 -- cannot be captured)."""
        self.device = torch.device(device)
        T.require(self.device.type == "cuda", "SAPipeline needs a GPU: the HIP path has no CPU fallback")
        N.lib()
        self.net = net if net is not None else SABackbone(arch, params, self.device, max_translate_range,
                                                           aggregation_sa_feature, precision)
        self.check_overflow = bool(check_overflow)
        self.batch, self.points, self.channels = int(batch), int(points), int(channels)
        self.nslots = max(1, int(streams))
        self.coalesce = max(1, int(coalesce))
        self._fill = 0
        self.graphs = bool(graphs)
        self._next = 0
        self.submitted = 0
        with torch.cuda.device(self.device):
            self._build()

    # ------------------------------------------------------------------------------------------------ set-up
    def _build(self):
        dev = self.device
        shape = (self.batch * self.coalesce, self.points, self.channels)
        self.slots = []
        for _ in range(self.nslots):
            s = _Slot()
            s.pipe, s.stream, s.seq, s.graph = self, torch.cuda.Stream(device=dev), -1, None
            s.inp = torch.zeros(shape, dtype=torch.float32, device=dev)
            s.event = torch.cuda.Event()
            s.out_xyz = s.out_feat = s.lists = None
            s.launched, s.outs = True, []
            self.slots.append(s)
        if not self.graphs:
            return
        # caches (packed weights, identity indices, helper streams) are filled by eager runs BEFORE any capture: a
        # tensor first created inside a capture would live in that graph's private pool
        warm = torch.zeros(shape, dtype=torch.float32, device=dev)
        warm[:, :, :3] = torch.rand((shape[0], self.points, 3), device=dev) * 20.0
        for _ in range(2):
            self.net(warm)
        torch.cuda.synchronize(dev)
        for s in self.slots:
            s.inp.copy_(warm)
        torch.cuda.synchronize(dev)
        for s in self.slots:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s.stream):
                xl, fl, il = self.net(s.inp)
            s.graph, s.lists = g, (xl, fl, il)
            s.out_xyz, s.out_feat = xl[-1], fl[-1]
        torch.cuda.synchronize(dev)
        # every graph replayed once as part of the set-up (first replay uploads the executable graph)
        for s in self.slots:
            with torch.cuda.stream(s.stream):
                s.graph.replay()
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------------------------------------ use
    def submit(self, batch, out=None, sync_source=True):
        """Enqueue one batch [B, points, channels] fp32 (device tensor; a pinned host tensor is copied
        asynchronously).  Returns immediately.  sync_source=False skips the event that orders the slot's stream
        behind the stream that produced `batch` (for inputs known to be complete, e.g. a resident pool).
        out = (xyz [B,m,3], feat [B,m,C]): the results are additionally copied there on the slot's stream.
        With coalesce > 1 the slot is launched by the submit that fills it (or by flush / drain / result)."""
        T.require(isinstance(batch, torch.Tensor) and tuple(batch.shape) == (self.batch, self.points, self.channels),
                  "SAPipeline.submit expects a [%d,%d,%d] tensor" % (self.batch, self.points, self.channels))
        T.require(batch.dtype == torch.float32, "SAPipeline.submit expects fp32 (got %s)" % batch.dtype)
        s = self.slots[self._next]
        part = self._fill
        if part == 0:                       # a new round of this slot: earlier tickets of the slot are now stale
            s.seq, s.launched, s.outs = self.submitted, False, []
        dst = s.inp[part * self.batch:(part + 1) * self.batch]
        with torch.cuda.device(self.device):
            if batch.is_cuda:
                T.require(batch.device == self.device, "batch lives on %s, the pipeline on %s" % (batch.device, self.device))
                if sync_source:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    s.stream.wait_event(ev)
            with torch.cuda.stream(s.stream):
                if batch.is_cuda and batch.stride(2) == 1:
                    N.copy_blocks([(batch, dst, self.batch, self.points, self.channels)])
                else:
                    dst.copy_(batch, non_blocking=True)
        if out is not None:
            s.outs.append((part, out))
        t = Ticket(s, s.seq, part, out)
        self.submitted += 1
        self._fill += 1
        if self._fill == self.coalesce:
            self._launch(s)
        return t

    def _launch(self, s):
        """Replay (or run eagerly) slot `s` over whatever its input buffer holds; the slot the next submit fills is
        the one after it."""
        if s.launched:
            return
        with torch.cuda.device(self.device), torch.cuda.stream(s.stream):
            if s.graph is not None:
                s.graph.replay()
            else:
                xl, fl, il = self.net(s.inp)
                s.lists, s.out_xyz, s.out_feat = (xl, fl, il), xl[-1], fl[-1]
            B = self.batch
            for part, (ox, of) in s.outs:
                N.copy_blocks([(s.out_xyz[part * B:(part + 1) * B], ox, B, s.out_xyz.shape[1], 3),
                               (s.out_feat[part * B:(part + 1) * B], of, B, s.out_feat.shape[1], s.out_feat.shape[2])])
            s.event = torch.cuda.Event()
            s.event.record(s.stream)
        s.launched = True
        if s is self.slots[self._next]:
            self._next = (self._next + 1) % self.nslots
            self._fill = 0

    def flush(self):
        """Launch the slot that is waiting for more batches, if any (coalesce > 1)."""
        if self._fill:
            self._launch(self.slots[self._next])

    def run_alone(self, batch):
        """One batch by itself on slot 0 (latency measurements): submit + launch + wait."""
        self.flush()
        self._next = 0
        return self.submit(batch).wait()

    def drain(self):
        self.flush()
        for s in self.slots:
            s.stream.synchronize()

    def forward_eager(self, batch):
        """The same network, eager launches on the current stream (the reference result of the tests)."""
        return self.net(batch)
