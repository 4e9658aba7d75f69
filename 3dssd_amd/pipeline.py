"""Batches in flight: the executor that serves `SingleStageDetector.network_forward`-style callers
(lib/modeling/single_stage_detector.py:115-125 runs ONE batch per `sess.run`; the reference has no
overlap of batches at all, lib/core/evaluator.py:94-135).

Why it exists: the layer-1 D-FPS is a 4 095-step dependent chain that keeps ONE compute unit per frame busy for
~3 ms, so one batch of 8 frames alone uses 8 of the 256 CUs most of the time.  Throughput comes from overlapping
the sampler chains of some frames with the chip-filling kernels (ball queries, grouped MLPs, distance matrices) of
others.  A PACKAGE is `coalesce` consecutive batches in one static input buffer, run through the backbone in one
pass (frames never interact: every kernel indexes its frame only).  Two ways to overlap packages:

  mode="staged" (default)  three HIP streams, software-pipelined inside the executor.  A package is TWO captured
      hipGraphs: stage A = input split + layer-1 D-FPS + centres (SABackbone.forward_staged up to its yield) on the
      sampler stream S; stage B = everything else on one of two main streams M0 / M1, behind an event.  S runs
      stage A of package k+1 / k+2 while M0 / M1 run stage B of packages k / k+1.  The captured graphs are linear
      chains (no helper-stream branch), so with the default one sampler + two main streams the executor needs 3
      hardware queues: ROCm's DEFAULT of 4 is enough, no environment variable.  Measured (profiles/r04_sweep_queues.txt): the same throughput on 4, 8 and 16
      queues, and at least that of the 16-slot mode in the short and in the long run.
  mode="slots"  `streams` slots, each a HIP stream with one captured hipGraph of the whole backbone (rounds 2-3).
      Needs as many hardware queues as slots (`request_hw_queues(16)` BEFORE the HIP runtime starts), and collapses
      when it does not get them (measured: profiles/r04_sweep_queues.txt).

    pipe = SAPipeline(arch, params, "cuda:0", batch=8, points=16384)
    t = [pipe.submit(b) for b in batches]          # returns at once; packages launch as they fill
    xyz, feat = t[0].result(copy=True)             # [B,256,3], [B,256,512]; launches a partly filled package first

Slots are reused round-robin: the tensors a ticket hands out are the slot's static output buffers and stay valid
until the slot's next round starts (`nslots x coalesce` further submits; use copy=True, or pass out=(xyz, feat) to
`submit`, to keep them longer); `result()` raises if the slot was already reused.  A package that is only partly
filled is launched by `flush()`, `drain()` or the first `result()` / `wait()` on one of its tickets: every slot has
captured graphs for `coalesce`, `coalesce / 2` and `coalesce / 4` batches (one memory pool: they never run at the same
time), the smallest one that holds the package runs, and parts it has beyond the fill are copies of the package's
first batch (real frames: all-zero frames would be the degenerate worst case of every data-dependent kernel).  Lifetime rule for inputs: `submit` copies the batch on an executor stream; a CUDA batch is
`record_stream`-ed there, so the caller may drop it right after `submit`.  fp16 range guard (csrc/mlp_act.h): every
round of a slot has its own flag word, zeroed on the slot's stream before the round and copied to pinned host memory
after it -- a ticket raises for ITS package only.  Nothing here is a collective: on a multi-GPU node every rank owns
one pipeline and its share of the frames (sharding.py).
"""
import ctypes
import gc
import os
import time
import warnings

import torch

from .backbone import SABackbone
from .utils import _native as N
from .utils.tf_ops import _tensor as T

DEFAULT_STREAMS = 16          # mode="slots"
DEFAULT_PACKAGES = 4          # mode="staged": packages in the ring (>= main streams + 1 so that stage A runs ahead)
MAIN_STREAMS = 2              # mode="staged"
_FLAG_RING = 4096


def request_hw_queues(n):
    """mode="slots" wants one hardware queue per slot; ROCm's default is 4.  GPU_MAX_HW_QUEUES is read when the HIP
    runtime starts, so this must run before the first CUDA call of the process (bench.py calls it first thing; a
    server sets the variable in its launcher).  Returns True when the setting can still take effect."""
    if torch.cuda.is_initialized():
        if os.environ.get("GPU_MAX_HW_QUEUES") != str(n):
            warnings.warn("GPU_MAX_HW_QUEUES=%s cannot take effect: the HIP runtime is already initialised (%s hardware "
                          "queues); streams beyond that share queues" % (n, os.environ.get("GPU_MAX_HW_QUEUES", "4")))
            return False
        return True
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(n))
    return os.environ["GPU_MAX_HW_QUEUES"] == str(n)


def hw_queues():
    """Hardware queues this process's HIP runtime was started with (what GPU_MAX_HW_QUEUES said then; 4 if unset)."""
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


class _Round:
    """One use of a slot: the package of up to `coalesce` batches that is launched together."""
    __slots__ = ("slot", "fill", "size", "launched", "event", "flag", "outs", "srcs", "redo")

    def __init__(self, slot, flag):
        self.slot, self.fill, self.size, self.launched, self.event, self.flag, self.outs = slot, 0, 0, False, None, flag, []
        self.srcs = []            # (part, tensor) of the batches whose copy into the slot's buffer waits for the launch
        self.redo = None          # (lists, tail outputs) of the package re-run in split bf16 after an fp16 range overflow


class Ticket:
    """One submitted batch.  `result()` blocks the host until THIS batch's package is complete (and launches the
    package first when it is still waiting for further batches)."""
    __slots__ = ("_round", "_part", "_out")

    def __init__(self, rnd, part, out):
        self._round, self._part, self._out = rnd, part, out

    def done(self):
        r = self._round
        return bool(r.launched and r.event.query())

    def wait(self):
        r = self._round
        if not r.launched:
            r.slot.pipe._launch(r.slot)
        r.event.synchronize()
        return self

    def _stale(self):
        return self._round.slot.round is not self._round

    def _views(self):
        r, B = self._round, self._round.slot.pipe.batch
        xl, fl, _ = r.redo[0] if r.redo is not None else r.slot.lists[r.size]
        return xl[-1][self._part * B:(self._part + 1) * B], fl[-1][self._part * B:(self._part + 1) * B]

    def result(self, copy=False):
        """(new_xyz [B,m,3], features [B,m,C]) of the backbone's last row.  With `out=` given at submit time those
        tensors are returned; otherwise this batch's part of the slot's static buffers (copy=True: clones)."""
        self.wait()
        r = self._round
        pipe = r.slot.pipe
        if pipe.uses_coop_samplers and N.lib().sa_coop_error_state(0) != 0:
            # the word is process-wide and sticky; only a network with multi-workgroup samplers (frames beyond 16384
            # points) consults it, and SAPipeline.clear_sampler_error() resets it once the cause is dealt with
            raise RuntimeError("SA backbone: a multi-workgroup sampler launch gave up waiting for its partner workgroups "
                               "(csrc/fps_coop.hip / ffps_fly.hip: such launches must stay on one stream) -- the results of "
                               "the packages in flight are invalid; SAPipeline.clear_sampler_error() clears the sticky word")
        if pipe.check_overflow and int(pipe._flags[r.flag]) != 0 and r.redo is None:
            # fp16 scales guard their operand range (csrc/mlp_act.h): this round's own word, already on the host.  The
            # package is run again with every scale in split bf16 (no range limit, ~1e-5 of fp32) while its input is still
            # in the slot -- a slower correct answer and a warning instead of an exception
            if not pipe.rerun_overflow or self._stale():
                raise FloatingPointError(
                    "SA backbone: an activation left the fp16 range (|x| > 65504, inf or NaN) in a scale evaluated in fp16 in "
                    "the package this batch ran in -- its results are invalid%s.  Use precision='bf16x3' for these weights."
                    % (" and its slot has been reused, so it cannot be re-run" if pipe.rerun_overflow else ""))
            pipe._rerun_bf16x3(r)
        if self._out is not None:
            return self._out
        if r.redo is not None:                                # tensors of the re-run: owned by the round, never reused
            xyz, feat = self._views()
            return (xyz.clone(), feat.clone()) if copy else (xyz, feat)
        if self._stale():
            raise RuntimeError("this ticket's slot has been reused by a later submit (%d batches in flight at most): "
                               "call result() earlier, or submit(..., out=...) / result(copy=True)"
                               % (pipe.nslots * pipe.coalesce))
        xyz, feat = self._views()
        if copy:
            st = r.slot.stream_b
            with torch.cuda.stream(st):
                xyz, feat = xyz.clone(), feat.clone()
            st.synchronize()
        return xyz, feat

    def detections(self, copy=False):
        """This batch's part of what the pipeline's `tail` produced behind the backbone (SAPipeline(tail=DetectionHead):
        boxes, scores, classes, NMS indices / counts -- modeling.single_stage_detector.DETECTION_KEYS), as views of the
        slot's static tensors (copy=True: clones).  Same validity rules as result()."""
        self.result()                                         # waits, and raises what result() raises
        r = self._round
        T.require(r.slot.pipe.tail is not None, "this pipeline has no tail (SAPipeline(tail=...))")
        if self._stale():
            raise RuntimeError("this ticket's slot has been reused by a later submit")
        B, p = r.slot.pipe.batch, self._part
        if r.redo is not None:
            return {k: (v[p * B:(p + 1) * B].clone() if copy else v[p * B:(p + 1) * B]) for k, v in r.redo[1].items()}
        out = {k: v[p * B:(p + 1) * B] for k, v in r.slot.extras[r.size].items()}
        if copy:
            st = r.slot.stream_b
            with torch.cuda.stream(st):
                out = {k: v.clone() for k, v in out.items()}
            st.synchronize()
        return out

    def all_outputs(self):
        """(xyz_list, feature_list, fps_idx_list) of this batch, as SABackbone.forward returns them (views of the slot's
        static buffers)."""
        self.wait()
        B, p = self._round.slot.pipe.batch, self._part
        cut = lambda t: None if t is None else t[p * B:(p + 1) * B]
        if self._round.redo is not None:
            return tuple([cut(t) for t in lst] for lst in self._round.redo[0])
        if self._stale():
            raise RuntimeError("this ticket's slot has been reused by a later submit")
        return tuple([cut(t) for t in lst] for lst in self._round.slot.lists[self._round.size])


class _Slot:
    # graphs: {package size (batches): (stage-A graph or None, stage-B / whole graph)}; lists: {size: forward()'s lists}
    __slots__ = ("pipe", "index", "stream_a", "stream_b", "inp", "overflow", "graphs", "lists", "round", "last_event",
                 "copy_batches", "src_ptrs", "inp_ptr", "extras", "mid")


class SAPipeline:
    def __init__(self, arch, params, device="cuda:0", batch=8, points=16384, channels=4, streams=None,
                 graphs=True, max_translate_range=(-3.0, -2.0, -3.0), aggregation_sa_feature=True, net=None,
                 precision=None, check_overflow=True, coalesce=1, mode="staged", timeline=False, linear_graphs=None,
                 main_streams=MAIN_STREAMS, sampler_streams=1, tail=None, rerun_overflow=True):
        """arch / params as for SABackbone.  mode / coalesce: module docstring.  `streams`: the number of slots --
        packages in the ring for mode="staged" (default 4), slots = HIP streams for mode="slots" (default 16).
        graphs=False launches eagerly on the same streams (same throughput with large packages, more host work).
        timeline=True records HIP timing events around every package (`timeline()`).  linear_graphs=True issues the
        F-FPS || D-FPS launch of the 'FS' layers on the capturing stream instead of a helper-stream branch, so that a
        captured graph is one linear chain and its replay needs no internal branch stream -- the default for
        mode="staged" (with branches the three streams + two branch streams share ROCm's default 4 hardware queues and
        block one another: 10.8 k instead of 16.2 k frames/s, profiles/r04_sweep_queues.txt); mode="slots" defaults to
        the branch form (15.7 k against 14.6 k on 16 queues).  main_streams / sampler_streams: streams the two stages
        alternate between (2 / 1 by default: a second sampler stream was measured).
        tail: a callable(lists) -> {name: tensor [batch x package, ...]} issued right behind the backbone on the same stream
        (captured into the stage-B graph): what follows the backbone in the caller's network, e.g. the detection head +
        decode + NMS (modeling.single_stage_detector.DetectionHead: `tail=lambda net: DetectionHead(net.variables, ...)`
        is also accepted -- a factory called with the pipeline's network).  It must only launch on the current stream and
        allocate shapes that depend on the package size alone; tickets return its outputs through detections().
        rerun_overflow: a package whose fp16 scales raised the range flag is run again, eagerly, with every scale in split
        bf16 (a second packing of the same parameters, built on first use) and its tickets return THAT result with a
        RuntimeWarning; False (or a slot already reused): FloatingPointError as before."""
        self.device = torch.device(device)
        T.require(self.device.type == "cuda", "SAPipeline needs a GPU: the HIP path has no CPU fallback")
        T.require(mode in ("staged", "slots"), "SAPipeline mode must be 'staged' or 'slots'")
        N.lib()
        if linear_graphs is None:
            linear_graphs = mode == "staged"
        self.linear_graphs = bool(linear_graphs)
        self.n_main = max(1, int(main_streams))
        # frames above 16384 points: their multi-workgroup sampler must stay on ONE stream (csrc/fps_coop.hip)
        self.n_samp = 1 if int(points) > 16384 else max(1, int(sampler_streams))
        # a caller's network: its multi-workgroup samplers (csrc/fps_coop.hip beyond 16384 points, csrc/ffps_fly.hip)
        # spin on partner workgroups and must never be in flight on two streams -- the layer-2 sampler runs on the
        # alternating main streams here, so a network built with ffps_fly is refused; the layer-1 sampler of large frames
        # runs on the ONE sampler stream of mode="staged" only
        self.fly_stage = bool(net is not None and net.settings.get("ffps_fly"))
        if net is not None:
            T.require(not net.settings.get("ffps_fly") or mode == "staged",
                      "SAPipeline(mode='slots') cannot run a network built with ffps_fly=True: the on-the-fly F-FPS needs all its "
                      "launches on one stream; mode='staged' gives the layer-2 sampler a stage and a stream of its own")
            T.require(not net.settings.get("coop_capture") or (mode == "staged" and int(points) > 16384),
                      "coop_capture is the staged executor's own setting for frames of more than 16384 points")
        self.net = net if net is not None else SABackbone(arch, params, self.device, max_translate_range,
                                                           aggregation_sa_feature, precision,
                                                           dfps_side_stream=5 if linear_graphs else None,
                                                           coop_capture=(mode == "staged" and bool(graphs) and int(points) > 16384))
        if net is not None:                           # what the caller's network really does, not what was asked for here
            self.linear_graphs = net.settings.get("dfps_side_stream") == 5
        if tail is not None and getattr(tail, "_is_tail_factory", False):
            tail = tail(self.net)
        self.tail = tail
        self.check_overflow = bool(check_overflow)
        # multi-workgroup samplers (csrc/fps_coop.hip) run for frames of more than 16384 points only (ffps_fly is refused above)
        self.uses_coop_samplers = int(points) > 16384
        self.rerun_overflow = bool(rerun_overflow)
        self._safe_net = None
        self._net_args = (arch, max_translate_range, aggregation_sa_feature)
        self.reruns = 0
        self.mode = mode
        self.batch, self.points, self.channels = int(batch), int(points), int(channels)
        if streams is None:
            streams = DEFAULT_PACKAGES if mode == "staged" else DEFAULT_STREAMS
        self.nslots = max(1, int(streams))
        self.coalesce = max(1, int(coalesce))
        self.sizes = sorted({self.coalesce, max(1, self.coalesce // 2), max(1, self.coalesce // 4)})
        self.graphs = bool(graphs)
        self.record_timeline = bool(timeline)
        self._timeline = []
        # host-side diagnostics: `host_trace` = a list the issuing calls append (call site, perf_counter_ns) to, or None;
        # `_ev_pool` = events created ahead of time (reserve_events) so that a measured region creates none
        self.host_trace = None
        self._ev_pool = []
        self._idle_events = []
        self._next = 0
        self.submitted = 0
        self._part_bytes = self.batch * self.points * self.channels * 4
        self._flags = torch.zeros(_FLAG_RING, dtype=torch.int32).pin_memory()
        self._flag_next = 0
        T.require(not (mode == "slots" and self.graphs and self.points > 16384),
                  "SAPipeline(mode='slots', graphs=True) cannot serve frames of more than 16384 points: their layer-1 sampler "
                  "is the multi-workgroup kernel, whose captured (plain) launches must all be issued on ONE stream "
                  "(csrc/fps_coop.hip) -- use mode='staged', or graphs=False")
        if mode == "slots" and self.nslots > hw_queues():
            warnings.warn("SAPipeline(mode='slots') with %d slots on %d hardware queues: slots beyond the queue count "
                          "serialise (call pipeline.request_hw_queues(%d) before the first CUDA call, or use "
                          "mode='staged')" % (self.nslots, hw_queues(), self.nslots))
        with torch.cuda.device(self.device):
            self._build()

    # ------------------------------------------------------------------------------------------------ set-up
    def _build(self):
        dev = self.device
        shape = (self.batch * self.coalesce, self.points, self.channels)
        if self.mode == "staged":
            self.sampler_streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_samp)]
            self.sampler_stream = self.sampler_streams[0]
            self.main_streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_main)]
            # a network with the on-the-fly F-FPS (measurement builds): FOUR stages per package -- A (layer-1 sampler) on
            # the sampler stream, B1 (layer 1) on a main stream, F (layer-2 sampler: every ffps_fly launch of the process)
            # on ONE stream of its own, B2 (the rest) on the main stream again
            self.fly_stream = torch.cuda.Stream(device=dev) if self.fly_stage else None
        self.slots = []
        for i in range(self.nslots):
            s = _Slot()
            s.pipe, s.index, s.graphs, s.lists, s.extras, s.mid = self, i, {}, {}, {}, {}
            if self.mode == "staged":
                s.stream_a, s.stream_b = self.sampler_streams[i % self.n_samp], self.main_streams[i % self.n_main]
            else:
                s.stream_a = s.stream_b = torch.cuda.Stream(device=dev)
            s.inp = torch.zeros(shape, dtype=torch.float32, device=dev)
            s.inp_ptr = s.inp.data_ptr()
            s.copy_batches = N.lib().sa_copy_batches
            s.src_ptrs = (ctypes.c_void_p * max(self.coalesce, 1))()
            s.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
            s.round, s.last_event = None, None
            self.slots.append(s)
        if not self.graphs:
            return
        # caches (packed weights, identity indices, helper streams) are filled by eager runs BEFORE any capture: a
        # tensor first created inside a capture would live in that graph's private pool
        warm = torch.zeros(shape, dtype=torch.float32, device=dev)
        warm[:, :, :3] = torch.rand((shape[0], self.points, 3), device=dev) * 20.0
        for size in self.sizes:
            for _ in range(2):
                lists = self.net(warm[:size * self.batch])
                if self.tail is not None:
                    self.tail(lists)
        torch.cuda.synchronize(dev)
        for s in self.slots:
            s.inp.copy_(warm)
        torch.cuda.synchronize(dev)
        # No garbage collection INSIDE a capture: a collection that Python starts between two launches of a captured pass
        # finalises whatever cyclic garbage exists -- an earlier pipeline's graphs, events and streams included -- and
        # destroying those while a stream of this process is capturing aborts the process inside the HIP runtime (seen as
        # an intermittent "Fatal Python error: Aborted ... Garbage-collecting" when several pipelines are built in one
        # process).  torch.cuda.graph collects once on entry; the automatic collector is off until the captures are done.
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            self._capture_all()
        finally:
            if gc_was_on:
                gc.enable()
        torch.cuda.synchronize(dev)
        # every graph replayed once as part of the set-up (the first replay uploads the executable graph)
        self._first_replays()

    def _capture_all(self):
        for s in self.slots:
            pool = None
            for size in reversed(self.sizes):                 # the largest first: the smaller ones fit into its pool
                view = s.inp[:size * self.batch]
                kw = {} if pool is None else {"pool": pool}
                with self._flag_word(s):
                    if self.mode == "staged":
                        gen = self.net.forward_staged(view, self.fly_stage)
                        ga = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(ga, stream=s.stream_a, **kw):
                            next(gen)
                        pool = pool or ga.pool()
                        if self.fly_stage:                            # B1 on the main stream, F on the fly stream
                            mid = []
                            for st in (s.stream_b, self.fly_stream):
                                g = torch.cuda.CUDAGraph()
                                with torch.cuda.graph(g, stream=st, pool=pool):
                                    next(gen)
                                mid.append((st, g))
                            s.mid[size] = mid
                        gb = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gb, stream=s.stream_b, pool=pool):
                            try:
                                next(gen)
                                raise RuntimeError("forward_staged yielded twice")
                            except StopIteration as e:
                                lists = e.value
                            if self.tail is not None:
                                s.extras[size] = self.tail(lists)
                        s.graphs[size] = (ga, gb)
                    else:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=s.stream_b, **kw):
                            lists = self.net(view)
                            if self.tail is not None:
                                s.extras[size] = self.tail(lists)
                        pool = pool or g.pool()
                        s.graphs[size] = (None, g)
                s.lists[size] = lists

    def _first_replays(self):
        dev = self.device
        for s in self.slots:
            for size in self.sizes:
                ga, gb = s.graphs[size]
                if ga is not None:
                    with torch.cuda.stream(s.stream_a):
                        ga.replay()
                    torch.cuda.synchronize(dev)
                for st, g in s.mid.get(size, []):
                    with torch.cuda.stream(st):
                        g.replay()
                    torch.cuda.synchronize(dev)
                with torch.cuda.stream(s.stream_b):
                    gb.replay()
                torch.cuda.synchronize(dev)

    class _flag_word:
        """While a slot's kernels are enqueued / captured, the network's fp16 range flag IS the slot's own word."""

        def __init__(self, slot):
            self.v, self.slot = slot.pipe.net.variables, slot

        def __enter__(self):
            self.saved, self.v.overflow = self.v.overflow, self.slot.overflow

        def __exit__(self, *exc):
            self.v.overflow = self.saved

    # ------------------------------------------------------------------------------------------------ diagnostics
    def _stamp(self, site):
        tr = self.host_trace
        if tr is not None:
            tr.append((site, time.perf_counter_ns()))

    def reserve_events(self, n):
        """Create `n` HIP events now (timing-capable: usable for either purpose), to be handed out by the next launches
        instead of creating them on the way."""
        with torch.cuda.device(self.device):
            while len(self._ev_pool) < n:
                self._ev_pool.append(torch.cuda.Event(enable_timing=True))

    def _event(self, timing):
        if self._ev_pool:
            return self._ev_pool.pop()
        return torch.cuda.Event(enable_timing=timing)

    # ------------------------------------------------------------------------------------------------ use
    def submit(self, batch, out=None, sync_source=True, defer_copy=False):
        """Enqueue one batch [B, points, channels] fp32 (device tensor; a pinned host tensor is copied
        asynchronously).  Returns immediately.  sync_source=False skips the event that orders the copy behind the
        stream that produced `batch` (for inputs known to be complete, e.g. a resident pool).  defer_copy=True (with
        sync_source=False, a dense 16-byte-aligned device batch) only NOTES the source: it is copied when its PACKAGE is
        launched -- one launch for the whole package instead of one per batch -- so the caller promises to leave the
        tensor's CONTENTS unchanged until the package has been launched (the submit that fills it, flush(), or a
        result()); the executor keeps the tensor alive, the reference may be dropped at once.  Without it the copy is
        enqueued by submit itself, as it always was (the deferred form used to be implicit).
        out = (xyz [B,m,3], feat [B,m,C]): the results are additionally copied there on the slot's stream.
        With coalesce > 1 the package is launched by the submit that fills it (or by flush / drain / result)."""
        T.require(isinstance(batch, torch.Tensor) and tuple(batch.shape) == (self.batch, self.points, self.channels),
                  "SAPipeline.submit expects a [%d,%d,%d] tensor" % (self.batch, self.points, self.channels))
        T.require(batch.dtype == torch.float32, "SAPipeline.submit expects fp32 (got %s)" % batch.dtype)
        self._stamp("submit:enter")
        s = self.slots[self._next]
        r = s.round
        if r is None or r.launched:          # a new round of this slot: earlier tickets of the slot are now stale
            r = s.round = _Round(s, self._flag_next)
            self._flag_next = (self._flag_next + 1) % _FLAG_RING
        part = r.fill
        st = s.stream_a
        fast = (defer_copy and batch.is_cuda and batch.is_contiguous() and not sync_source and (batch.data_ptr() & 15) == 0 and
                (self._part_bytes & 15) == 0)
        if fast:
            # resident, dense, already complete: the submit only NOTES the source (the round keeps the tensor alive); the
            # package is filled by ONE sa_copy_batches launch in front of its sampling stage (_launch) -- 16 copy launches
            # and 25 us of host time each sat there before round 5
            T.require(batch.device == self.device, "batch lives on %s, the pipeline on %s" % (batch.device, self.device))
            r.srcs.append((part, batch))
        else:
            dst = s.inp[part * self.batch:(part + 1) * self.batch]
            with torch.cuda.device(self.device):
                if batch.is_cuda:
                    T.require(batch.device == self.device, "batch lives on %s, the pipeline on %s" % (batch.device, self.device))
                    if sync_source:
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(self.device))
                        st.wait_event(ev)
                    batch.record_stream(st)   # the caller may drop `batch` now: its block is not reused before the copy ran
                with torch.cuda.stream(st):
                    if batch.is_cuda and batch.stride(2) == 1:
                        N.copy_blocks([(batch, dst, self.batch, self.points, self.channels)])
                    else:
                        dst.copy_(batch, non_blocking=True)
        self._stamp("submit:copy_blocks")
        if out is not None:
            for o in out:
                if o.is_cuda:
                    o.record_stream(s.stream_b)
            r.outs.append((part, out))
        t = Ticket(r, part, out)
        self.submitted += 1
        r.fill += 1
        if r.fill == self.coalesce:
            self._launch(s)
        return t

    def _launch(self, s):
        """Run slot `s` over what its input buffer holds: the smallest captured package size that holds the fill (parts
        beyond the fill = copies of the first batch); the slot the next submit fills is the one after it."""
        r = s.round
        if r is None or r.launched:
            return
        B = self.batch
        tl = self.record_timeline
        stamp = self._stamp
        stamp("launch:enter")
        size = r.size = min(z for z in self.sizes if z >= r.fill)
        view = s.inp[:size * B]
        with torch.cuda.device(self.device):
            a, b = s.stream_a, s.stream_b
            marks = []
            if r.srcs:                            # the deferred input copies of this package: one launch for a run of parts
                srcs, r.srcs = r.srcs, []
                i = 0
                while i < len(srcs):
                    j = i
                    while j + 1 < len(srcs) and srcs[j + 1][0] == srcs[j][0] + 1 and j + 1 - i < 32:     # sa_copy_batches: <= 32 per launch
                        j += 1
                    n = j - i + 1
                    for q in range(n):
                        s.src_ptrs[q] = srcs[i + q][1].data_ptr()
                    N.check(s.copy_batches(n, s.src_ptrs, s.inp_ptr + srcs[i][0] * self._part_bytes, self._part_bytes, a.cuda_stream),
                            "copy_batches")
                    i = j + 1
                for _part, t in srcs:
                    t.record_stream(a)            # the caller's tensor may be freed now: its block is not reused before the copy ran
                stamp("launch:copy_batches")
            with torch.cuda.stream(a):
                if tl:
                    marks.append(self._mark(a))
                    stamp("launch:mark_reached")
                if r.fill < size:
                    k = size - r.fill
                    s.inp[r.fill * B:size * B].view(k, B, self.points, self.channels).copy_(
                        s.inp[:B].unsqueeze(0).expand(k, B, self.points, self.channels))
            ga, gb = s.graphs.get(size, (None, None))
            staged = self.mode == "staged"
            gen = None
            if staged:
                if s.last_event is not None:
                    a.wait_event(s.last_event)            # stage B of the slot's previous round still reads stage A's outputs
                    stamp("launch:wait_prev_round")
                with torch.cuda.stream(a):
                    s.overflow.zero_()                    # the round's fp16 range word, in front of BOTH stages (b waits for a)
                    if ga is not None:
                        ga.replay()
                    else:
                        gen = self.net.forward_staged(view, self.fly_stage)
                        with self._flag_word(s):
                            next(gen)
                    stamp("launch:stage_A")
                    ev = self._event(tl)
                    ev.record(a)
                    if tl:
                        marks.append(ev)
                if self.fly_stage:                                # B1 on the main stream, F on the one fly stream, each behind an event
                    for j, st in enumerate((b, self.fly_stream)):
                        st.wait_event(ev)
                        with torch.cuda.stream(st):
                            if ga is not None:
                                s.mid[size][j][1].replay()
                            else:
                                with self._flag_word(s):
                                    next(gen)
                            ev = self._event(False)
                            ev.record(st)
                b.wait_event(ev)
                stamp("launch:event_A_to_B")
            with torch.cuda.stream(b):
                if not staged:
                    s.overflow.zero_()
                stamp("launch:zero_flag")
                if gb is not None:
                    gb.replay()
                else:
                    with self._flag_word(s):
                        if gen is not None:
                            try:
                                next(gen)
                                raise RuntimeError("forward_staged yielded twice")
                            except StopIteration as e:
                                s.lists[size] = e.value
                        else:
                            s.lists[size] = self.net(view)
                        if self.tail is not None:
                            s.extras[size] = self.tail(s.lists[size])
                xl, fl, _ = s.lists[size]
                for part, (ox, of) in r.outs:
                    N.copy_blocks([(xl[-1][part * B:(part + 1) * B], ox, B, xl[-1].shape[1], 3),
                                   (fl[-1][part * B:(part + 1) * B], of, B, fl[-1].shape[1], fl[-1].shape[2])])
                stamp("launch:stage_B")
                self._flags[r.flag:r.flag + 1].copy_(s.overflow, non_blocking=True)
                stamp("launch:flag_copy")
                r.event = self._event(tl)
                r.event.record(b)
                stamp("launch:event_done")
            if tl:
                marks.append(r.event)
                self._timeline.append((s.index, r.fill, marks))
                if len(self._timeline) > 4096:            # a caller that never collects: keep the newest
                    del self._timeline[:2048]
        s.last_event = r.event
        r.launched = True
        if s is self.slots[self._next]:
            self._next = (self._next + 1) % self.nslots

    def _mark(self, stream):
        ev = self._event(True)
        ev.record(stream)
        return ev

    def timeline(self, base, clear=True):
        """[(slot, batches, [ms since `base` ...])] of the packages launched since the last call (timeline=True):
        reached-by-its-stream, (stage A complete,) package complete -- device-side times from HIP events.  `base` is a
        timing event recorded before them; call after a synchronize.  base=None only discards what was recorded."""
        out = [] if base is None else [(i, fill, [round(base.elapsed_time(e), 3) for e in marks])
                                      for i, fill, marks in self._timeline]
        if clear:
            self._timeline = []
        return out

    def flush(self):
        """Launch the package that is waiting for more batches, if any (coalesce > 1)."""
        s = self.slots[self._next]
        if s.round is not None and not s.round.launched:
            self._launch(s)

    def run_alone(self, batch):
        """One batch by itself on slot 0 (latency measurements): submit + launch + wait."""
        self.drain()
        self._next = 0
        return self.submit(batch).wait()

    def drain(self):
        self.flush()
        for s in self.slots:
            s.stream_a.synchronize()
            s.stream_b.synchronize()

    def wait_idle(self):
        """drain() without a spinning host thread: one blocking HIP event (hipEventBlockingSync) per executor stream,
        recorded behind everything issued so far and waited for asleep.  A host that shares its cores -- or runs under a
        CPU quota -- keeps them free while the GPU works."""
        self.flush()
        streams = []
        for s in self.slots:
            for st in (s.stream_a, s.stream_b):
                if all(st is not t for t in streams):
                    streams.append(st)
        if len(self._idle_events) < len(streams):
            with torch.cuda.device(self.device):
                self._idle_events = [torch.cuda.Event(blocking=True) for _ in streams]
        for st, ev in zip(streams, self._idle_events):
            ev.record(st)
        for ev in self._idle_events[:len(streams)]:
            ev.synchronize()

    def streams_used(self):
        """HIP streams the executor issues on (the helper branch inside the captured graphs not counted)."""
        return len({id(s.stream_a) for s in self.slots} | {id(s.stream_b) for s in self.slots})

    def forward_eager(self, batch):
        """The same network, eager launches on the current stream (the reference result of the tests)."""
        return self.net(batch)

    def clear_sampler_error(self):
        """Reset the process-wide sticky word a multi-workgroup sampler raises when it loses its partner workgroups
        (-> the value it held).  Call after the packages in flight have been drained and discarded: tickets of later
        packages are valid again."""
        self.drain()
        return int(N.lib().sa_coop_error_state(1))

    def _rerun_bf16x3(self, r):
        """The package of round `r` again, eager launches on its main stream, every grouped-MLP scale in split bf16: the
        slot's input buffer still holds the package (the caller checked that the slot has not been reused)."""
        s = r.slot
        if self._safe_net is None:
            arch, mtr, agg = self._net_args
            T.require(arch is not None, "SAPipeline was built around a caller's network without `arch`: it cannot re-run an "
                                        "overflowed package (pass rerun_overflow=False, or precision='bf16x3')")
            self._safe_net = SABackbone(arch, self.net.variables.params, self.device, mtr, agg, precision="bf16x3",
                                        dfps_side_stream=5)
        warnings.warn("SAPipeline: an activation left the fp16 range in the package of %d batch(es) on slot %d; the package "
                      "was run again in split bf16 (about three times the MLP time) -- consider precision='bf16x3' for these "
                      "weights" % (r.fill, s.index), RuntimeWarning, stacklevel=3)
        B = self.batch
        with torch.cuda.device(self.device), torch.cuda.stream(s.stream_b):
            view = s.inp[:r.size * B].clone()                 # the re-run must not race the slot's next round
            lists = self._safe_net(view)
            extras = self.tail(lists) if self.tail is not None else None
            for part, (ox, of) in r.outs:                     # out= tensors of the submits: overwritten with the good result
                N.copy_blocks([(lists[0][-1][part * B:(part + 1) * B], ox, B, lists[0][-1].shape[1], 3),
                               (lists[1][-1][part * B:(part + 1) * B], of, B, lists[1][-1].shape[1], lists[1][-1].shape[2])])
        s.stream_b.synchronize()
        self._safe_net.raise_if_overflow()                    # split bf16 has no range guard to trip; a NaN input still raises
        r.redo = (lists, extras)
        self.reruns += 1

    def tail_eager(self, batch):
        """(lists, tail outputs) of one batch, eager launches on the current stream."""
        lists = self.net(batch)
        return lists, (self.tail(lists) if self.tail is not None else None)
