"""Where does a step spend its time when 16 steps are in flight?

rocprofv3 serialises the streams (profiles/r02_trace16_summary.txt: 2 kernels in flight), so the regime the headline
number is measured in is invisible to it.  This tool appends a one-thread "stamp" launch (tools/microbench/stamp.hip:
tag + device wall clock) after every C-ABI call of the backbone -- also inside the captured graphs -- and compares,
call by call, the time from the previous stamp of the same step (queue wait + kernel) for one step alone and for the
bench's 16-stream regime.  The stamps add ~40 tiny dependent launches per step: read the columns as a distribution of
where the time goes, not as exact kernel times.

    python tools/stage_trace.py [--streams 16] [--steps 96] > gpurun_out/stage_trace.txt
"""
import argparse
import ctypes
import importlib
import os
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pkg(name):
    return importlib.import_module("3dssd_amd." + name)


def stamp_lib():
    src = os.path.join(ROOT, "tools", "microbench", "stamp.hip")
    so = os.path.join(ROOT, "tools", "microbench", "libstamp.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
    h = ctypes.CDLL(so)
    h.stamp.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_void_p]
    h.stamp.restype = ctypes.c_int
    return h


class StampProxy:
    """Wraps the ctypes library: every sa_* call is followed by a stamp on the stream the call was given."""
    def __init__(self, real, stamps, buf, cap):
        self._real, self._st, self._buf, self._cap = real, stamps, buf, cap
        self.gid = 0
        self.seq = 0
        self.labels = {}

    def mark(self, label, stream):
        self.seq += 1
        self.labels.setdefault(self.seq, label)
        self._st.stamp(self._buf.data_ptr(), self._cap, (self.gid << 16) | self.seq, stream)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("sa_") or name.endswith("_ws_bytes"):
            return fn

        def wrapped(*args):
            r = fn(*args)
            self.mark(name[3:], args[-1])
            return r
        return wrapped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=16)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=16384)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfgs, syn, native = pkg("configs"), pkg("synthetic"), pkg("utils._native")
    arch = cfgs.KITTI_3DSSD_ARCH
    net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)
    pts = torch.from_numpy(syn.kitti_like_batch(args.batch, n=args.points)).to(dev)
    for _ in range(2):
        net(pts)
    torch.cuda.synchronize()
    cap = 1 << 18
    buf = torch.zeros(2 + 2 * cap, dtype=torch.int64, device=dev)
    st = stamp_lib()
    proxy = StampProxy(native.lib(), st, buf, cap)
    native._LIB = proxy
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
    graphs = []
    for gid, s in enumerate(streams):
        proxy.gid, proxy.seq = gid, 0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st.stamp(buf.data_ptr(), cap, (gid << 16) | 0, torch.cuda.current_stream().cuda_stream)   # seq 0: the step starts
            net(pts)
        graphs.append(g)
    nseq = proxy.seq
    torch.cuda.synchronize()
    native._LIB = proxy._real

    def collect():
        torch.cuda.synchronize()
        h = buf.cpu().numpy()
        n = int(min(h[0], cap))
        tags, clk = h[2:2 + 2 * n:2], h[3:3 + 2 * n:2]
        buf.zero_()
        torch.cuda.synchronize()
        return tags >> 16, tags & 0xFFFF, clk

    # wall clock ticks per microsecond, against the host clock
    buf.zero_()
    st.stamp(buf.data_ptr(), cap, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    time.sleep(0.25)
    st.stamp(buf.data_ptr(), cap, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    _g, _s, clk = collect()
    ticks_per_us = float(clk[1] - clk[0]) / ((t1 - t0) * 1e6)

    def spans(gid_a, seq_a, clk_a):
        """per replay: array [nseq + 1] of stamp times; returns list of arrays (us)"""
        out = []
        for g in np.unique(gid_a):
            m = gid_a == g
            sq, ck = seq_a[m], clk_a[m]
            order = np.argsort(ck, kind="stable")
            sq, ck = sq[order], ck[order]
            starts = np.nonzero(sq == 0)[0]
            for a, b in zip(starts, list(starts[1:]) + [len(sq)]):
                if b - a != nseq + 1:
                    continue
                t = np.zeros(nseq + 1)
                t[sq[a:b]] = ck[a:b] / ticks_per_us
                out.append(t)
        return out

    # (a) one step alone
    for g in graphs:                                   # every graph once (first replay uploads it)
        g.replay()
    torch.cuda.synchronize()
    collect()
    for _ in range(6):
        with torch.cuda.stream(streams[0]):
            graphs[0].replay()
        streams[0].synchronize()
    alone = spans(*collect())
    # (b) the bench regime: steps round-robin over the streams
    def run(k):
        for i in range(k):
            j = i % len(streams)
            with torch.cuda.stream(streams[j]):
                graphs[j].replay()
    run(2 * len(streams))
    torch.cuda.synchronize()
    collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    g_a, s_a, c_a = collect()
    loaded = spans(g_a, s_a, c_a)
    # drop the ramp: keep replays that started after the first 1/4 and ended before the last 1/4 of the window
    lo, hi = c_a.min() / ticks_per_us, c_a.max() / ticks_per_us
    mid = [t for t in loaded if t[0] > lo + 0.2 * (hi - lo) and t[-1] < hi - 0.15 * (hi - lo)] or loaded

    def per_call(reps):
        d = np.stack([np.diff(t) for t in reps])       # [replays, nseq]
        return np.median(d, 0), np.median([t[-1] - t[0] for t in reps])

    da, ta = per_call(alone)
    dl, tl = per_call(mid)
    print("streams %d, steps %d (with stamps): %.3f ms per step, %.0f frames/s; one step alone %.3f ms, under load %.3f ms "
          "(%d replays kept of %d); wall clock %.1f ticks/us"
          % (len(streams), args.steps, wall_ms / args.steps, args.batch * args.steps / wall_ms * 1e3, ta / 1e3, tl / 1e3,
             len(mid), len(loaded), ticks_per_us))
    print("%-4s %-34s %10s %12s %8s %8s" % ("seq", "call", "alone us", "loaded us", "ratio", "share"))
    for k in range(nseq):
        print("%-4d %-34s %10.1f %12.1f %8.1f %7.1f%%" % (k + 1, proxy.labels.get(k + 1, "?"), da[k], dl[k],
                                                           dl[k] / max(da[k], 1e-3), 100.0 * dl[k] / max(tl, 1e-9)))


if __name__ == "__main__":
    main()
