"""Throughput benchmark of the SA hot path: point-cloud frames/s through the full 3DSSD SA backbone
(configs/kitti/3dssd/3dssd.yaml rows 1-6), KITTI-shape synthetic frames (16384 x 4), batch 8 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one batch of 8 frames per GPU through the backbone, device-resident in and out.  Steps are
issued round-robin on --streams HIP streams (frames in flight: FPS is a serial chain that keeps only one
CU per frame busy, so throughput comes from overlapping the batches' chains); the timed region is
bracketed by barrier + synchronize on both sides and all K steps complete inside it.  Rank 0 prints ONE
JSON line; `roofline` describes the kernel with the largest share of GPU time, `stages` every kernel,
`cpu_baseline` the CPU oracle timed on this host on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The steps are issued on many HIP streams; the ROCm default of 4 hardware queues would serialise them
# (measured: 16 queues + 16 streams = 1.6x the throughput of the default).  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, MI355X_MICROARCH.md
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA
VALU_F32_PEAK_TF = 157.3
CUS_USED_FPS = 8            # FPS runs one workgroup per frame: batch 8 -> 8 of 256 CUs


def pkg(name):
    return importlib.import_module("3dssd_amd." + name)


# ------------------------------------------------------------------------------------------------
# per-kernel timing: a proxy around the ctypes library records a pair of events around every C-ABI
# call (all kernels are launched on torch's current stream, so torch events bracket them exactly).
class TimingProxy:
    def __init__(self, real):
        self._real = real
        self.records = []   # (name, args, start_event, end_event)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("sa_") or name.endswith("_ws_bytes"):
            return fn

        def wrapped(*args):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            st = fn(*args)
            e.record()
            self.records.append((name, args, s, e))
            return st
        return wrapped


def _algorithmic(name, a):
    """(flops, bytes, label) of one C-ABI call from its scalar arguments (SURVEY.md 8d conventions:
    inputs read once, outputs written once)."""
    if name == "sa_fps_ex":
        b, n, c, m = a[0:4]
        return (3 * c + 2) * b * (m - 1) * n, b * (n * c * 4 + m * 4), "fps n=%d->%d c=%d" % (n, m, c)
    if name == "sa_fps_with_distance_ex":
        b, n, m = a[0:3]
        return 2 * b * (m - 1) * n, b * ((m - 1) * n * 4 + m * 4), "fps_with_distance n=%d->%d" % (n, m)
    if name in ("sa_calc_square_dist_split", "sa_calc_square_dist_split_ws"):
        b, n, m, c0, c1 = a[0:5]
        return 2 * b * n * m * (c0 + c1), b * (n * m * 4 + (n + m) * (c0 + c1) * 4), "calc_square_dist n=%d c=%d" % (n, c0 + c1)
    if name in ("sa_query_ball_point_multi", "sa_query_ball_point_grid"):
        b, n, m, nb = a[0:4]
        ns = [a[6][i] for i in range(nb)]
        return 8 * b * n * m, b * (n * 12 + m * 12 + sum(m * s * 4 + m * 4 for s in ns)), "ball_query%s n=%d m=%d bands=%d" % ("_grid" if name.endswith("grid") else "", n, m, nb)
    if name == "sa_group_mlp_max":
        b, n, m, ns, c = a[0:5]
        nl = a[10]
        dims = [a[11][i] for i in range(nl + 1)]
        macs = sum(dims[i] * dims[i + 1] for i in range(nl))
        by = b * (n * (c + 3) * 4 + m * 12 + m * ns * 4 + m * 4 + m * dims[-1] * 4)
        return 2 * b * m * ns * macs, by, "group_mlp_max m=%d ns=%d %s" % (m, ns, "-".join(map(str, dims)))
    if name == "sa_dense":
        rows, K, N = a[0:3]
        return 2 * rows * K * N, rows * (K + N) * 4 + K * N * 4, "dense %dx%d->%d" % (rows, K, N)
    if name == "sa_gather_point":
        b, n, m, c = a[0:4]
        return 0, b * m * (2 * c * 4 + 4), "gather_point m=%d c=%d" % (m, c)
    if name == "sa_vote_translate":
        return 0, a[0] * 36, "vote_translate"
    return 0, 0, name


def profile_stages(net, pts, iters):
    """Average duration of every kernel of one backbone step, measured live with events."""
    native = pkg("utils._native")
    real = native.lib()
    proxy = TimingProxy(real)
    native._LIB = proxy
    try:
        for _ in range(iters):
            net(pts)
        torch.cuda.synchronize()
    finally:
        native._LIB = real
    agg = {}
    order = []
    for name, args, s, e in proxy.records:
        fl, by, label = _algorithmic(name, args)
        key = (name, label)
        if key not in agg:
            agg[key] = dict(kernel=name, label=label, calls=0, ms=0.0, flops=fl, bytes=by)
            order.append(key)
        agg[key]["calls"] += 1
        agg[key]["ms"] += s.elapsed_time(e)
    stages = []
    for key in order:
        d = agg[key]
        per_step_calls = d["calls"] // iters
        ms = d["ms"] / d["calls"]
        st = dict(kernel=d["kernel"], label=d["label"], calls_per_step=per_step_calls, avg_ms=round(ms, 5),
                  gflop=round(d["flops"] / 1e9, 4), mbytes=round(d["bytes"] / 1e6, 4))
        if ms > 0:
            st["tflops"] = round(d["flops"] / ms / 1e9, 3)
            st["gbs"] = round(d["bytes"] / ms / 1e6, 2)
        stages.append(st)
    return stages


def _pmc_traffic(stage):
    """HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC pass
    (profiles/r01_traffic.json, made by tools/gpu_prof.sh + tools/summarize_prof.py: FETCH_SIZE doubled per the
    gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE).  Only for kernels whose template instance is unique to
    the stage; None otherwise or when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    label = stage["label"]
    name = None
    if stage["kernel"] == "sa_fps_ex" and " c=3" in label:
        n = int(label.split("n=")[1].split("->")[0])
        ppt = 1
        while ppt * 1024 < n:
            ppt *= 2
        name = "fps3_reg_kernel<%d>" % ppt
        if 8192 <= n <= 16384 and int(os.environ.get("SA_FPS_BUCKET_MIN_N", "8192")) > 0:
            name = "fps3_wave_bucket_kernel"          # the culled kernel takes the layer-1 shape (fps.hip dispatch)
    elif stage["kernel"] == "sa_fps_with_distance_ex":
        n = int(label.split("n=")[1].split("->")[0])
        ppt = 1
        while ppt * 1024 < n:
            ppt *= 2
        name = "fpsdist_reg_kernel<%d>" % ppt
    try:
        d = json.load(open(path))
        return int(d[name]["hbm_bytes_per_launch"]) if name in d else None
    except Exception:
        return None


def _pmc_mlp_util():
    """MFMA utilisation of the grouped-MLP kernels from the committed PMC pass (profiles/r01_traffic.json):
    sum of SQ_VALU_MFMA_BUSY_CYCLES over sum of (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), per launch, over the MLP
    kernels of one step.  None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    busy = cap = 0.0
    per_kernel = {}
    for k, e in d.items():
        if ("mlp_r" in k or "group_mlp" in k) and "mfma_util" in e:
            busy += e["mfma_busy_cycles_per_launch"]
            cap += e["gui_active_cycles_per_launch"] / 8.0 * 1024.0
            per_kernel[k] = e["mfma_util"]
    if cap <= 0:
        return None
    return dict(mfma_util=round(busy / cap, 4), per_kernel=per_kernel,
                source="profiles/r01_traffic.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)")


def roofline_of(stage):
    k = stage["kernel"]
    if k in ("sa_group_mlp_max", "sa_dense", "sa_calc_square_dist_split", "sa_calc_square_dist_split_ws"):
        peak = MFMA_BF16_PEAK_TF if not k.startswith("sa_calc_square_dist") else VALU_F32_PEAK_TF
        a = stage.get("tflops", 0.0)
        return dict(kernel=stage["label"], bound="mfma", achieved=a, peak=peak, unit="TFLOP/s",
                    frac=round(a / peak, 5), traffic=None)
    a = stage.get("gbs", 0.0)
    r = dict(kernel=stage["label"], bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s",
             frac=round(a / HBM_PEAK_GBS, 6), traffic=_pmc_traffic(stage),
             algorithmic_bytes=int(stage["mbytes"] * 1e6))
    if k.startswith("sa_fps"):
        # FPS is a serial dependent chain on ONE CU per frame: neither HBM- nor MFMA-bound;
        # the fp32 VALU rate is the meaningful ceiling, quoted beside the (tiny) algorithmic HBM figure.
        r["note"] = ("latency/VALU-bound serial chain; valu_tflops counts the reference's n*(m-1) pair evaluations "
                     "(the wave-bucket kernel used for n >= 8192 skips ~94% of them, bit-identically)")
        r["valu_tflops"] = stage.get("tflops", 0.0)
        r["valu_frac"] = round(stage.get("tflops", 0.0) / VALU_F32_PEAK_TF, 5)
        # one workgroup (one CU) per frame by construction: fraction of the fp32 VALU peak of the CUs it can use
        r["cus_used"] = CUS_USED_FPS
        r["valu_frac_of_cus_used"] = round(stage.get("tflops", 0.0) / (VALU_F32_PEAK_TF * CUS_USED_FPS / 256.0), 4)
    return r


def cpu_baseline(arch, params, batch, budget_s=20.0):
    """The CPU oracle (a scalar C/OpenMP restatement of the reference kernels; the reference has no CPU
    path of its own) on `batch` frames of the same workload, repeated until ~budget_s."""
    from oracle import sa_oracle as O
    cfgs, syn = pkg("configs"), pkg("synthetic")
    pts = syn.kitti_like_batch(batch)
    O.lib()
    t0 = time.time()
    reps = 0
    while True:
        O.sa_backbone(pts, arch, params, cfgs.KITTI_MAX_TRANSLATE_RANGE)
        reps += 1
        dt = time.time() - t0
        if dt > budget_s * 0.6 or reps >= 4:
            break
    return dict(value=round(reps * batch / dt, 4), unit="frames/s", cores=os.cpu_count() or 1, kind="port",
                sample="%d x %d frames of the same 16384-pt workload through oracle.sa_backbone "
                       "(OpenMP over frames/queries), %.1f s" % (reps, batch, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step (BASELINE.json configs[1])")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--streams", type=int, default=16, help="HIP streams the steps are issued on")
    ap.add_argument("--profile-iters", type=int, default=3)
    ap.add_argument("--graphs", type=int, default=1, help="1 (default): capture one hipGraph per stream and replay it; 0: eager launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    sh = pkg("sharding")
    rank, local_rank, world = sh.init()
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pkg("utils._native").lib()
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    params = syn.random_backbone_params(arch)
    net = pkg("backbone").SABackbone(arch, params, dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)

    # this rank's frames: global frame f -> rank f mod world (weak scaling: `batch` frames per GPU)
    frames = sh.frames_of_rank(0, args.batch * world, rank, world)
    pts = torch.from_numpy(np.stack([syn.kitti_like_frame(f, args.points) for f in frames])).to(dev)

    global CUS_USED_FPS
    CUS_USED_FPS = len(frames)          # one FPS workgroup (one CU) per frame
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    graphs = None
    if args.graphs:
        # one captured hipGraph per stream (launch-bound host loop -> one replay per step); every graph
        # owns its intermediate and output buffers, the input frames are static
        for _ in range(2):
            net(pts)
        torch.cuda.synchronize()
        graphs = []
        for st in streams:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                xl, fl, _ = net(pts)
            graphs.append((g, xl[-1], fl[-1]))
        torch.cuda.synchronize()

    def run(k):
        outs = []
        for i in range(k):
            j = i % len(streams)
            with torch.cuda.stream(streams[j]):
                if graphs is not None:
                    graphs[j][0].replay()
                    outs.append((graphs[j][1], graphs[j][2]))
                else:
                    xl, fl, _ = net(pts)
                    outs.append((xl[-1], fl[-1]))
        return outs

    run(args.warmup)
    torch.cuda.synchronize()
    sh.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run(args.steps)
    host_issue_ms = (time.perf_counter() - t0) / args.steps * 1e3
    torch.cuda.synchronize()
    sh.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t_max, frames_total = sh.reduce_timing(elapsed, args.steps * len(frames), device=dev)
    assert outs[-1][1].shape == (len(frames), 256, 512)

    # single-stream latency of one batch (no overlap), for the record
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        net(pts)
    torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) / 3 * 1e3

    if rank == 0:
        stages = profile_stages(net, pts, args.profile_iters)
        tot = {}
        for s in stages:
            tot[s["label"]] = s["avg_ms"] * s["calls_per_step"]
        dom = max(stages, key=lambda s: s["avg_ms"] * s["calls_per_step"])
        mlp = [s for s in stages if s["kernel"] == "sa_group_mlp_max"]
        mlp_ms = sum(s["avg_ms"] * s["calls_per_step"] for s in mlp)
        mlp_fl = sum(s["gflop"] * s["calls_per_step"] for s in mlp)
        bq = [s for s in stages if s["kernel"] in ("sa_query_ball_point_multi", "sa_query_ball_point_grid")]
        bq_ms = sum(s["avg_ms"] * s["calls_per_step"] for s in bq)
        bq_mb = sum(s["mbytes"] * s["calls_per_step"] for s in bq)
        line = {
            "metric": "point-cloud frames/sec through full SA backbone, KITTI 16384-pt",
            "value": round(frames_total / t_max, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 (split hi/lo, 3 MFMA passes, fp32 accumulate) for the grouped MLP; fp32 for FPS / ball query",
            "data": "synthetic KITTI-shape frames (seeded), random-init weights",
            "config": {"workload": "configs[1]: full 3DSSD SA backbone (3dssd.yaml rows 1-6), %d-pt frames, batch=%d per GPU"
                                   % (args.points, args.batch),
                       "frames_per_step_per_gpu": len(frames), "streams": len(streams),
                       "sharding": "frame f -> rank f mod N, no data-path collective"},
            "single_stream_batch_latency_ms": round(latency_ms, 3),
            "host_issue_ms_per_step": round(host_issue_ms, 3),
            "hip_graphs": bool(args.graphs),
            "roofline": roofline_of(dom),
            "roofline_grouped_mlp": {"bound": "mfma", "achieved": round(mlp_fl / mlp_ms, 3) if mlp_ms else 0.0,
                                     "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                                     "frac": round(mlp_fl / mlp_ms / MFMA_BF16_PEAK_TF, 5) if mlp_ms else 0.0,
                                     "note": "algorithmic fp32-equivalent flops; the split-bf16 form issues 3x as many MFMA flops",
                                     "pmc": _pmc_mlp_util()},
            "roofline_ball_query": {"bound": "hbm", "achieved": round(bq_mb / bq_ms, 2) if bq_ms else 0.0,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(bq_mb / bq_ms / HBM_PEAK_GBS, 6) if bq_ms else 0.0},
            "stages": stages,
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(arch, params, min(args.batch, 8))
        print(json.dumps(line))
    sh.barrier()


if __name__ == "__main__":
    main()
