"""Throughput benchmark of the SA hot path: point-cloud frames/s through the full 3DSSD SA backbone
(configs/kitti/3dssd/3dssd.yaml rows 1-6), KITTI-shape synthetic frames (16384 x 4), batch 8 per GPU.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without torchrun: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --data default|dup10|dense|rings64       (sensitivity of the data-dependent kernels)
    python bench.py --workload configs2|configs4|group|detector   (the other single-GPU configurations of BASELINE.json; points -> boxes)
    python bench.py --executor staged|slots                  (3dssd_amd/pipeline.py: default staged)

A "step" is one batch of 8 DIFFERENT frames per GPU through the backbone, device-resident in and out: step i takes
frames 8i .. 8i+7 (mod --pool, default 256 distinct frames per GPU) from a resident pool.  The executor is the
package's own (3dssd_amd/pipeline.py, SAPipeline): a step = one block copy of the batch into a static input buffer;
16 consecutive batches form a package that runs as two captured hipGraphs on three HIP streams (layer-1 sampling stage
on a sampler stream, the rest on one of two main streams) -- all inside the timed region.  (The layer-1 D-FPS is a
serial chain that keeps one CU per frame busy: throughput comes from the sampling stage of one package running beside
the chip-filling kernels of others.)  Order of a run: priming (every slot replayed twice, >= 150 ms of work, independent
of --warmup) -> W warm-up steps -> dress rehearsals (the timed sequence itself, untimed, >= 3 times and on until the last
three agree within 6 %: first-use costs and a host / device disturbance land there, not in the measurement) -> barrier +
synchronize -> EXACTLY K steps -> synchronize + barrier; all K steps complete inside the bracket.  The host sleeps through
most of a window (blocking HIP events in front of the contract's torch.cuda.synchronize()) and the cyclic GC is frozen: a
process under a CPU quota that spins is frozen by the kernel scheduler for the rest of a 100 ms period (measured,
DESIGN.md section 5).  After the bracket, --verify batches go through the same pipeline again and every output is
compared bit for bit with the eager single-stream result of the same batch.  Rank 0 prints ONE JSON line: `config`
starts with flat scalars that say what decided a short run (timed / probe / rehearsal windows, per-call host stamps,
page faults, context switches, cgroup quota and throttle counts, package timeline),
`roofline` describes the kernel with the largest share of GPU time, `stages` every C-ABI call, `cpu_baseline` the CPU
oracle timed on this host on a bounded sample of the same workload.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# (--executor slots asks for one hardware queue per slot -- pipeline.request_hw_queues, before the first CUDA call in
# main(); the default staged executor runs on ROCm's default of 4 queues and sets nothing)
# the host driver of these boxes supports dmabuf IPC only: without this RCCL's peer set-up (N > 1) fails with
# "hipIpcGetMemHandle: invalid argument".  Already exported on the boxes; kept for a shell that lost it.  Not a kernel knob.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, MI355X_MICROARCH.md
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 / fp16 MFMA
VALU_F32_PEAK_TF = 157.3
MAX_CLOCK_MHZ = 2400.0      # MI355X_MICROARCH.md "Max clock"; cycles_per_pick is quoted at this clock (DVFS runs lower)
FPS_FLOP_PER_PAIR = 11      # 3 sub + 3 mul/fma + min + compare/select chain, SURVEY.md 8d (3c + 2 with c = 3)
TRAFFIC_PROFILES = [os.path.join("profiles", "r06_traffic.json"), os.path.join("profiles", "r05_traffic.json"), os.path.join("profiles", "r04_traffic.json")]


def pkg(name):
    return importlib.import_module("3dssd_amd." + name)


# ------------------------------------------------------------------------------------------------ environment
def env_knobs():
    """Every environment variable that can change what is measured.  The package itself reads none (kernel-selection
    knobs exist only in the `make TUNE=1` build, SA3D_LIB selects such a build); they are recorded in the line, and a
    run with any SA_* / SA3D_* variable set is refused unless --allow-knobs."""
    keys = sorted(k for k in os.environ if k.startswith(("SA_", "SA3D_")))
    rec = {k: os.environ[k] for k in keys}
    for k in ("GPU_MAX_HW_QUEUES", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "HSA_ENABLE_IPC_MODE_LEGACY",
              "HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL"):
        if k in os.environ:
            rec[k] = os.environ[k]
    return rec, keys


# ------------------------------------------------------------------------------------------------
# per-kernel timing: a proxy around the ctypes library records a pair of events around every C-ABI
# call (all kernels are launched on torch's current stream, so torch events bracket them exactly).
class TimingProxy:
    def __init__(self, real):
        self._real = real
        self.records = []   # (name, args, start_event, end_event)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("sa_") or name.endswith(("_ws_bytes", "_rows", "_state")):       # host-side queries: no launch to time
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
    if name in ("sa_fps_ex", "sa_fps_ex2", "sa_fps_ex3", "sa_farthest_point_sample"):
        b, n, c, m = a[0:4]
        return (3 * c + 2) * b * (m - 1) * n, b * (n * c * 4 + m * 4), "fps n=%d->%d c=%d" % (n, m, c)
    if name in ("sa_fps_with_distance_ex", "sa_fps_with_distance_ex2"):
        b, n, m = a[0:3]
        return 2 * b * (m - 1) * n, b * ((m - 1) * n * 4 + m * 4), "fps_with_distance n=%d->%d" % (n, m)
    if name == "sa_fps_dual_ex":                     # matrix sampler || coordinate sampler, one launch
        b, nf, mf, nd, md = a[0], a[1], a[2], a[11], a[12]
        return (2 * b * (mf - 1) * nf + 11 * b * (md - 1) * nd,
                b * ((mf - 1) * nf * 4 + mf * 4 + nd * 12 + md * 4), "fps_dual F n=%d->%d | D n=%d->%d" % (nf, mf, nd, md))
    if name == "sa_calc_square_dist_self_ws":
        b, n, c0, c1 = a[0:4]
        return 2 * b * n * n * (c0 + c1), b * (n * n * 4 + 2 * n * (c0 + c1) * 4), "calc_square_dist n=%d c=%d" % (n, c0 + c1)
    if name in ("sa_calc_square_dist_split", "sa_calc_square_dist_split_ws"):
        b, n, m, c0, c1 = a[0:5]
        return 2 * b * n * m * (c0 + c1), b * (n * m * 4 + (n + m) * (c0 + c1) * 4), "calc_square_dist n=%d c=%d" % (n, c0 + c1)
    if name in ("sa_query_ball_point_multi", "sa_query_ball_point_grid", "sa_query_ball_point_grid_ex"):
        b, n, m, nb = a[0:4]
        ns = [a[6][i] for i in range(nb)]
        return 8 * b * n * m, b * (n * 12 + m * 12 + sum(m * s * 4 + m * 4 for s in ns)), "ball_query%s n=%d m=%d bands=%d" % ("_grid" if "grid" in name else "", n, m, nb)
    if name == "sa_group_mlp_max":
        b, n, m, ns, c = a[0:5]
        nl = a[10]
        dims = [a[11][i] for i in range(nl + 1)]
        macs = sum(dims[i] * dims[i + 1] for i in range(nl))
        by = b * (n * (c + 3) * 4 + m * 12 + m * ns * 4 + m * 4 + m * dims[-1] * 4)
        return 2 * b * m * ns * macs, by, "group_mlp_max m=%d ns=%d %s" % (m, ns, "-".join(map(str, dims)))
    if name == "sa_group_mlp_max_layer":
        k, b, n, m = a[0:4]
        c, nl = a[5], a[11]
        ns = [a[4][i] for i in range(k)]
        fl = by = 0
        shapes = []
        for i in range(k):
            dims = [a[12][i * (nl + 1) + j] for j in range(nl + 1)]
            macs = sum(dims[j] * dims[j + 1] for j in range(nl))
            fl += 2 * b * m * ns[i] * macs
            by += b * (m * ns[i] * 4 + m * 4 + m * dims[-1] * 4)
            shapes.append("%d:%s" % (ns[i], "-".join(map(str, dims))))
        by += b * (n * (c + 3) * 4 + m * 12)
        return fl, by, "group_mlp_max_layer m=%d %s" % (m, " ".join(shapes))
    if name == "sa_dense":
        rows, K, N = a[0:3]
        return 2 * rows * K * N, rows * (K + N) * 4 + K * N * 4, "dense %dx%d->%d" % (rows, K, N)
    if name == "sa_group_point":
        b, n, c, m, ns = a[0:5]
        return 0, b * (n * c * 4 + m * ns * 4 + m * ns * c * 4), "group_point m=%d ns=%d c=%d" % (m, ns, c)
    if name in ("sa_query_ball_point", "sa_query_ball_point_dilated"):
        b, n, m = a[0:3]
        ns = a[4] if name == "sa_query_ball_point" else a[5]
        return 8 * b * n * m, b * (n * 12 + m * 12 + m * ns * 4 + m * 4), "ball_query n=%d m=%d ns=%d" % (n, m, ns)
    if name == "sa_gather_point":
        b, n, m, c = a[0:4]
        return 0, b * m * (2 * c * 4 + 4), "gather_point m=%d c=%d" % (m, c)
    if name == "sa_vote_translate":
        return 0, a[0] * 36, "vote_translate"
    if name == "sa_vote_tail":                       # hidden conv1d + offset conv1d + translation, one launch
        rows, K, H = a[0:3]
        return 2 * rows * (K * H + H * 3), rows * (K + H + 3 + 3 + 3) * 4 + (K * H + H * 3) * 4, "vote_tail %dx%d->%d->3" % (rows, K, H)
    return 0, 0, name


EXECUTOR_NOTES = {
    "staged": "3dssd_amd.pipeline.SAPipeline(mode='staged'): 3 HIP streams on ROCm's default 4 hardware queues (no "
              "environment variable), %(n)d packages of %(C)d batches x %(B)d frames in a ring; a package = two captured, "
              "linear hipGraphs -- stage A (input split + layer-1 D-FPS + centres) on the sampler stream, stage B "
              "(everything else) on one of two main streams behind an event; one block copy per step; a partly filled "
              "package runs the smallest captured size (C, C/2, C/4) that holds it",
    "slots": "3dssd_amd.pipeline.SAPipeline(mode='slots'): %(n)d slots = %(n)d HIP streams, each with one captured hipGraph "
             "of the backbone over %(C)d batches x %(B)d frames; one block copy per step, one replay per %(C)d steps",
}
# ---- the secondary measurements the headline line carries as flat scalars --------------------------------
# key -> (bench.py arguments of the sub-run, what to read from its line).  Each runs in a process of its own right after the
# headline (the hardware-queue count and the executor's streams are per process), bounded by --extras-budget seconds in all.
EXTRA_RUNS = [
    ("steady512_frames_s", ["--steps", "512", "--warmup", "64"], "value"),
    ("rings64_frames_s", ["--data", "rings64", "--steps", "512", "--warmup", "64"], "value"),
    ("detector_frames_s", ["--workload", "detector", "--steps", "512", "--warmup", "64"], "value"),
    ("configs4_frames_s", ["--workload", "configs4"], "value"),
    ("configs2_frames_s", ["--workload", "configs2"], "value"),
    ("group_b128_hbm_frac", ["--workload", "group", "--batch", "128", "--profile-iters", "1"], "roofline.frac"),
    ("group_b32_hbm_frac", ["--workload", "group", "--batch", "32", "--profile-iters", "1"], "roofline.frac"),
    ("group_b8_hbm_frac", ["--workload", "group", "--batch", "8", "--profile-iters", "1"], "roofline.frac"),
    ("dense_frames_s", ["--data", "dense", "--steps", "256", "--warmup", "32"], "value"),
]
EXTRA_KEYS = [k for k, _a, _w in EXTRA_RUNS] + ["rccl_smoke"]


def run_extras(args, line):
    """Fill line["config"][EXTRA_KEYS] (already present, None) from sub-runs of this script; never raises."""
    cfg = line["config"]
    t_start = time.perf_counter()
    budget = float(args.extras_budget)
    detail, skipped = {}, []
    base = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-executor", "--extras-budget", "0", "--verify", "0"]
    if args.allow_knobs:
        base.append("--allow-knobs")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES" or args.hwq_from_env}
    for key, extra, what in EXTRA_RUNS:
        if key == "steady512_frames_s" and args.steps >= 512 and args.data == "default":
            cfg[key] = line["value"]
            continue
        left = budget - (time.perf_counter() - t_start)
        if left < 8.0:
            skipped.append(key)
            continue
        cmd = base + (extra if "--profile-iters" in extra else extra + ["--profile-iters", "0"])
        t0 = time.perf_counter()
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=min(left, 90.0), env=env).stdout.strip().splitlines()
            d = json.loads(out[-1])
            v = d
            for part in what.split("."):
                v = v[part]
            cfg[key] = v
            detail[key] = {"seconds": round(time.perf_counter() - t0, 1), "steps": d.get("steps"), "ms_per_step": d.get("ms_per_step"),
                           "verify": (d.get("verify") or {}).get("all_equal_eager")}
        except Exception as e:  # noqa: BLE001 -- a secondary figure must not take the headline down
            detail[key] = {"error": repr(e)[:200], "seconds": round(time.perf_counter() - t0, 1)}
    left = budget - (time.perf_counter() - t_start)
    if left >= 8.0:
        r = pkg("sharding").rccl_smoke_subprocess(timeout_s=int(min(left, 120.0)))
        cfg["rccl_smoke"] = r.get("status")
        detail["rccl_smoke"] = r
    else:
        skipped.append("rccl_smoke")
    line["extras"] = {"seconds": round(time.perf_counter() - t_start, 1), "budget_s": budget, "skipped": skipped, "runs": detail,
                      "note": "sub-runs of this script in processes of their own right after the headline measurement (never inside "
                              "its timed region): 512-step steady state, --data rings64 / dense, points -> boxes (--workload "
                              "detector), configs[4] / configs[2], the unfused ball_query + group workload at 128 / 32 / 8 frames "
                              "per call (roofline.frac = SURVEY 8d bytes / kernel time / 8 TB/s), and a world-size-1 RCCL smoke"}


MLP_CALLS = ("sa_group_mlp_max", "sa_group_mlp_max_layer")
MFMA_CALLS = MLP_CALLS + ("sa_dense", "sa_vote_tail")


def profile_stages(fn, iters):
    """Average duration of every C-ABI call of one step (`fn()`), measured live with events on the launch stream.
    Eager launches on ONE stream: these are kernel durations, not the overlapped multi-stream step time.  Flop rates:
    `tflops_nominal` divides the reference's m x nsample rows by the time; `tflops_executed` (grouped-MLP
    calls) the rows the kernels really evaluate (plan headers: granules x rows per granule)."""
    if iters <= 0:
        return []
    native = pkg("utils._native")
    lu = pkg("utils.layers_util")
    real = native.lib()
    proxy = TimingProxy(real)
    native._LIB = proxy
    lu.PLAN_LOG = []
    try:
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    finally:
        native._LIB = real
        plan_log, lu.PLAN_LOG = lu.PLAN_LOG, None
    executed = {}                                   # m of the layer -> executed flops per call (mean over the iterations)
    for (b, m, ns, macs, plan) in plan_log:
        h = plan[:4].cpu().tolist()
        executed[m] = executed.get(m, 0.0) + 2.0 * h[0] * (h[3] or 8) * macs / iters
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
            st["tflops_nominal"] = round(d["flops"] / ms / 1e9, 3)
            st["gbs"] = round(d["bytes"] / ms / 1e6, 2)
            if d["kernel"] in MLP_CALLS and " m=" in d["label"]:
                m = int(d["label"].split(" m=")[1].split()[0])
                if m in executed:
                    st["gflop_executed"] = round(executed[m] / per_step_calls / 1e9, 4)
                    st["tflops_executed"] = round(executed[m] / per_step_calls / ms / 1e9, 3)
        stages.append(st)
    return stages


def _profile_json():
    for rel in TRAFFIC_PROFILES:
        try:
            return json.load(open(os.path.join(ROOT, rel))), rel
        except Exception:
            continue
    return None, None


def _fps_kernel_names(stage):
    """rocprofv3 names of the kernel a sampler call dispatches to (csrc/fps.hip sa_fps_ex2 / fps_coop.hip), newest
    spelling first."""
    label = stage["label"]
    n = int(label.split("n=")[1].split("->")[0])
    ppt = 1
    while ppt * 1024 < n:
        ppt *= 2
    if stage["kernel"] in ("sa_fps_with_distance_ex", "sa_fps_with_distance_ex2"):
        return ["fpsdist_reg_kernel<%d>" % ppt]
    if " c=3" in label and 8192 <= n <= 16384:
        return ["fps3_wave_bucket_kernel<false>", "fps3_wave_bucket_kernel"]
    if " c=3" in label and n <= 16384:
        return ["fps3_reg_kernel<%d>" % ppt]
    c = int(label.split("c=")[1])
    return ["fps_coop_kernel<%d, %d>" % (c, 4 if c == 3 else 1)]


def _pmc_traffic(stage):
    """HBM bytes per launch of the stage's kernel from the COMMITTED rocprofv3 PMC pass (profiles/, made by
    tools/gpu_prof128.sh + tools/summarize_128f.py: FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md +
    WRITE_SIZE).  A snapshot keyed by kernel name, NOT collected in this run; (None, None) when absent."""
    d, rel = _profile_json()
    if d is None or not stage["kernel"].startswith("sa_f"):
        return None, None
    for name in _fps_kernel_names(stage):
        if name in d and "hbm_bytes_per_launch" in d[name]:
            return int(d[name]["hbm_bytes_per_launch"]), rel
    return None, None


def _pmc_mlp_util():
    """MFMA utilisation of the grouped-MLP kernels from the committed PMC pass: sum of SQ_VALU_MFMA_BUSY_CYCLES over
    sum of (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), per launch, over the MLP kernels of one step."""
    d, rel = _profile_json()
    if d is None:
        return None
    busy = cap = 0.0
    per_kernel = {}
    for k, e in d.items():
        if ("mlp_r" in k or "mlp_multi" in k or "group_mlp" in k or "mlp_gemm" in k) and "mfma_util" in e:
            busy += e["mfma_busy_cycles_per_launch"]
            cap += e["gui_active_cycles_per_launch"] / 8.0 * 1024.0
            per_kernel[k] = e["mfma_util"]
    if cap <= 0:
        return None
    return dict(mfma_util=round(busy / cap, 4), per_kernel=per_kernel,
                source="committed profile %s (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE), not this run" % rel)


def fps_bucket_evaluated(pts_xyz, m):
    """Pair evaluations the culled layer-1 sampler really performs on these frames (sa_fps_bucket_stats: bucket
    re-evaluations x 256 point slots), next to the reference's (m - 1) * n."""
    lib = pkg("utils._native").lib()
    b, n, _ = pts_xyz.shape
    if not (8192 <= n <= 16384):
        return None
    out = torch.empty((b, m), dtype=torch.int32, device=pts_xyz.device)
    stats = torch.zeros((b, 2), dtype=torch.int64, device=pts_xyz.device)
    st = lib.sa_fps_bucket_stats(b, n, m, pts_xyz.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    if st != 0:
        return None
    torch.cuda.synchronize()
    return int(stats[:, 0].sum().item()) * 256


def roofline_of(stage, frames, clock_mhz, evaluated_pairs=None):
    k = stage["kernel"]
    if k in MFMA_CALLS or k.startswith("sa_calc_square_dist"):
        peak = MFMA_BF16_PEAK_TF if not k.startswith("sa_calc_square_dist") else VALU_F32_PEAK_TF
        a = stage.get("tflops_executed", stage.get("tflops_nominal", 0.0))
        return dict(kernel=stage["label"], bound="mfma", achieved=a, peak=peak, unit="TFLOP/s",
                    frac=round(a / peak, 5), traffic=None)
    a = stage.get("gbs", 0.0)
    tr, rel = _pmc_traffic(stage)
    if not (k.startswith("sa_fps") or k == "sa_farthest_point_sample"):
        return dict(kernel=stage["label"], bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(a / HBM_PEAK_GBS, 6), traffic=tr,
                    traffic_source=("committed profile %s, not collected in this run" % rel) if tr is not None else None,
                    algorithmic_bytes=int(stage["mbytes"] * 1e6), avg_launch_ms=stage.get("avg_ms"))
    # FPS: a serial dependent chain -- neither HBM- nor MFMA-bound.  Reported as the reference's pair
    # evaluations x 11 flop / time against the fp32 VALU peak ("VALU-equivalent"), with what is really evaluated, the
    # latency per pick and the (tiny, by nature) HBM figures beside it.
    label = stage["label"]
    n = int(label.split("n=")[1].split("->")[0])
    m = int(label.split("->")[1].split()[0])
    c = int(label.split("c=")[1]) if "c=" in label else 3
    ms = stage.get("avg_ms", 0.0)
    pairs = frames * (m - 1) * n
    flop_pair = 3 * c + 2
    tf = pairs * flop_pair / ms / 1e9 if ms > 0 else 0.0
    names = _fps_kernel_names(stage)
    coop = names[0].startswith("fps_coop")
    if coop:
        g = 1
        while 1024 * (4 if c == 3 else 1) * g < n:
            g *= 2
        wgs, note = frames * g, ("latency-bound serial chain: %d cooperating workgroups of 1024 threads per frame "
                                 "(csrc/fps_coop.hip), one cross-workgroup arg-max exchange per pick; a cooperative launch "
                                 "holds 256 workgroups, more frames run as consecutive launches inside the call" % g)
        launches = -(-wgs // 256)
    else:
        wgs, note = frames, ("latency-bound serial chain: one workgroup (= one CU) per frame, m-1 dependent picks; "
                             "bucket culling evaluates only the pairs near each pick")
        launches = 1
    r = dict(kernel=label, device_kernel=names[0], bound="latency", achieved=round(tf, 4), peak=VALU_F32_PEAK_TF,
             unit="TFLOP/s", frac=round(tf / VALU_F32_PEAK_TF, 5),
             basis="reference pair evaluations x %d flop / kernel time, against the fp32 VALU peak" % flop_pair,
             traffic=tr, traffic_source=("committed profile %s, not collected in this run" % rel) if tr is not None else None,
             algorithmic_bytes=int(stage["mbytes"] * 1e6), hbm_gbs=a, hbm_frac=round(a / HBM_PEAK_GBS, 7),
             avg_launch_ms=round(ms / launches, 5), sequential_launches_per_call=launches,
             us_per_pick=round(ms * 1e3 / launches / max(m - 1, 1), 4),
             cycles_per_pick=round(ms * 1e3 / launches / max(m - 1, 1) * clock_mhz, 1), clock_mhz=clock_mhz,
             workgroups=wgs, cus_used=min(wgs, 256), reference_pair_evaluations=pairs, note=note)
    if evaluated_pairs is not None:
        r["evaluated_pairs"] = int(evaluated_pairs)
        r["evaluated_frac"] = round(evaluated_pairs / max(pairs, 1), 5)
        r["evaluated_tflops"] = round(evaluated_pairs * flop_pair / ms / 1e9, 4) if ms > 0 else 0.0
        r["evaluated_frac_of_peak_on_cus_used"] = round(r["evaluated_tflops"] / (VALU_F32_PEAK_TF * min(wgs, 256) / 256.0), 5)
    return r


def cpu_baseline(arch, params, batch, points, variant, budget_s=20.0):
    """The CPU oracle (a scalar C/OpenMP restatement of the reference kernels; the reference has no CPU
    path of its own) on `batch` frames of the same workload, repeated until ~budget_s."""
    from oracle import sa_oracle as O
    cfgs, syn = pkg("configs"), pkg("synthetic")
    pts = np.stack([syn.frame_of(variant, f, points) for f in range(batch)])
    O.lib()
    t0 = time.time()
    reps = 0
    while True:
        O.sa_backbone(pts, arch, params, cfgs.KITTI_MAX_TRANSLATE_RANGE)
        reps += 1
        dt = time.time() - t0
        if dt > budget_s * 0.6 or reps >= 4:
            break
    cores = os.cpu_count() or 1
    return dict(value=round(reps * batch / dt, 4), unit="frames/s", cores=cores, kind="port",
                effective_parallelism="OpenMP over frames in the FPS stages (<= %d of the %d threads busy there: the "
                                      "4 095-pick chain of a frame is serial on the CPU too), over queries / rows in "
                                      "the ball-query and MLP stages" % (batch, cores),
                sample="%d x %d frames of the same %d-pt workload (--data %s) through oracle.sa_backbone, %.1f s"
                       % (reps, batch, points, variant, dt))


# ------------------------------------------------------------------------------------------------ launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(args):
    """`python bench.py --gpus N` outside torchrun: re-exec under torch.distributed.run with one rank per GPU
    (the N in-graph towers of lib/core/trainer.py:120-155 become N processes)."""
    if not args.launch_check:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and not args.allow_shared_device:
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible on this node -- refusing to run fewer ranks than "
                     "requested (--allow-shared-device runs the ranks on the GPUs there are, for a functional check of "
                     "the multi-rank path; not a scaling point)" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def pin_rank(args, local_rank, world):
    """--cpu-affinity: the issuing thread of this rank on a core of its own ('auto': the allowed cores divided evenly
    among the ranks of the node; or a core number).  Returns the core, or None when the flag is absent."""
    if args.cpu_affinity is None:
        return None
    allowed = sorted(os.sched_getaffinity(0))
    core = allowed[(local_rank * max(1, len(allowed) // max(world, 1))) % len(allowed)] if args.cpu_affinity == "auto" \
        else int(args.cpu_affinity)
    os.sched_setaffinity(0, {core})
    return core


def launch_check(args, sh):
    """Launch path only (CPU-capable, gloo when no GPU): spawn, rendezvous, world == --gpus, the frame partition, one
    reduction, the result gather."""
    import torch.distributed as dist
    rank, local_rank, world = sh.init(backend=None if torch.cuda.is_available() else "gloo")
    assert world == args.gpus, "launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    core = pin_rank(args, local_rank, world)
    frames = sh.frames_of_rank(0, args.batch * world, rank, world)
    sh.barrier()
    t_max, total = sh.reduce_timing(1.0 + rank, len(frames))
    # the partition: every global frame owned by exactly one rank (all-gather of the per-rank frame lists)
    per_rank, cores = [len(frames)], [core]
    partition_ok = True
    if dist.is_initialized():
        lists = [None] * world
        dist.all_gather_object(lists, (frames, core))
        per_rank, cores = [len(f) for f, _ in lists], [c for _, c in lists]
        partition_ok = sorted(f for fr, _ in lists for f in fr) == list(range(args.batch * world))
    # the result gather of configs[3] on stand-in outputs of the real shapes, every rank's different (rank order is checked
    # by digest): the same call the timed workloads make after their timed region
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1)) if torch.cuda.is_available() else torch.device("cpu")
    gathered = sh.gather_check(torch.full((args.batch, 256, 3), float(rank), device=dev),
                               torch.full((args.batch, 256, 512), 0.5 + rank, device=dev))
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "frames_total": total, "t_max": t_max,
                          "frames_per_rank": per_rank, "partition_ok": bool(partition_ok), "cores": cores, "gather": gathered}))
    sh.barrier()


# ------------------------------------------------------------------------------------------------ workloads
def sclk_mhz(dev=None):
    """Current shader clock (pp_dpm_sclk's starred level, MHz).  With `dev`: of the card whose PCI address is that
    device's (an int, or None when it cannot be matched); without: of every card the kernel driver exposes (a list)."""
    import glob

    def star(f):
        try:
            for ln in open(f):
                if "*" in ln:
                    return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        except Exception:  # noqa: BLE001
            pass
        return None
    files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if dev is None:
        return [v for v in (star(f) for f in files) if v is not None] or None
    try:
        pr = torch.cuda.get_device_properties(dev)
        want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        for f in files:
            if want in os.path.realpath(os.path.dirname(f)):
                return star(f)
    except Exception:  # noqa: BLE001
        pass
    return None


def _cgroup_dirs():
    """Directories of this process's cgroup (v2) from the leaf up to the mount point, those that exist."""
    rel = ""
    try:
        for ln in open("/proc/self/cgroup"):
            parts = ln.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2].strip("/")
    except Exception:  # noqa: BLE001
        pass
    out, cur = [], rel
    while True:
        d = os.path.join("/sys/fs/cgroup", cur) if cur else "/sys/fs/cgroup"
        if os.path.isdir(d):
            out.append(d)
        if not cur:
            break
        cur = os.path.dirname(cur)
    return out


def _cgroup_cpu_stat():
    """(nr_throttled, throttled_usec) of this process's cgroup: the deepest level that has the counters (v2 cpu.stat, or
    the v1 files), or None."""
    files = [os.path.join(d, "cpu.stat") for d in _cgroup_dirs()] + ["/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"]
    for f in files:
        try:
            d = dict(ln.split() for ln in open(f).read().splitlines() if len(ln.split()) == 2)
            if "nr_throttled" not in d:
                continue
            t = int(d.get("throttled_usec", int(d.get("throttled_time", 0)) // 1000))
            return int(d["nr_throttled"]), t
        except Exception:  # noqa: BLE001
            continue
    return None


def cgroup_cpu_quota():
    """The tightest CPU quota over this process's cgroup levels, in cores (cpu.max = "quota period"), or None (no limit)."""
    best = None
    for d in _cgroup_dirs():
        try:
            q, per = open(os.path.join(d, "cpu.max")).read().split()[:2]
            if q != "max":
                c = int(q) / float(per)
                best = c if best is None else min(best, c)
        except Exception:  # noqa: BLE001
            continue
    return best


def _ioctl_trace():
    """tools/microbench/ioctl_trace.c preloaded (LD_PRELOAD, diagnostics runs only): (reset, read) or None."""
    import ctypes
    try:
        h = ctypes.CDLL(None)
        rd, rs = h.ioctl_trace_read, h.ioctl_trace_reset
    except (AttributeError, OSError):
        return None
    buf = (ctypes.c_uint64 * 4)()
    txt = ctypes.create_string_buffer(1024)

    def read():
        rd(buf)
        out = {"calls": int(buf[0]), "total_us": round(buf[1] / 1e3, 1), "max_us": round(buf[2] / 1e3, 1), "max_request": hex(int(buf[3]))}
        try:
            h.ioctl_trace_top(txt, 1024)
            out["top"] = txt.value.decode(errors="replace")
        except AttributeError:
            pass
        return out
    return rs, read


class HostWatch:
    """What the HOST did during a timed window, from counters that cost nothing inside it: page faults and context
    switches of this process (getrusage), CFS throttling of its cgroup, and the per-call time stamps the executor wrote
    into a pre-allocated list (SAPipeline.host_trace).  start() / stop() sit just outside the timed bracket."""

    def __init__(self, pipe=None):
        import resource
        self._ru = lambda: resource.getrusage(resource.RUSAGE_SELF)
        self.pipe = pipe
        self.ioctl = _ioctl_trace()

    def start(self):
        self.trace = []
        self.cg0 = _cgroup_cpu_stat()
        self.ru0 = self._ru()
        if self.ioctl:
            self.ioctl[0]()
        if self.pipe is not None:
            self.pipe.host_trace = self.trace
        self.t0 = time.perf_counter_ns()

    def issued(self):                       # run(steps) has returned: everything is enqueued
        self.t_issued = time.perf_counter_ns()

    def stop(self):
        self.t1 = time.perf_counter_ns()
        if self.pipe is not None:
            self.pipe.host_trace = None
        self.ru1 = self._ru()
        self.cg1 = _cgroup_cpu_stat()
        self.io = self.ioctl[1]() if self.ioctl else None

    def summary(self):
        """Flat scalars only (the driver record keeps those).  A stamp's delta = the time from the previous stamp to it,
        i.e. the cost of reaching that call site: `host_stall_at` names the site of the largest one."""
        tr = [("start", self.t0)] + self.trace
        deltas = [((tr[i][1] - tr[i - 1][1]) / 1e3, tr[i][0], i) for i in range(1, len(tr))]
        out = {}
        if deltas:
            worst = max(deltas)
            ds = sorted(d for d, _, _ in deltas)
            out.update(host_stall_max_ms=round(worst[0] / 1e3, 4), host_stall_at="%s (stamp %d of %d)" % (worst[1], worst[2], len(deltas)),
                       host_issue_median_us=round(ds[len(ds) // 2], 2), host_issue_p99_us=round(ds[min(len(ds) - 1, int(len(ds) * 0.99))], 2),
                       host_stamps=len(deltas), host_issue_total_ms=round((self.t_issued - self.t0) / 1e6, 4))
            by = {}
            for d, site, _ in deltas:
                by[site] = by.get(site, 0.0) + d
            top = sorted(by.items(), key=lambda kv: -kv[1])[:3]
            out["host_issue_top_sites"] = "; ".join("%s %.0f us" % kv for kv in top)
        out.update(majflt=self.ru1.ru_majflt - self.ru0.ru_majflt, minflt=self.ru1.ru_minflt - self.ru0.ru_minflt,
                   nvcsw=self.ru1.ru_nvcsw - self.ru0.ru_nvcsw, nivcsw=self.ru1.ru_nivcsw - self.ru0.ru_nivcsw,
                   cpu_user_ms=round((self.ru1.ru_utime - self.ru0.ru_utime) * 1e3, 2),
                   cpu_sys_ms=round((self.ru1.ru_stime - self.ru0.ru_stime) * 1e3, 2))
        if self.cg0 is not None and self.cg1 is not None:
            out.update(cgroup_nr_throttled=self.cg1[0] - self.cg0[0], cgroup_throttled_us=self.cg1[1] - self.cg0[1])
        if self.io is not None:
            out.update(ioctl_calls=self.io["calls"], ioctl_total_us=self.io["total_us"], ioctl_max_us=self.io["max_us"],
                       ioctl_max_request=self.io["max_request"], ioctl_top=self.io.get("top"))
        return out


def timed_region(sh, dev, run, steps, warmup, frames_per_step, prime=None, before=None, after=None, rehearse=None, watch=None,
                 idle_wait=None):
    """`prime()` (untimed, independent of --warmup) -> warmup steps -> `rehearse()` (untimed: the timed sequence itself,
    same brackets, repeated until its duration is stable) -> barrier + synchronize -> EXACTLY `steps` steps -> synchronize
    + barrier.  before() / after() run just outside the timed bracket.  The cyclic garbage collector is collected once and
    frozen BEFORE the rehearsals and stays off across the bracket (a generation-2 pass over a process that has imported
    torch takes 35 ms -- longer than a 20-step window).  idle_wait(t0): an optional sleeping wait for the executor's
    streams in front of the bracket's torch.cuda.synchronize() (which then returns at once): the host thread does not
    spin while the GPU works (a process under a CPU quota is frozen for the rest of a 100 ms period once it has used the
    quota up: measured, profiles/r05_stall_quota.txt)."""
    import gc
    primed = prime() if prime is not None else None
    run(warmup)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    gc.freeze()
    try:
        if rehearse is not None:
            rehearse()
        sh.barrier()
        torch.cuda.synchronize()
        if before is not None:
            before()
        if watch is not None:
            watch.start()
        t0 = time.perf_counter()
        outs = run(steps)
        host_issue_ms = (time.perf_counter() - t0) / steps * 1e3
        if watch is not None:
            watch.issued()
        if idle_wait is not None:
            idle_wait(t0)
        torch.cuda.synchronize()
        sh.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if watch is not None:
            watch.stop()
    finally:
        gc.unfreeze()
        gc.enable()
    if after is not None:
        after()
    t_max, frames_total = sh.reduce_timing(elapsed, steps * frames_per_step, device=dev)
    return t_max, frames_total, host_issue_ms, outs, primed


def overlap_probe(pipe, run, k):
    """How the packages of the headline run overlap, measured by the application itself (rocprofv3's kernel trace
    serialises the streams: profiles/r02_trace16_summary.txt shows 1.4 kernels in flight under the profiler).  A
    second, untimed run of k steps with the executor's own HIP timing events around every package (SAPipeline
    timeline): a package's device-side span is several times the time between completions, i.e. that many are in
    flight."""
    torch.cuda.synchronize()
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    torch.cuda.synchronize()
    pipe.timeline(base)
    pipe.record_timeline = True
    run(k)
    pipe.drain()
    pipe.record_timeline = False
    rows = pipe.timeline(base)
    iv = sorted((ms[0], ms[-1], fill) for _slot, fill, ms in rows)
    ends = sorted(e for _, e, _ in iv)
    # steady part: from the first completion to the last start
    t0, t1 = ends[0], max(s for s, _, _ in iv)
    if t1 <= t0:
        t0, t1 = iv[0][0], ends[-1]
    busy = sum(max(0.0, min(e, t1) - max(s, t0)) * f for s, e, f in iv)
    spans = [e - s for s, e, _ in iv]
    done = [e for e in ends if t0 <= e <= t1]
    return {"steps": k, "packages": len(iv), "window_ms": round(ends[-1] - iv[0][0], 3),
            "package_span_ms_mean": round(sum(spans) / len(spans), 3),
            "package_span_ms_min": round(min(spans), 3), "package_span_ms_max": round(max(spans), 3),
            "steps_in_flight_mean": round(busy / (t1 - t0), 2),
            "ms_between_completions": round((t1 - t0) / max(len(done) - 1, 1), 4),
            "note": "device-side spans (reached by its stream -> complete) of the packages of a second, untimed run; steps in "
                    "flight = sum of (span inside the steady window x batches of the package) / its length"}


def mlp_row_stats(net, batches):
    """Rows of the grouped tensor per step: nominal (m x nsample, what the reference's conv2d evaluates), distinct
    (sum of clamp(cnt, 1, ns)) and evaluated (the plan's granules of 8 or 4 rows), from the plan headers of eager steps
    over `batches` (mean per step)."""
    lu = pkg("utils.layers_util")
    nominal = distinct = evaluated = 0
    fl_nom = fl_eval = 0.0
    for pts in batches:
        lu.PLAN_LOG = []
        net(pts)
        torch.cuda.synchronize()
        log, lu.PLAN_LOG = lu.PLAN_LOG, None
        for (b, m, ns, macs, plan) in log:
            h = plan[:4].cpu().tolist()
            nominal += b * m * ns
            distinct += h[2]
            gr = h[3] or 8                                   # rows per granule of this plan (csrc/mlp_plan.h: 8 or 4)
            evaluated += h[0] * gr
            fl_nom += 2.0 * b * m * ns * macs
            fl_eval += 2.0 * h[0] * gr * macs
    k = max(len(batches), 1)
    return dict(nominal=nominal // k, distinct=distinct // k, evaluated=evaluated // k,
                frames=int(batches[0].shape[0]) if len(batches) else 0,
                evaluated_frac=round(evaluated / max(nominal, 1), 4), batches_sampled=len(batches),
                gflop_nominal=round(fl_nom / 1e9 / k, 3), gflop_evaluated=round(fl_eval / 1e9 / k, 3))


def _sha1(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()


def verify_pipeline(pipe, batches, nverify, submit_from=None):
    """`nverify` DISTINCT batches through the pipeline with all slots in flight, each output compared bit for bit with
    the eager single-stream result of the same batch (a replay that read another slot's scratch or input would differ)."""
    nverify = min(nverify, len(batches))
    if nverify <= 0:
        return None
    submit_from = batches if submit_from is None else submit_from      # (the pinned host copies with --host-input)
    eager, eager_tail = [], []
    for i in range(nverify):
        lists, extra = pipe.tail_eager(batches[i])
        eager.append((lists[0][-1].clone(), lists[1][-1].clone()))
        eager_tail.append(None if extra is None else {k: v.clone() for k, v in extra.items()})
    torch.cuda.synchronize()
    equal, first = True, None
    per_round = pipe.nslots * pipe.coalesce                     # every slot full: that many batches in flight
    for r0 in range(0, nverify, per_round):
        tickets = [(i, pipe.submit(submit_from[i], sync_source=False, defer_copy=True)) for i in range(r0, min(r0 + per_round, nverify))]
        pipe.flush()
        for i, t in tickets:
            x, f = t.result()
            ok = torch.equal(x, eager[i][0]) and torch.equal(f, eager[i][1])
            if eager_tail[i] is not None:                        # --workload detector: boxes, scores, classes, NMS indices / counts
                got = t.detections()
                ok = ok and all(torch.equal(got[k], v) for k, v in eager_tail[i].items())
            equal = equal and ok
            if first is None:
                first = (_sha1(f), _sha1(eager[i][1]))
    return {"batches": nverify, "slots_in_flight": min(pipe.nslots, -(-nverify // pipe.coalesce)),
            "batches_per_replay": pipe.coalesce, "all_equal_eager": bool(equal),
            "tail_outputs_compared": sorted(eager_tail[0]) if eager_tail and eager_tail[0] is not None else None,
            "output_sha1_replay": first[0], "output_sha1_eager": first[1],
            "note": "every batch through the pipeline (graph replay over batches_per_replay batches at once, all slots "
                    "busy with other batches) against the eager result of THAT batch alone on one stream; sha1 of the "
                    "[B,256,512] feature output of pool batch 0 both ways"}


def workload_backbone(args, sh, rank, world, dev, points, graphs_ok, tag, detector=False):
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    params = syn.random_backbone_params(arch)
    tail = None
    if detector:
        # points -> boxes (README.md:10 of the reference quotes frames/s of the whole detector): the 'Det'
        # head, anchor-free decode, BEV boxes, per-class NMS and the gather of the kept rows run as the executor's tail,
        # captured behind stage B (3dssd_amd/modeling/single_stage_detector.py)
        syn.random_head_params(512, 1, cfgs.KITTI_ANGLE_CLS_NUM, params=params)
        tail = pkg("modeling.single_stage_detector").detection_tail(
            cfgs.KITTI_3DSSD_HEAD, cls_num=1, angle_cls_num=cfgs.KITTI_ANGLE_CLS_NUM,
            max_output_size=cfgs.KITTI_MAX_OUTPUT_NUM, nms_threshold=cfgs.KITTI_NMS_THRESH)
    quota = cgroup_cpu_quota()
    graphs_note = None
    if args.graphs is None:
        # hipGraphs unless this process has less than two cores' worth of CPU quota: ROCm's graph runtime keeps a thread
        # spinning (0.8 cores) for as long as graph work is pending, which a tight CFS quota turns into 50-90 ms freezes of the
        # whole process (profiles/r05_stall_quota_final.txt); eager launches run at the same rate with 0.3 cores
        args.graphs = 0 if (quota is not None and quota < 2.0) else 1
        if not args.graphs:
            graphs_note = "eager launches chosen: cgroup CPU quota %.2f cores < 2 (hipGraph replays keep a runtime thread spinning)" % quota
    use_graphs = bool(args.graphs) and graphs_ok
    P = pkg("pipeline")
    capture_error = None

    def make(graphs):
        net = None
        if args.ffps_fly:
            # measurement only: the layer-2 F-FPS without the distance matrix, as a stage of its own on
            # ONE dedicated stream of the staged executor (its multi-workgroup launches must never overlap)
            net = pkg("backbone").SABackbone(arch, params, dev, cfgs.KITTI_MAX_TRANSLATE_RANGE, True, None, dfps_side_stream=5,
                                             ffps_fly=True)
        return P.SAPipeline(arch, params, dev, batch=args.batch, points=points, channels=4, streams=max(1, args.streams), net=net,
                            graphs=graphs, max_translate_range=cfgs.KITTI_MAX_TRANSLATE_RANGE,
                            coalesce=max(1, args.coalesce), mode=args.executor, linear_graphs=args.linear_graphs,
                            main_streams=args.main_streams, sampler_streams=args.sampler_streams, tail=tail)
    try:
        pipe = make(use_graphs)
    except Exception as e:  # noqa: BLE001
        if not use_graphs:
            raise
        # hipGraph capture failed on this box: the executor runs the same kernels eagerly on the same streams (measured at
        # the same throughput with packages of 128 frames) -- recorded in the line, never silent
        capture_error = repr(e)[:300]
        print("bench.py: hipGraph capture failed (%s); falling back to eager launches" % capture_error, file=sys.stderr)
        torch.cuda.synchronize()
        use_graphs = False
        pipe = make(False)
    C = pipe.coalesce
    net = pipe.net
    # this rank's frame pool: global frame f -> rank f mod world (weak scaling: `batch` frames per GPU per step);
    # `pool` distinct frames per GPU, resident; step i takes pool batch i mod nb
    nb = max(1, args.pool // args.batch)
    frames = sh.frames_of_rank(0, nb * args.batch * world, rank, world)
    batches = [torch.from_numpy(np.stack([syn.frame_of(args.data, f, points) for f in frames[i * args.batch:(i + 1) * args.batch]])).to(dev)
               for i in range(nb)]
    if args.host_input:
        # the PCIe-inclusive variant (never the headline): the pool lives in pinned HOST memory, submit() copies a batch
        # host -> device asynchronously on the executor's stream in front of the package
        sub = [b.cpu().pin_memory() for b in batches]
    else:
        sub = batches
    torch.cuda.synchronize()
    cursor = [0]

    def run(k):
        # k steps = k batches; a package is launched by the submit that fills it (`coalesce` batches), the last, partly
        # filled one by flush()
        tickets = [None] * k
        for i in range(k):
            tickets[i] = pipe.submit(sub[cursor[0] % nb], sync_source=False, defer_copy=True)   # the pool is never rewritten
            cursor[0] += 1
        pipe.flush()
        return tickets

    clocks = {}
    base = [None]
    watch = HostWatch(pipe)
    ev_per_region = 3 * (args.steps // C + 2)

    def prime():
        # untimed and independent of --warmup: every slot replayed at least twice on real frames, and the chip kept
        # busy for >= 150 ms (clocks, first-use costs of every slot's graphs).  The sysfs clock read sits HERE, far from t0.
        clocks["before"] = sclk_mhz(dev)
        t0 = time.perf_counter()
        n = 0
        while n < 2 * pipe.nslots * C or time.perf_counter() - t0 < 0.15:
            run(C)
            n += C
            if n % (pipe.nslots * C) == 0:
                (pipe.wait_idle if args.blocking_wait else pipe.drain)()
        pipe.drain()
        return {"batches": n, "packages": n // C, "wall_ms": round((time.perf_counter() - t0) * 1e3, 1),
                "note": "untimed, before the warm-up steps: every slot replayed >= 2x on pool frames, >= 150 ms of work"}

    rehearsal = []
    cpu0 = (time.process_time(), time.perf_counter(), _cgroup_cpu_stat())

    def sleeping_wait(t0):
        # the host sleeps through most of the window it expects (0.8 x the fastest rehearsal), then waits for the
        # executor's streams on blocking HIP events; the bracket's torch.cuda.synchronize() follows and returns at once
        if rehearsal:
            remain = 0.8e-3 * min(rehearsal) - (time.perf_counter() - t0)
            if remain > 3e-4:
                time.sleep(remain)
        pipe.wait_idle()
    idle_wait = sleeping_wait if args.blocking_wait else None

    def rehearse():
        # untimed dress rehearsals of the timed region: the SAME calls (K submits, the package sizes K produces, timing
        # events on, events from the pool, host stamps on, collector frozen, the same waits) between the same synchronize
        # brackets, so that the timed region runs no code path, touches no page and creates no runtime object for the
        # first time.  Repeated until steady: at least --rehearse times, and on until the last three are within 6 % of the
        # fastest seen (a host or device disturbance in a rehearsal -- another tenant's burst, a monitoring sample --
        # postpones the timed region by a few windows instead of landing in it), at most --rehearse-max times.
        lo, hi = max(0, args.rehearse), max(args.rehearse, args.rehearse_max)
        t_start = time.perf_counter()
        while len(rehearsal) < hi:
            if len(rehearsal) >= lo and (lo == 0 or (len(rehearsal) >= 3 and max(rehearsal[-3:]) <= 1.06 * min(rehearsal))
                                         or time.perf_counter() - t_start > 3.0):
                break
            pipe.reserve_events(ev_per_region)
            pipe.record_timeline = True
            pipe.host_trace = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            if idle_wait is not None:
                idle_wait(t0)
            torch.cuda.synchronize()
            rehearsal.append(round((time.perf_counter() - t0) * 1e3, 3))
            pipe.host_trace = None
            pipe.record_timeline = False
            pipe.timeline(None)

    def before():
        pipe.reserve_events(ev_per_region + 1)
        pipe.record_timeline = True
        base[0] = pipe._event(True)
        base[0].record()
        torch.cuda.synchronize()

    def after():
        pipe.record_timeline = False
        clocks["after"] = sclk_mhz(dev)

    t_max, frames_total, host_issue_ms, tickets, primed = timed_region(sh, dev, run, args.steps, args.warmup, args.batch,
                                                                       prime, before, after, rehearse, watch, idle_wait)
    host = watch.summary()
    cg1 = _cgroup_cpu_stat()
    host["cgroup_cpu_quota_cores"] = cgroup_cpu_quota()
    if cpu0[2] is not None and cg1 is not None:          # prime + warm-up + rehearsals + timed region
        host["cgroup_nr_throttled_since_priming"] = cg1[0] - cpu0[2][0]
        host["cgroup_throttled_us_since_priming"] = cg1[1] - cpu0[2][1]
    host["process_cpu_cores_since_priming"] = round((time.process_time() - cpu0[0]) / max(time.perf_counter() - cpu0[1], 1e-9), 2)
    packages = pipe.timeline(base[0])
    x_last, f_last = tickets[-1].result()
    assert f_last.shape == (args.batch, 256, 512) and x_last.shape == (args.batch, 256, 3)
    detections = None
    if detector:
        d_last = tickets[-1].detections()
        assert d_last["pred_3d_bbox"].shape == (args.batch, cfgs.KITTI_MAX_OUTPUT_NUM, 7)
        detections = {"boxes_kept_per_frame_last_batch": d_last["nms_cnt"].reshape(-1).cpu().tolist(),
                      "max_output_num": cfgs.KITTI_MAX_OUTPUT_NUM, "nms_threshold": cfgs.KITTI_NMS_THRESH,
                      "outputs": "pred_3d_bbox [B,%d,7], pred_3d_score, pred_3d_cls_category (lib/builder/postprocessor.py:90-118), "
                                 "fixed size, zero rows behind the count" % cfgs.KITTI_MAX_OUTPUT_NUM}
    # configs[3] "RCCL result gather" (untimed): every rank's last batch of outputs all-gathered and checked by digest
    gathered = sh.gather_check(x_last, f_last) if args.gather else None
    overlap = overlap_probe(pipe, run, min(args.steps, 96)) if rank == 0 else None
    # latency of one batch alone on the device (no overlap), for the record
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(3):
        pipe.run_alone(sub[i % nb])
    latency_ms = (time.perf_counter() - t1) / 3 * 1e3          # one replay (batch x coalesce frames) alone on the device
    verify = verify_pipeline(pipe, batches, args.verify, sub)
    if verify is not None and not verify["all_equal_eager"]:
        sys.exit("bench.py: a pipeline output differs from the eager result of the same batch -- refusing to report")
    other = None
    if use_graphs and not args.no_other_executor and world == 1:
        # the other executor on the same workload, reported beside the headline.  A process of its own: the hardware
        # queue count is fixed when the HIP runtime starts, and a second pipeline here would share queues with the first.
        torch.cuda.synchronize()
        alt = ({"executor": "slots", "streams": 16, "coalesce": 4} if args.executor == "staged" else
               {"executor": "staged", "streams": P.DEFAULT_PACKAGES, "coalesce": 16})
        cmd = [sys.executable, os.path.abspath(__file__), "--executor", alt["executor"], "--coalesce", str(alt["coalesce"]),
               "--streams", str(alt["streams"]), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--batch", str(args.batch), "--points", str(points), "--pool", str(args.pool),
               "--data", args.data, "--no-cpu-baseline", "--no-other-executor", "--profile-iters", "0", "--verify", "0",
               "--rehearse", str(args.rehearse), "--rehearse-max", str(args.rehearse_max), "--blocking-wait", str(args.blocking_wait)]
        if args.allow_knobs:
            cmd.append("--allow-knobs")
        try:
            env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES" or args.hwq_from_env}
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env).stdout.strip().splitlines()
            d1 = json.loads(out[-1])
            other = {"executor": alt, "value": d1["value"], "unit": d1["unit"], "steps": d1["steps"], "warmup": d1["warmup"],
                     "ms_per_step": d1["ms_per_step"], "hw_queues": d1["config"].get("hw_queues"),
                     "one_package_alone_ms": d1["config"].get("one_package_alone_ms"),
                     "note": "`bench.py --executor %s` in a process of its own right after the headline run: same kernels, "
                             "same frames" % alt["executor"]}
        except Exception as e:  # noqa: BLE001 -- the secondary figure must not take the headline down
            other = {"error": repr(e)}
    if rank != 0:
        return None
    clock_mhz = MAX_CLOCK_MHZ
    # the calls as the pipeline issues them: one replay = `coalesce` batches in one pass
    launch = [torch.cat([batches[(i * C + j) % nb] for j in range(C)]) for i in range(min(max(nb // C, 1), 4))]
    stages = profile_stages((lambda: pipe.tail(net(launch[0]))) if detector else (lambda: net(launch[0])), args.profile_iters)
    rows = mlp_row_stats(net, launch)
    fpl = args.batch * C                                        # frames per launch
    ms_step = t_max / args.steps * 1e3
    window_ms = t_max * 1e3
    line = {
        "metric": ("point-cloud frames/sec points -> boxes (SA backbone + Det head + decode + BEV NMS), KITTI 16384-pt" if detector else
                   "point-cloud frames/sec through full SA backbone, KITTI 16384-pt" if points == 16384 else
                   "point-cloud frames/sec through full SA backbone, %d-pt frames" % points),
        "value": round(frames_total / t_max, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "grouped MLP: fp16 x fp16 -> fp32 accumulate (one MFMA pass) on the scales whose contractions are all >= 128 "
                 "wide (layer3, layer4), split bf16 hi/lo (three passes) on the narrow scales and the aggregation layers; "
                 "fp32 for FPS / ball query / distance matrix",
        "data": "synthetic KITTI-shape frames (seeded, --data %s), %d DISTINCT frames per GPU cycled through the steps, "
                "random-init weights%s" % (args.data, nb * args.batch,
                                           "; INPUTS IN PINNED HOST MEMORY (--host-input): every step includes its PCIe copy"
                                           if args.host_input else ""),
        "config": dict(
            # flat scalars FIRST (the driver record keeps those): what was run, and what decides a short window
            [("workload", ("%s: 3DSSD points -> boxes (3dssd.yaml rows 1-6 + HEAD row, decode, per-class BEV NMS), %d-pt frames, batch=%d per GPU"
                           if detector else "%s: full 3DSSD SA backbone (3dssd.yaml rows 1-6), %d-pt frames, batch=%d per GPU")
               % (tag, points, args.batch)),
             ("executor", args.executor), ("hip_graphs", use_graphs), ("frames_per_launch", fpl),
             ("timed_window_ms", round(window_ms, 3)),
             ("probe_window_ms", overlap["window_ms"] if overlap else None),
             ("timed_over_probe", round(window_ms / overlap["window_ms"], 3) if overlap and overlap["window_ms"] > 0 and
              overlap["steps"] == args.steps else None)] +
            # the other claims of DESIGN.md, measured by THIS process right after the headline (run_extras),
            # and the two cgroup scalars that say whether the host was throttled -- all inside the driver's first ~20 keys
            [(k, None) for k in EXTRA_KEYS] +
            [("cgroup_cpu_quota_cores", host.get("cgroup_cpu_quota_cores")), ("cgroup_nr_throttled", host.get("cgroup_nr_throttled")),
             ("rehearsals", len(rehearsal)), ("rehearsal_ms_first", rehearsal[0] if rehearsal else None),
             ("rehearsal_ms_min", min(rehearsal) if rehearsal else None), ("rehearsal_ms_max", max(rehearsal) if rehearsal else None),
             ("rehearsal_ms_last", rehearsal[-1] if rehearsal else None), ("blocking_wait", bool(args.blocking_wait)),
             ("host_issue_ms_per_step", round(host_issue_ms, 4))] +
            [(k, v) for k, v in host.items() if k not in ("cgroup_cpu_quota_cores", "cgroup_nr_throttled")] +
            [("pkg%d_%s_ms" % (i, nm), ms[j]) for i, (_sl, _f, ms) in enumerate(packages[:2])
             for j, nm in ((0, "reached"), (len(ms) - 1, "done"))] +
            [("sclk_before", clocks.get("before")), ("sclk_after", clocks.get("after")),
             ("other_executor_value", other.get("value") if other else None),
             ("one_package_alone_ms", round(latency_ms, 3)), ("ramp_dominated", bool(window_ms < 20.0 * latency_ms)),
             ("steps_in_flight_mean", overlap["steps_in_flight_mean"] if overlap else None),
             ("ms_between_completions", overlap["ms_between_completions"] if overlap else None),
             ("frames_per_step_per_gpu", args.batch), ("data", args.data), ("pool_frames_per_gpu", nb * args.batch),
             ("inputs", "pinned host memory, copied per step (PCIe-inclusive)" if args.host_input else "resident in HBM"),
             ("slots", pipe.nslots), ("streams_used", pipe.streams_used()), ("hw_queues", P.hw_queues()),
             ("graph_capture_error", capture_error), ("graphs_note", graphs_note), ("ffps_fly", bool(args.ffps_fly)), ("batches_per_replay", C), ("linear_graphs", pipe.linear_graphs),
             ("gc", "frozen across the timed bracket"), ("load_avg_1min", round(os.getloadavg()[0], 2)), ("host_cores", os.cpu_count()),
             ("sharding", "frame f -> rank f mod N, no data-path collective"),
             ("executor_note", EXECUTOR_NOTES[args.executor] % {"C": C, "B": args.batch, "n": pipe.nslots}),
             # nested objects last
             ("package_sizes", pipe.sizes), ("priming", primed), ("rehearsal_ms", rehearsal),
             ("timed_packages_ms", {"columns": ["slot", "batches", "reached", "stage_A_done", "done"] if args.executor == "staged"
                                               else ["slot", "batches", "reached", "done"],
                                    "rows": [[i, f] + ms for i, f, ms in packages[:32]],
                                    "note": "device-side times (HIP events, ms since the start of the timed region) of "
                                            "the packages the timed steps ran in: reached by its stream, (layer-1 "
                                            "sampling done,) complete"}),
             ("other_executor", other), ("gather", gathered)]),
        "timed_window_ms": round(window_ms, 3),
        "single_stream_batch_latency_ms": round(latency_ms, 3),
        "latency_note": "one batch submitted alone and waited for: its package is launched at once, i.e. one pass over "
                        "frames_per_launch frames (the other parts of the package are copies of the batch)",
        "ramp_dominated": bool(window_ms < 20.0 * latency_ms),
        "ramp_note": "a run shorter than ~20 single-package latencies mostly measures filling and draining the executor "
                     "(every package starts with the ~3 ms layer-1 D-FPS); the steady-state rate needs --steps >= %d"
                     % (100 * C),
        "host_issue_ms_per_step": round(host_issue_ms, 3),
        "hip_graphs": use_graphs,
        "env_knobs": env_knobs()[0],
        "verify": verify,
        "other_executor": other,
        "mlp_rows_per_step": rows,
        "overlap": overlap,
    }
    if detections is not None:
        line["detections"] = detections
    if args.allow_shared_device and world > torch.cuda.device_count():
        line["shared_device"] = ("%d ranks on %d GPU(s): a functional check of the multi-rank path, NOT a scaling point"
                                 % (world, torch.cuda.device_count()))
    if stages:
        dom = max(stages, key=lambda s: s["avg_ms"] * s["calls_per_step"])
        mlp = [s for s in stages if s["kernel"] in MLP_CALLS]
        mlp_only_ms = sum(s["avg_ms"] * s["calls_per_step"] for s in mlp)
        plan_ms = sum(s["avg_ms"] * s["calls_per_step"] for s in stages if s["kernel"] in ("sa_group_mlp_plan", "sa_group_mlp_plan2"))
        mlp_ms = mlp_only_ms + plan_ms
        mlp_fl = sum(s["gflop"] * s["calls_per_step"] for s in mlp)
        bq = [s for s in stages if s["kernel"] in ("sa_query_ball_point_multi", "sa_query_ball_point_grid", "sa_query_ball_point_grid_ex")]
        bq_ms = sum(s["avg_ms"] * s["calls_per_step"] for s in bq)
        bq_mb = sum(s["mbytes"] * s["calls_per_step"] for s in bq)
        gflop_step = sum(s["gflop"] * s["calls_per_step"] for s in stages if s["kernel"] in MFMA_CALLS)
        mb_step = sum(s["mbytes"] * s["calls_per_step"] for s in stages)
        evaluated = None
        if dom["kernel"] in ("sa_fps_ex", "sa_fps_ex2", "sa_fps_ex3") and " c=3" in dom["label"]:
            m1 = int(dom["label"].split("->")[1].split()[0])
            evaluated = fps_bucket_evaluated(launch[0][:, :, :3].contiguous(), m1)
        line["roofline"] = roofline_of(dom, fpl, clock_mhz, evaluated)
        line["whole_step"] = {
            "mlp_gflop_algorithmic": round(gflop_step, 2),
            "mfma_tflops": round(gflop_step / (ms_step * C), 2), "mfma_frac_of_bf16_peak": round(gflop_step / (ms_step * C) / MFMA_BF16_PEAK_TF, 5),
            "algorithmic_mbytes": round(mb_step, 2), "hbm_gbs": round(mb_step / (ms_step * C), 2),
            "hbm_frac": round(mb_step / (ms_step * C) / HBM_PEAK_GBS, 5),
            "note": "algorithmic work of one replay = %d step(s) (SURVEY 8d) / (%d x ms_per_step) of the overlapped "
                    "multi-stream run" % (C, C)}
        ev_tf = rows["gflop_evaluated"] / mlp_ms if mlp_ms else 0.0
        line["roofline_grouped_mlp"] = {
            "bound": "mfma", "achieved": round(ev_tf, 3), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(ev_tf / MFMA_BF16_PEAK_TF, 5),
            "nominal_tflops": round(mlp_fl / mlp_ms, 3) if mlp_ms else 0.0,
            "nominal_frac": round(mlp_fl / mlp_ms / MFMA_BF16_PEAK_TF, 5) if mlp_ms else 0.0,
            "kernel_ms": round(mlp_only_ms, 5), "plan_ms": round(plan_ms, 5),
            "note": "achieved / frac = EXECUTED flops (rows the kernels run: the distinct rows of each ball in 8-row "
                    "granules) / single-stream time of the MLP kernels incl. the row-plan kernels; nominal_* counts the "
                    "reference's m x nsample rows (SURVEY 8d, 30.9 GFLOP/frame); issued MFMA flops = executed x 3 on the "
                    "split-bf16 scales, x 1 on the fp16 scales",
            "pmc": _pmc_mlp_util()}
        line["roofline_ball_query"] = {
            "bound": "hbm", "achieved": round(bq_mb / bq_ms, 2) if bq_ms else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(bq_mb / bq_ms / HBM_PEAK_GBS, 6) if bq_ms else 0.0,
            "north_star_target_40pct_hbm": "not met / not applicable on the benchmarked path: group_point is fused "
                                           "into the MLP gather (the 87 MB/frame grouped tensor is never written), and "
                                           "the grid ball query moves 19 MB per launch and is latency-bound, not "
                                           "bandwidth-bound"}
        line["stages"] = stages
        line["stages_note"] = ("stages, mlp_rows_per_step and the roofline objects describe the calls as the pipeline issues "
                               "them: ONE pass over frames_per_launch = %d frames (%d step(s)); calls_per_step counts calls per "
                               "such pass" % (fpl, C))
    if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only (the contract)
        line["cpu_baseline"] = cpu_baseline(arch, params, min(args.batch, 8), points, args.data)
    return line


def workload_ffps_isolated(args, sh, rank, world, dev):
    """BASELINE.json configs[2]: feature-distance FPS isolated, [32, 16384, 3+64] -> 4096 per frame (the fused
    on-the-fly form, SURVEY 8d: the matrix form would need a 1.07 GB matrix per frame)."""
    S, syn = pkg("utils.tf_ops.sampling.tf_sampling"), pkg("synthetic")
    batch, n, c, m = 32, 16384, 67, 4096
    rng = np.random.default_rng(20260925)
    frames = sh.frames_of_rank(0, batch * world, rank, world)
    xyz = np.stack([syn.frame_of(args.data, f, n)[:, :3] for f in frames])
    feat = rng.normal(0, 0.5, (len(frames), n, c - 3)).astype(np.float32)
    pts = torch.from_numpy(np.concatenate([xyz, feat], 2)).to(dev)

    def run(k):
        return [S.farthest_point_sample(m, pts) for _ in range(k)]

    t_max, frames_total, host_issue_ms, outs, _ = timed_region(sh, dev, run, args.steps, args.warmup, len(frames))
    assert outs[-1].shape == (len(frames), m)
    if rank != 0:
        return None
    clock_mhz = MAX_CLOCK_MHZ
    stages = profile_stages(lambda: S.farthest_point_sample(m, pts), max(1, min(args.profile_iters, 2)))
    line = {"metric": "frames/sec through feature-distance FPS 16384->4096 (3+64 channels), isolated",
            "value": round(frames_total / t_max, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t_max / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic KITTI-shape xyz + N(0,0.5) features (seeded)",
            "config": {"workload": "configs[2]: F-FPS isolated, [%d,%d,%d] -> %d per frame, batch=%d per GPU" % (batch, n, c, m, batch),
                       "frames_per_step_per_gpu": len(frames), "data": args.data, "share_grid": bool(args.share_grid)},
            "timed_window_ms": round(t_max * 1e3, 3),
            "host_issue_ms_per_step": round(host_issue_ms, 3), "env_knobs": env_knobs()[0]}
    if stages:
        line["roofline"] = roofline_of(max(stages, key=lambda s: s["avg_ms"]), len(frames), clock_mhz)
        line["stages"] = stages
    if not args.no_cpu_baseline and world == 1:
        from oracle import sa_oracle as O
        O.lib()
        nb = min(len(frames), os.cpu_count() or 1, 32)
        sub = pts[:nb].cpu().numpy()
        t0 = time.time()
        O.farthest_point_sample(m, sub)
        dt = time.time() - t0
        line["cpu_baseline"] = dict(value=round(nb / dt, 4), unit="frames/s", cores=min(os.cpu_count() or 1, nb), kind="port",
                                    sample="%d frames of the same workload through oracle.farthest_point_sample "
                                           "(OpenMP over frames), %.1f s" % (nb, dt))
    return line


def workload_group_materialised(args, sh, rank, world, dev):
    """The reference's UNFUSED neighbour-grouping sequence at the backbone's own shapes (layers_util.py:134-165):
    per SA layer the ball query of every band, then group_point(xyz, idx) and group_point(features, idx) with the
    grouped tensors [B,m,ns,C] materialised in HBM -- what BASELINE.json's "ball_query+group" HBM target refers to
    (87.44 MB per frame).  The benchmarked backbone never does this (the gather is fused into the MLP);
    this workload exists to put a measured HBM fraction next to that target."""
    cfgs, syn = pkg("configs"), pkg("synthetic")
    G = pkg("utils.tf_ops.grouping.tf_grouping")
    arch = cfgs.KITTI_3DSSD_ARCH
    net = pkg("backbone").SABackbone(arch, syn.random_backbone_params(arch), dev, cfgs.KITTI_MAX_TRANSLATE_RANGE)
    frames = sh.frames_of_rank(0, args.batch * world, rank, world)
    pts = torch.from_numpy(np.stack([syn.frame_of(args.data, f, args.points) for f in frames])).to(dev)
    xl, fl, _ = net(pts)
    torch.cuda.synchronize()
    jobs = []                     # (xyz [B,n,3], features [B,n,C], centres [B,m,3], bands)
    for li, row in enumerate(arch):
        radius, nsample, dilated, layer_type = row[2], row[3], row[13], row[11]
        if layer_type != "SA_Layer" or not radius:
            continue
        x_in, f_in = xl[row[0][0]], fl[row[1][0]]
        ctr = xl[li + 1] if row[14] == -1 else xl[row[14]]
        bands = [((0.0 if i == 0 or not dilated else float(radius[i - 1])), float(radius[i]), int(nsample[i])) for i in range(len(radius))]
        jobs.append((x_in.contiguous(), f_in.contiguous(), ctr.contiguous(), bands, dilated))

    by_alg = 0
    for x_in, f_in, ctr, bands, _d in jobs:          # inputs once + idx/cnt written + grouped tensors written once
        b, n, m, c = x_in.shape[0], x_in.shape[1], ctr.shape[1], f_in.shape[2]
        by_alg += b * (n * (3 + c) * 4 + m * 12)
        for _lo, _hi, ns in bands:
            by_alg += b * (m * ns * 4 + m * 4 + m * ns * (3 + c) * 4)

    import contextlib

    def step():
        outs = []
        # --share-grid: the bands of a layer query ONE point set that nothing rewrites in between -- what shared_grid() is for
        with (G.shared_grid() if args.share_grid else contextlib.nullcontext()):
            for x_in, f_in, ctr, bands, dilated in jobs:
                for lo, hi, ns in bands:
                    idx, cnt = (G.query_ball_point_dilated(lo, hi, ns, x_in, ctr) if dilated else G.query_ball_point(hi, ns, x_in, ctr))
                    outs.append((G.group_point(x_in, idx), G.group_point(f_in, idx)))
        return outs

    def run(k):
        # every step's grouped tensors are dropped when the next step starts (11 GB per step at 128 frames: keeping all
        # K steps alive made the allocator, not the kernels, set ms_per_step -- VERDICT r4 weak #10)
        outs = None
        for _ in range(k):
            outs = None
            outs = step()
        return outs

    t_max, frames_total, host_issue_ms, outs, _ = timed_region(sh, dev, run, args.steps, args.warmup, len(frames))
    del outs
    if rank != 0:
        return None
    stages = profile_stages(step, max(1, args.profile_iters))
    ms_step = t_max / args.steps * 1e3
    # device time of the whole sequence: the same calls captured into one hipGraph and replayed (no host launch gaps, no
    # per-call event floor: an event pair around a 2 us kernel reads 7-9 us)
    graph_ms = None
    try:
        side = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            keep = step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            g.replay()
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        graph_ms = e0.elapsed_time(e1) / 20
        del keep
    except Exception as e:  # noqa: BLE001
        graph_ms = None
        print("bench.py: graph timing of the group workload failed: %r" % (e,), file=sys.stderr)
    grp = [s for s in stages if s["kernel"] == "sa_group_point"]
    bq = [s for s in stages if s["kernel"].startswith("sa_query_ball")]
    grp_ms = sum(s["avg_ms"] * s["calls_per_step"] for s in grp)
    bq_ms = sum(s["avg_ms"] * s["calls_per_step"] for s in bq)
    gbs = by_alg / (grp_ms + bq_ms) / 1e6 if grp_ms + bq_ms > 0 else 0.0
    line = {"metric": "frames/sec through ball_query + group_point with materialised grouped tensors (reference op sequence), KITTI 16384-pt",
            "value": round(frames_total / t_max, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 / int32 copies",
            "data": "synthetic KITTI-shape frames (seeded, --data %s); layer inputs taken from one backbone forward" % args.data,
            "config": {"workload": "ball_query + group_point materialised at the shapes of 3dssd.yaml rows 1-3, 6 (one band per call, like the reference), batch=%d per GPU" % args.batch,
                       "frames_per_step_per_gpu": len(frames), "data": args.data, "share_grid": bool(args.share_grid)},
            "timed_window_ms": round(t_max * 1e3, 3), "host_issue_ms_per_step": round(host_issue_ms, 3), "env_knobs": env_knobs()[0],
            "roofline": {"kernel": "ball_query + group_point, all layers (one stream, HIP events per C-ABI call)", "bound": "hbm",
                         "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                         "traffic": None, "algorithmic_bytes": int(by_alg), "algorithmic_mb_per_frame": round(by_alg / len(frames) / 1e6, 2),
                         "kernel_ms": round(grp_ms + bq_ms, 5),
                         "sequence_as_one_graph": None if graph_ms is None else {
                             "ms": round(graph_ms, 5), "gbs": round(by_alg / graph_ms / 1e6, 1),
                             "frac": round(by_alg / graph_ms / 1e6 / HBM_PEAK_GBS, 5),
                             "note": "the same %d calls captured into ONE hipGraph, device time per replay (HIP events around "
                                     "20 replays): no host gaps and no per-call event floor" % sum(s["calls_per_step"] for s in grp + bq)},
                         "group_point_only": {"ms": round(grp_ms, 5),
                                              "gbs": round(sum(s["mbytes"] * s["calls_per_step"] for s in grp) / grp_ms, 1) if grp_ms else 0.0,
                                              "frac": round(sum(s["mbytes"] * s["calls_per_step"] for s in grp) / grp_ms / HBM_PEAK_GBS, 5) if grp_ms else 0.0},
                         "ball_query_only_ms": round(bq_ms, 5),
                         "note": "achieved = SURVEY 8d algorithmic bytes (87.4 MB per frame: inputs once, idx / cnt and the "
                                 "grouped tensors written once) / sum of the kernel times of one step; the host-bound eager "
                                 "issue (ms_per_step) is not the kernel time"},
            "stages": stages}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="configs1", choices=["configs1", "configs2", "configs4", "group", "detector"],
                    help="BASELINE.json configs[1] (default, the metric's configuration), configs[2], configs[4]; group: the "
                         "reference's unfused ball_query + group_point sequence with materialised grouped tensors; detector: "
                         "configs[1] + Det head + decode + per-class BEV NMS (points -> boxes) through the same executor")
    ap.add_argument("--extras-budget", type=float, default=75.0,
                    help="seconds for the secondary measurements the configs[1] line carries as flat scalars (steady state, rings64, "
                         "dense, detector, configs[2] / [4], group HBM fractions, RCCL smoke), each a sub-run of this script; 0: none")
    ap.add_argument("--ffps-fly", type=int, default=0,
                    help="1: layer-2 F-FPS without the distance matrix (csrc/ffps_fly.hip) as a fourth stage on a stream of its own "
                         "(staged executor only; a measurement, not the default: DESIGN.md section 6)")
    ap.add_argument("--share-grid", action="store_true",
                    help="--workload group: the per-band ball-query calls of a layer share one grid (tf_grouping.shared_grid(); off by default)")
    ap.add_argument("--data", default="default", choices=list(pkg("synthetic").DATA_VARIANTS),
                    help="default: SURVEY 8d generator; dup10: 10 %% duplicated rows (KITTI padding); dense: uniform box, every ball full")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--pool", type=int, default=None, help="distinct frames per GPU the steps cycle through")
    ap.add_argument("--executor", default=None, choices=["staged", "slots"],
                    help="3dssd_amd/pipeline.py mode: staged (default; 3 streams, sampler stage ‖ the rest) or slots (one stream + graph per slot)")
    ap.add_argument("--streams", type=int, default=None, help="pipeline slots: packages in the ring (staged) / HIP streams (slots)")
    ap.add_argument("--coalesce", type=int, default=None, help="batches per package (frames per replay = batch x coalesce)")
    ap.add_argument("--linear-graphs", type=int, default=None, help="1: the F-FPS || D-FPS launch on the capturing stream (captured graphs are linear chains; default for staged), 0: on a helper-stream branch (default for slots)")
    ap.add_argument("--main-streams", type=int, default=2, help="staged executor: streams stage B alternates between")
    ap.add_argument("--sampler-streams", type=int, default=1, help="staged executor: streams stage A alternates between")
    ap.add_argument("--hw-queues", type=int, default=None, help="GPU_MAX_HW_QUEUES for this run (default: unset for staged, = --streams for slots)")
    ap.add_argument("--verify", type=int, default=None, help="batches re-run through the pipeline and compared with eager (0: skip)")
    ap.add_argument("--profile-iters", type=int, default=3)
    ap.add_argument("--graphs", type=int, default=None, help="1: captured hipGraphs (the default); 0: eager launches on the same streams (same throughput; "
                         "the default under a CPU quota below 2 cores: a process replaying hipGraphs burns 0.8 cores in a runtime thread while graph work is pending)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-executor", action="store_true", help="skip the secondary measurement with the other executor")
    ap.add_argument("--allow-knobs", action="store_true", help="run although SA_* / SA3D_* environment variables are set (recorded in the line)")
    ap.add_argument("--allow-shared-device", action="store_true",
                    help="--gpus N with fewer than N GPUs visible: ranks share devices (functional check of the multi-rank path)")
    ap.add_argument("--host-input", action="store_true", help="frame pool in pinned host memory: every step pays its host->device copy (PCIe-inclusive rate; not the headline)")
    ap.add_argument("--gather", type=int, default=None, help="after the timed region all-gather every rank's last batch of outputs and check rank order by sha1 (default: on when --gpus > 1)")
    ap.add_argument("--rehearse", type=int, default=3, help="untimed dress rehearsals of the timed region right before it (same calls, same brackets): at least this many (0: none)")
    ap.add_argument("--rehearse-max", type=int, default=40, help="... and on until the last three are within 6 %% of the fastest, at most this many")
    ap.add_argument("--blocking-wait", type=int, default=1, help="1: wait for the executor's streams asleep (blocking HIP events) in front of every torch.cuda.synchronize() of the rehearsals and the timed bracket; 0: spin")
    ap.add_argument("--cpu-affinity", default=None, help="pin this rank's issuing thread: 'auto' = core LOCAL_RANK x (cores / ranks), or a core number")
    ap.add_argument("--launch-check", action="store_true", help="exercise only the launch path (works without a GPU)")
    args = ap.parse_args()
    assert args.gpus >= 1
    defaults = {"configs1": dict(steps=512, warmup=64, batch=8, points=16384, pool=256, verify=64, executor="staged"),
                "configs2": dict(steps=4, warmup=1, batch=32, points=16384, streams=1, pool=32, verify=0, coalesce=1, executor="slots"),
                "configs4": dict(steps=16, warmup=4, batch=16, points=65536, streams=4, pool=96, verify=8, coalesce=2, executor="staged"),
                "group": dict(steps=20, warmup=5, batch=8, points=16384, streams=1, pool=8, verify=0, coalesce=1, executor="slots"),
                "detector": dict(steps=512, warmup=64, batch=8, points=16384, pool=256, verify=64, executor="staged")}[args.workload]
    for k, v in defaults.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.gather is None:
        args.gather = 1 if args.gpus > 1 else 0
    if args.streams is None:
        args.streams = 4 if args.executor == "staged" else 16
    if args.coalesce is None:
        args.coalesce = 16 if args.executor == "staged" else 4
    # hardware queues: fixed when the HIP runtime starts, so before the first CUDA call of this process
    args.hwq_from_env = "GPU_MAX_HW_QUEUES" in os.environ
    if args.hw_queues is not None:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)
    elif args.executor == "slots" and args.workload in ("configs1", "detector") and not args.launch_check:
        pkg("pipeline").request_hw_queues(args.streams)

    knobs, sa_keys = env_knobs()
    if "SA_ABLATE" in os.environ:
        sys.exit("bench.py: SA_ABLATE is set -- ablation runs skip kernels and are not benchmark results (tools/ablate.sh "
                 "drives them through tools/, never through this script)")
    if sa_keys and not args.allow_knobs:
        sys.exit("bench.py: kernel-selection variables set in the environment (%s): refusing to measure a non-default "
                 "configuration; pass --allow-knobs to run anyway (they are recorded in the line)" % ", ".join(sa_keys))

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_spawn(args)                                # does not return

    sh = pkg("sharding")
    if args.launch_check:
        return launch_check(args, sh)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    ndev = torch.cuda.device_count()
    rank, local_rank, world = sh.env_world()
    if local_rank >= ndev and not args.allow_shared_device:
        sys.exit("rank %d has no GPU (%d visible)" % (local_rank, ndev))
    device_index = local_rank % ndev
    rank, local_rank, world = sh.init(backend="gloo" if (args.allow_shared_device and world > ndev) else None,
                                      device_index=device_index)
    assert world == args.gpus, ("bench.py --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus N` or "
                                "torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    # the issuing thread on a core of its own (the HIP runtime's helper threads keep the process-wide mask they were
    # created with, so this runs AFTER set_device created them)
    pin_rank(args, local_rank, world)
    pkg("utils._native").lib()

    if args.workload == "configs1":
        line = workload_backbone(args, sh, rank, world, dev, args.points, True, "configs[1]")
        if rank == 0 and world == 1 and args.extras_budget > 0 and not args.host_input:
            run_extras(args, line)
    elif args.workload == "detector":
        line = workload_backbone(args, sh, rank, world, dev, args.points, True, "detector", detector=True)
    elif args.workload == "configs4":
        # 65536-pt frames: layer-1 FPS is the multi-workgroup kernel (fps_coop.hip); captured as a plain launch, every
        # such launch on the staged executor's one sampler stream (--executor slots falls back to eager launches)
        line = workload_backbone(args, sh, rank, world, dev, args.points, args.executor == "staged", "configs[4]")
    elif args.workload == "group":
        line = workload_group_materialised(args, sh, rank, world, dev)
    else:
        line = workload_ffps_isolated(args, sh, rank, world, dev)
    if rank == 0:
        print(json.dumps(line))
    sh.barrier()


if __name__ == "__main__":
    main()
