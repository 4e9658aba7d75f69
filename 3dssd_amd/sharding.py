"""Frame sharding across the GPUs of one node.

Every op of the SA path indexes its own batch item only and inference BatchNorm uses frozen statistics,
so frames are independent units: frame f goes to rank f mod world_size, weights are replicated, and
there is NO collective on the data path.  torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests) is used only to agree on the wall time (max over ranks), to add up the
frame counts, and optionally to gather the small backbone outputs on rank 0.
"""
import hashlib
import os
import time

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device_index=None):
    """Process group of the node's ranks.  backend None: "nccl" (= RCCL) on GPUs, "gloo" on CPU.  device_index: the GPU
    this rank uses (default LOCAL_RANK); ranks that SHARE a device (a functional check on a box with fewer GPUs than
    ranks) must use "gloo" -- RCCL refuses two ranks on one device."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank if device_index is None else device_index)   # RCCL binds its communicator to the current device
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def frames_of_rank(first_frame, n_frames, rank, world):
    """Global frame ids handled by `rank`: f with f mod world == rank (round-robin)."""
    return [f for f in range(first_frame, first_frame + n_frames) if f % world == rank]


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def reduce_timing(elapsed_s, frames_done, device="cpu"):
    """(max elapsed over ranks, total frames over ranks).  `device`: where the two scalars live for the reduction (the
    rank's GPU under RCCL; ignored -- CPU -- under gloo)."""
    if not dist.is_initialized():
        return float(elapsed_s), int(frames_done)
    if dist.get_backend() != "nccl":
        device = "cpu"
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    c = torch.tensor([float(frames_done)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(c.item()))


def gather_outputs(xyz, feat):
    """all_gather of the per-rank backbone outputs ([B_local,256,3], [B_local,256,512]) -> lists ordered
    by rank.  ~0.53 MB per frame; not part of the timed path."""
    if not dist.is_initialized():
        return [xyz], [feat]
    world = dist.get_world_size()
    xs = [torch.empty_like(xyz) for _ in range(world)]
    fs = [torch.empty_like(feat) for _ in range(world)]
    dist.all_gather(xs, xyz.contiguous())
    dist.all_gather(fs, feat.contiguous())
    return xs, fs


def _digest(xyz, feat):
    h = hashlib.sha1()
    h.update(xyz.detach().cpu().contiguous().numpy().tobytes())
    h.update(feat.detach().cpu().contiguous().numpy().tobytes())
    return h.digest()


def gather_check(xyz, feat):
    """BASELINE.json configs[3] "RCCL result gather", with proof that it happened: all-gather this rank's last batch of
    backbone outputs (gather_outputs), all-gather the 20-byte sha1 every rank computed of ITS OWN tensors, and check
    that the tensor received in position r hashes to rank r's digest (rank order, no mix-up); `ranks_seen` is an
    all-reduce(sum) of ones.  Not part of the timed path (lib/core/trainer.py:120-155 moves tower outputs likewise
    outside any timing).  Under gloo (CPU tests; two ranks sharing one GPU) tensors travel through host memory."""
    if not dist.is_initialized():
        return {"ranks_seen": 1, "world": 1, "backend": None, "rank_order_ok": True, "bytes_gathered": 0, "ms": 0.0}
    backend = dist.get_backend()
    world = dist.get_world_size()
    on_gpu = backend == "nccl"
    x = xyz.contiguous() if on_gpu else xyz.detach().cpu().contiguous()
    f = feat.contiguous() if on_gpu else feat.detach().cpu().contiguous()
    dev = x.device
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    xs, fs = gather_outputs(x, f)
    mine = torch.tensor(list(_digest(x, f)), dtype=torch.uint8, device=dev)
    digs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(digs, mine)
    ones = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    if on_gpu:
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    ok = all(bytes(digs[r].cpu().tolist()) == _digest(xs[r], fs[r]) for r in range(world))
    distinct = len({bytes(d.cpu().tolist()) for d in digs})
    nbytes = sum(t.numel() * t.element_size() for t in xs + fs)
    return {"ranks_seen": int(ones.item()), "world": world, "backend": backend + (" (RCCL)" if on_gpu else ""),
            "rank_order_ok": bool(ok), "distinct_rank_digests": distinct, "bytes_gathered": int(nbytes), "ms": round(ms, 3)}


# ---- RCCL smoke on ONE GPU -------------------------------------------------------------------------------
# No lease of this build ever had more than one GPU, so backend "nccl" (= RCCL) had never initialised anywhere.  A
# world-size-1 communicator proves the pieces an 8-GPU run depends on before such a node appears: librccl loads, its
# kernels run on gfx950, the HSA_ENABLE_IPC_MODE_LEGACY setting bench.py exports is accepted, and reduce_timing /
# gather_check (the only collectives of this path; lib/core/trainer.py:120-155 is the reference's analogue) work on
# DEVICE tensors.  It is not a scaling point.
def rccl_smoke(device_index=0):
    """Run in a process of its own (it creates and destroys the default process group).  -> dict, "status": "ok" on success."""
    import socket
    assert torch.cuda.is_available(), "rccl_smoke needs a GPU"
    assert not dist.is_initialized(), "rccl_smoke owns the default process group: call it in a fresh process"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    t0 = time.perf_counter()
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        barrier()
        t_max, frames = reduce_timing(1.25, 160, device=dev)                     # all_reduce MAX + SUM on device scalars
        g = torch.Generator(device="cpu").manual_seed(6)
        xyz = torch.randn(8, 256, 3, generator=g).to(dev)
        feat = torch.randn(8, 256, 512, generator=g).to(dev)
        chk = gather_check(xyz, feat)                                             # all_gather x3 + all_reduce, digests compared
        big = torch.ones(1 << 22, dtype=torch.float32, device=dev)               # a bandwidth-sized all_reduce (16 MiB)
        dist.all_reduce(big)
        torch.cuda.synchronize()
        ok = (t_max == 1.25 and frames == 160 and chk["ranks_seen"] == 1 and chk["rank_order_ok"] and
              chk["backend"].startswith("nccl") and float(big[0].item()) == 1.0)
        ver = None
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            pass
        return {"status": "ok" if ok else "wrong result", "backend": dist.get_backend(), "world": 1, "rccl_version": ver,
                "reduce_timing": [t_max, frames], "gather_check": chk, "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                "seconds": round(time.perf_counter() - t0, 2)}
    finally:
        dist.destroy_process_group()


def rccl_smoke_main():
    import json
    try:
        out = rccl_smoke()
    except Exception as e:  # noqa: BLE001 -- the caller reads the line, not the exit code
        out = {"status": "failed: %r" % (e,)}
    print("RCCL_SMOKE " + json.dumps(out), flush=True)


def rccl_smoke_subprocess(timeout_s=120):
    """rccl_smoke() in a child process (a hang or crash inside RCCL cannot take the caller down)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import importlib, sys; sys.path.insert(0, %r); importlib.import_module('3dssd_amd.sharding').rccl_smoke_main()" % root
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"status": "timeout after %d s" % timeout_s}
    for ln in reversed(p.stdout.splitlines()):
        if ln.startswith("RCCL_SMOKE "):
            return json.loads(ln[len("RCCL_SMOKE "):])
    return {"status": "failed: no result line (rc %d): %s" % (p.returncode, (p.stderr or "")[-300:])}
