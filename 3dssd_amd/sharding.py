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
