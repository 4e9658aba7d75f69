"""`python bench.py --gpus N` launches N ranks itself (one process per GPU, lib/core/trainer.py:120-155's N towers)
and refuses to run fewer than asked.  The launch path is exercised here on the CPU (gloo rendezvous on 127.0.0.1)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*flags):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), capture_output=True, text=True,
                          env=env, timeout=300)


def test_self_spawn_launches_the_requested_number_of_ranks():
    r = _run("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["launch_check"] is True and line["n_gpus"] == 2
    assert line["frames_total"] == 16 and line["t_max"] == 2.0      # 8 frames per rank; max over ranks of 1 + rank
    g = line["gather"]                                              # configs[3]'s result gather, [8,256,3] + [8,256,512] per rank
    assert g["ranks_seen"] == 2 and g["rank_order_ok"] is True and g["distinct_rank_digests"] == 2
    assert g["bytes_gathered"] == 2 * 8 * 256 * (3 + 512) * 4


def test_eight_rank_launch_rendezvous_partition_and_gather():
    # the 8-GPU node's launch path without the hardware -- `bench.py --gpus 8 --launch-check` spawns 8
    # processes (gloo here, RCCL on the GPU box), they rendezvous on 127.0.0.1, every rank takes its f mod 8 share of the
    # frames (configs[3]: batch 64 over 8 GPUs = 8 per rank), and the result gather sees all 8 ranks in rank order.
    r = _run("--gpus", "8", "--launch-check", "--cpu-affinity", "auto")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["launch_check"] is True and line["n_gpus"] == 8
    assert line["frames_total"] == 64 and line["t_max"] == 8.0      # max over ranks of 1 + rank
    assert line["frames_per_rank"] == [8] * 8 and line["partition_ok"] is True
    g = line["gather"]
    assert g["ranks_seen"] == 8 and g["rank_order_ok"] is True and g["distinct_rank_digests"] == 8
    assert g["bytes_gathered"] == 8 * 8 * 256 * (3 + 512) * 4
    assert len(set(line["cores"])) == min(8, os.cpu_count() or 1)    # --cpu-affinity auto: ranks on different cores


def test_more_gpus_than_visible_fails_loudly():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run("--gpus", str(have + 2), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in r.stderr


def test_bench_refuses_kernel_selection_variables_and_ablation():
    # ablation / kernel-selection switches must not leak into a benchmark line.  bench.py exits before touching
    # the GPU when SA_ABLATE is set (always) or when any SA_* / SA3D_* variable is set without --allow-knobs.
    for var, extra, msg in (("SA_ABLATE", [], "SA_ABLATE is set"), ("SA_ABLATE", ["--allow-knobs"], "SA_ABLATE is set"),
                            ("SA_MLP_WIDE", [], "refusing to measure"), ("SA3D_LIB", [], "refusing to measure")):
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        env[var] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"] + extra,
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and msg in (r.stderr + r.stdout), (var, extra, r.stderr[-500:])
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no line was printed
