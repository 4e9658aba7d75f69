"""CPU-side checks of 3dssd_amd/pipeline.py: what the module does NOT do at import (it used to write
GPU_MAX_HW_QUEUES into os.environ), the explicit queue request of the 16-slot mode, and the loud failure without a GPU."""
import importlib
import os
import subprocess
import sys

import pytest

from conftest import ROOT, pkg


def test_importing_the_pipeline_leaves_the_environment_alone():
    code = ("import os, sys, importlib; sys.path.insert(0, %r); before = dict(os.environ); "
            "importlib.import_module('3dssd_amd.pipeline'); "
            "changed = {k: v for k, v in os.environ.items() if before.get(k) != v}; print(sorted(changed))" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "[]"


def test_request_hw_queues_sets_the_variable_before_the_runtime_starts(monkeypatch):
    P = pkg("pipeline")
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("the HIP runtime is already up in this process")
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert P.hw_queues() == 4                                   # ROCm's default
    assert P.request_hw_queues(16) is True and os.environ["GPU_MAX_HW_QUEUES"] == "16" and P.hw_queues() == 16
    assert P.request_hw_queues(8) is False                      # an earlier request (or the launcher's setting) stands
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "oops")
    assert P.hw_queues() == 4


def test_pipeline_needs_a_gpu_and_a_known_mode():
    P, cfgs, syn = pkg("pipeline"), pkg("configs"), pkg("synthetic")
    with pytest.raises(ValueError, match="no CPU fallback"):
        P.SAPipeline(cfgs.KITTI_3DSSD_ARCH, {}, "cpu")
    import torch
    if not torch.cuda.is_available():
        with pytest.raises((ValueError, Exception)):
            P.SAPipeline(cfgs.KITTI_3DSSD_ARCH, {}, "cuda:0", mode="rings")
