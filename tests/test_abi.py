"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/sa_ops.h declares; the ctypes table matches the header; the Python operator modules expose the
reference's function names in both import styles."""
import ctypes
import importlib
import os
import re
import sys

import pytest

from conftest import ROOT, pkg


def _header_functions(name="sa_ops.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"\b(?:int|unsigned long)\s+(sa_\w+)\s*\(", src)


def test_library_exports_every_declared_symbol():
    native = pkg("utils._native")
    assert os.path.exists(native.LIB_PATH), "HIP extension not built (run __graft_entry__.build())"
    h = ctypes.CDLL(native.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(h, n), "lib3dssd_sa.so does not export %s" % n


def test_extra_library_is_separate_and_complete():
    """The reference's out-of-scope operators (points pooling, evaluation IoU) live in lib3dssd_extra.so /
    include/sa_extra.h; the product library exports none of them."""
    native = pkg("utils._native")
    assert os.path.exists(native.EXTRA_LIB_PATH), "lib3dssd_extra.so not built (make -C 3dssd_amd/csrc extra)"
    h, product = ctypes.CDLL(native.EXTRA_LIB_PATH), ctypes.CDLL(native.LIB_PATH)
    names = _header_functions("sa_extra.h")
    assert sorted(names) == sorted(native.EXTRA_SIGNATURES)
    for n in names:
        assert hasattr(h, n) and not hasattr(product, n), n
    native.lib_extra()


def test_ctypes_table_matches_header():
    native = pkg("utils._native")
    non_status = ["sa_query_ball_point_grid_ws_bytes", "sa_calc_square_dist_ws_bytes", "sa_group_mlp_max_ws_bytes", "sa_group_mlp_gemm_ws_bytes", "sa_ffps_fly_ws_bytes", "sa_host_crc32c", "sa_group_mlp_granule_rows"]      # return a size / a checksum / a granule size
    assert sorted(list(native.SIGNATURES) + non_status) == sorted(_header_functions())
    native.lib()       # resolves every symbol and sets argtypes


def test_product_library_reads_no_environment():
    """The kernel-selection knobs (SA_KNOB, csrc/sa_common.h) exist only in `make TUNE=1` builds: the shipped library
    imports no getenv and holds no SA_* variable name, so its behaviour is a function of its arguments only."""
    import subprocess
    native = pkg("utils._native")
    syms = subprocess.run(["nm", "-D", "--undefined-only", native.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms
    blob = open(native.LIB_PATH, "rb").read()
    for name in (b"SA_MLP_", b"SA_FPS_", b"SA_SQDIST_", b"SA_BQ_", b"SA_ABLATE"):
        assert name not in blob, name
    # and neither does the Python package (bench.py records / refuses what is left: SA3D_LIB selects a tuning build)
    import glob
    for f in glob.glob(os.path.join(ROOT, "3dssd_amd", "**", "*.py"), recursive=True):
        src = open(f).read()
        if f.endswith(os.path.join("utils", "_native.py")) or f.endswith("sharding.py") or f.endswith("pipeline.py"):
            continue                                   # SA3D_LIB / torchrun's RANK.. / GPU_MAX_HW_QUEUES default
        assert not re.search(r"os\.environ|getenv\(|environ\.get", src), f


def test_missing_library_fails_loudly(monkeypatch):
    native = pkg("utils._native")
    monkeypatch.setattr(native, "_LIB", None)
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/lib3dssd_sa.so")
    with pytest.raises(native.NativeLibraryError, match="no CPU fallback"):
        native.lib()


def test_reference_api_names_and_drop_in_import_style():
    s = pkg("utils.tf_ops.sampling.tf_sampling")
    g = pkg("utils.tf_ops.grouping.tf_grouping")
    for name in ("farthest_point_sample", "farthest_point_sample_with_distance", "gather_point"):
        assert callable(getattr(s, name))
    for name in ("query_ball_point", "query_ball_point_dilated", "group_point"):
        assert callable(getattr(g, name))
    # the reference's own import lines (lib/utils/layers_util.py:6-7) with 3dssd_amd/ on sys.path
    sys.path.insert(0, os.path.join(ROOT, "3dssd_amd"))
    try:
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            del sys.modules[k]
        m1 = importlib.import_module("utils.tf_ops.sampling.tf_sampling")
        m2 = importlib.import_module("utils.tf_ops.grouping.tf_grouping")
        m3 = importlib.import_module("utils.layers_util")
        assert callable(m1.farthest_point_sample) and callable(m2.query_ball_point_dilated)
        assert callable(m3.pointnet_sa_module_msg) and callable(m3.vote_layer)
    finally:
        sys.path.remove(os.path.join(ROOT, "3dssd_amd"))
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            del sys.modules[k]


def test_cpu_tensors_are_rejected_not_silently_computed():
    import torch
    s = pkg("utils.tf_ops.sampling.tf_sampling")
    with pytest.raises(ValueError, match="no CPU fallback"):
        s.farthest_point_sample(4, torch.zeros((1, 16, 3)))
    with pytest.raises(ValueError, match="positive npoint"):
        s.farthest_point_sample(0, torch.zeros((1, 16, 3)))


def test_prob_sample_name_exists_and_says_why_it_is_not_provided():
    # code written against the reference's tf_sampling module must get a clear error, not an AttributeError
    s = pkg("utils.tf_ops.sampling.tf_sampling")
    with pytest.raises(NotImplementedError, match="outside the set-abstraction path"):
        s.prob_sample(None, None)


def test_ball_query_workspace_holds_the_grids_and_one_record_per_query():
    # round 6 (csrc/ballquery_grid.hip): behind the frames' grids (128 x 128 cell starts, the cell lists as 16-byte points,
    # parameters, one far point) the workspace holds one 48-byte record per query -- its size depends on m; a host-only call
    lib = ctypes.CDLL(pkg("utils._native").LIB_PATH)
    f = lib.sa_query_ball_point_grid_ws_bytes
    f.argtypes, f.restype = [ctypes.c_int] * 3, ctypes.c_size_t
    assert f(0, 16384, 4096) == 0 and f(2, 0, 4096) == 0 and f(2, 16384, 0) == 0
    base = f(3, 16384, 1)
    assert f(3, 16384, 4097) - f(3, 16384, 1) == 3 * 4096 * 48
    assert base - 3 * 48 == 3 * 4 * (128 * 128 + 4 + 4 * 16384 + 8)
    assert f(3, 16384, 4096) % 16 == 0 and (f(1, 1000, 7) - 7 * 48) % 16 == 0        # 16-byte aligned pieces
