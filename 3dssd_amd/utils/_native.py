"""ctypes binding of lib3dssd_sa.so (the C ABI declared in include/sa_ops.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared object is
missing or a symbol cannot be resolved, importing/calling fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SA3D_LIB selects another build of the same library (kernel A/B experiments); default: the in-tree one
LIB_PATH = os.environ.get("SA3D_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "lib3dssd_sa.so")

_c_int, _c_long, _c_float, _vp = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p

# name -> argtypes; every function returns int status.  Pointers are passed as raw addresses.
SIGNATURES = {
    "sa_farthest_point_sample": [_c_int] * 4 + [_vp, _vp, _vp, _vp],
    "sa_farthest_point_sample_with_distance": [_c_int] * 3 + [_vp, _vp, _vp, _vp],
    "sa_gather_point": [_c_int] * 4 + [_vp, _vp, _vp, _vp],
    "sa_query_ball_point": [_c_int] * 3 + [_c_float, _c_int, _vp, _vp, _vp, _vp, _vp],
    "sa_query_ball_point_dilated": [_c_int] * 3 + [_c_float, _c_float, _c_int, _vp, _vp, _vp, _vp, _vp],
    "sa_group_point": [_c_int] * 5 + [_vp, _vp, _vp, _vp],
    "sa_calc_square_dist": [_c_int] * 4 + [_vp, _vp, _vp, _vp],
    "sa_fps_ex": [_c_int] * 4 + [_vp, _vp, _vp, _c_int, _c_int, _vp],
    "sa_fps_with_distance_ex": [_c_int] * 3 + [_vp, _vp, _vp, _c_int, _c_int, _vp],
    "sa_copy_blocks": [_c_int, _vp, _vp],
    "sa_copy_batches": [_c_int, _vp, _vp, _c_long, _vp],
    "sa_fps_dual_ex": [_c_int] * 3 + [_vp, _vp, _c_int, _c_int, _vp, _c_long, _vp, _c_long, _c_int, _c_int, _vp, _c_long, _vp,
                       _c_int, _c_int, _vp, _c_long, _vp],
    "sa_group_mlp_max_layer": [_c_int] * 4 + [_vp, _c_int, _vp, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _vp,
                               _vp, _vp, _vp, _vp],
    "sa_fps_ex2": [_c_int] * 4 + [_vp, _c_long, _vp, _vp, _c_int, _c_int, _vp, _c_long, _vp],
    "sa_fps_ex3": [_c_int] * 4 + [_vp, _c_long, _vp, _vp, _c_int, _c_int, _vp, _c_long, _c_int, _vp],
    "sa_coop_error_state": [_c_int],
    "sa_debug_fps_coop_orphan": [_c_int] * 4 + [_vp, _vp, _vp, ctypes.c_uint, _vp],
    "sa_fps_bucket_ex2": [_c_int] * 3 + [_vp, _c_long, _vp, _c_int, _c_int, _vp, _c_long, _vp],
    "sa_fps_with_distance_ex2": [_c_int] * 3 + [_vp, _vp, _vp, _c_int, _c_int, _vp, _c_long, _vp, _c_long, _vp],
    "sa_calc_square_dist_self_ws": [_c_int] * 4 + [_vp, _c_int, _vp, _c_int, _vp, _vp, _vp],
    "sa_fps_bucket_ex": [_c_int] * 3 + [_vp, _vp, _c_int, _c_int, _vp],
    "sa_ffps_fly_ex": [_c_int] * 4 + [_vp, _c_long, _vp, _c_long, _vp, _vp, _c_int, _c_int, _vp, _c_long, _vp],
    "sa_fps_bucket_stats": [_c_int] * 3 + [_vp, _vp, _vp, _vp],
    "sa_fps_generic": [_c_int] * 4 + [_vp, _vp, _vp, _c_int, _vp],
    "sa_calc_square_dist_split": [_c_int] * 5 + [_vp, _vp, _vp, _vp, _vp, _vp],
    "sa_calc_square_dist_split_ws": [_c_int] * 5 + [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sa_query_ball_point_multi": [_c_int] * 4 + [_vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp],
    "sa_query_ball_point_grid": [_c_int] * 4 + [_vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "sa_query_ball_point_grid_ex": [_c_int] * 4 + [_vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _c_int, _vp],
    "sa_group_mlp_max": [_c_int] * 5 + [_vp] * 5 + [_c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _vp,
                         ctypes.c_size_t, _c_int, _vp, _vp],
    "sa_group_mlp_plan": [_c_int] * 3 + [_vp, _vp, _vp, _vp, _c_int, _vp, _vp, _c_int, _vp],
    "sa_group_mlp_plan2": [_c_int] * 3 + [_vp, _vp, _vp, _vp, _c_int, _vp, _vp, _c_int, _vp, _vp],
    "sa_dense": [_c_long, _c_int, _c_int, _vp, _vp, _vp, _c_int, _vp, _vp],
    "sa_decode_anchor_free": [_c_int] * 4 + [_vp] * 7,
    "sa_boxes_to_bev": [_c_long, _vp, _vp, _vp],
    "sa_nms_bev": [_c_int] * 4 + [_c_float, _vp, _vp, _vp, _vp, _vp],
    "sa_nms_gather": [_c_int] * 5 + [_vp] * 7,
    "sa_vote_translate": [_c_long, _vp, _vp, _c_float, _c_float, _c_float, _vp, _vp],
    "sa_vote_tail": [_c_long, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_float, _c_float, _c_float, _vp, _vp],
    "sa_three_nn": [_c_int] * 3 + [_vp, _vp, _vp, _vp, _vp],
    "sa_three_interpolate": [_c_int] * 4 + [_vp, _vp, _vp, _vp, _vp],
    "sa_k_interpolate": [_c_int] * 5 + [_vp, _vp, _vp, _vp, _vp],
    "sa_query_boxes_3d_points": [_c_int] * 4 + [_vp, _vp, _vp, _vp, _vp],
    "sa_query_boxes_3d_mask": [_c_int] * 3 + [_vp, _vp, _vp, _vp],
    "sa_query_points_iou": [_c_int] * 4 + [_vp, _vp, _vp, _vp, _vp, _vp],
    "sa_gather_point_grad": [_c_int] * 4 + [_vp, _vp, _vp, _vp],
    "sa_group_point_grad": [_c_int] * 5 + [_vp, _vp, _vp, _vp],
    "sa_gather_by_mask": [_c_int] * 4 + [_vp, _vp, _vp, _vp, _vp],
    "sa_query_ball_point_withidx": [_c_int] * 3 + [_c_float, _c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "sa_selection_sort": [_c_int] * 4 + [_vp, _vp, _vp, _vp],
    "sa_pairwise_sqdist": [_c_int] * 4 + [_vp, _vp, _vp, _vp],
    "sa_farthest_point_sample_with_preidx": [_c_int] * 5 + [_vp, _vp, _vp, _vp, _vp],
    "sa_three_interpolate_grad": [_c_int] * 4 + [_vp, _vp, _vp, _vp, _vp],
    "sa_k_interpolate_grad": [_c_int] * 5 + [_vp, _vp, _vp, _vp, _vp],
}

# lib3dssd_extra.so (include/sa_extra.h): the reference's operators outside the set-abstraction path, frozen
EXTRA_LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "lib3dssd_extra.so")
EXTRA_SIGNATURES = {
    "sa_points_pooling": [_c_int] * 8 + [_vp] * 8,
    "sa_points_pooling_grad": [_c_int] * 8 + [_vp] * 5,
    "sa_calc_iou": [_c_int] * 3 + [_vp] * 5,
    "sa_calc_iou_match": [_c_int] + [_vp] * 5,
}

_ERRORS = {-1: "invalid argument", -2: "kernel launch failed (hipGetLastError)", -3: "unsupported size",
           -4: "an earlier multi-workgroup sampler launch gave up waiting for partner workgroups (its outputs are invalid; "
               "such launches must stay on one stream -- sa_coop_error_state(1) clears the sticky word)"}
_LIB = None
_EXTRA = None
_COOP_WORD_ASKED = [False]


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load lib3dssd_sa.so once; raise if it (or any declared symbol) is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                "HIP extension not built: %s is missing. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # torch first: it brings its own libamdhip64, and the kernels must register with the HIP runtime whose streams they
        # are launched on.  Loaded before torch, this library would pull in /opt/rocm's copy -- two runtimes in one
        # process, every launch then fails (seen with build() and smoke() in one interpreter).
        import torch  # noqa: F401
        h = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _c_int
        h.sa_query_ball_point_grid_ws_bytes.argtypes = [_c_int, _c_int, _c_int]     # the one non-status function
        h.sa_query_ball_point_grid_ws_bytes.restype = ctypes.c_size_t
        h.sa_group_mlp_max_ws_bytes.argtypes = [_c_int] * 3
        h.sa_group_mlp_max_ws_bytes.restype = ctypes.c_size_t
        h.sa_group_mlp_gemm_ws_bytes.argtypes = [_c_int] * 5 + [_vp]
        h.sa_group_mlp_gemm_ws_bytes.restype = ctypes.c_size_t
        h.sa_calc_square_dist_ws_bytes.argtypes = [_c_int] * 5
        h.sa_calc_square_dist_ws_bytes.restype = ctypes.c_size_t
        h.sa_group_mlp_granule_rows.argtypes = [_c_int] * 6 + [_vp, _vp, ctypes.c_size_t, _c_int]   # returns 4 or 8
        h.sa_group_mlp_granule_rows.restype = _c_int
        h.sa_ffps_fly_ws_bytes.argtypes = [_c_int] * 2
        h.sa_ffps_fly_ws_bytes.restype = ctypes.c_size_t
        h.sa_host_crc32c.argtypes = [_vp, ctypes.c_size_t, ctypes.c_uint32]          # host helper: returns the CRC
        h.sa_host_crc32c.restype = ctypes.c_uint32
        _LIB = h
    if not _COOP_WORD_ASKED[0]:
        # the sticky error word of the multi-workgroup samplers is pinned host memory: ask for it NOW, outside any stream
        # capture (hipHostMalloc is refused inside a global-mode capture -- ADVICE r5), as soon as a GPU is there
        try:
            import torch
            if not torch.cuda.is_available():
                _COOP_WORD_ASKED[0] = True                  # no device: nothing to allocate, do not ask again
            elif not torch.cuda.is_current_stream_capturing():
                _COOP_WORD_ASKED[0] = True
                _LIB.sa_coop_error_state(0)
        except Exception:  # noqa: BLE001 -- the library itself retries on the first sampler call
            _COOP_WORD_ASKED[0] = True
    return _LIB


def lib_extra():
    """Load lib3dssd_extra.so (`make extra`): points pooling and the evaluation IoU, not part of the hot path."""
    global _EXTRA
    if _EXTRA is None:
        if not os.path.exists(EXTRA_LIB_PATH):
            raise NativeLibraryError("HIP extension not built: %s is missing (make -C 3dssd_amd/csrc extra). "
                                     "There is no CPU fallback." % EXTRA_LIB_PATH)
        import torch  # noqa: F401  (same reason as in lib())
        h = ctypes.CDLL(EXTRA_LIB_PATH)
        for name, argtypes in EXTRA_SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = argtypes
            fn.restype = _c_int
        _EXTRA = h
    return _EXTRA


def check(status, what):
    if status != 0:
        msg = "%s: %s (status %d)" % (what, _ERRORS.get(status, "error"), status)
        if status == -1:
            raise ValueError(msg)
        raise RuntimeError(msg)


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def copy_blocks(jobs, stream=None):
    """One launch for up to four strided block copies; jobs = [(src, dst, frames, rows, cols), ...] with src / dst
    3-D fp32 views [frames, rows, >= cols] whose last dimension is dense (any frame / row strides).  stream: a raw HIP
    stream handle (default: torch's current stream)."""
    flat = []
    for src, dst, frames, rows, cols in jobs:
        assert src.stride(2) == 1 and dst.stride(2) == 1
        flat += [src.data_ptr(), dst.data_ptr(), frames, rows, cols, src.stride(0), src.stride(1), dst.stride(0), dst.stride(1)]
    arr = (ctypes.c_long * len(flat))(*flat)
    check(lib().sa_copy_blocks(len(jobs), arr, current_stream() if stream is None else stream), "copy_blocks")


def mlp_plan_ws(b, m, ns, device, c=None, dims=None):
    """Device scratch of one sa_group_mlp_max call: (int32 tensor, size in bytes).  The row plan; with c and dims =
    [c + 3, widths...] given, also the packed hidden activations of the GEMM chain where the scale is eligible for it
    (sa_group_mlp_gemm_ws_bytes)."""
    import torch
    if dims is not None:
        arr = (ctypes.c_int * len(dims))(*[int(d) for d in dims])
        nbytes = int(lib().sa_group_mlp_gemm_ws_bytes(int(b), int(m), int(ns), int(c), len(dims) - 1, arr))
    else:
        nbytes = int(lib().sa_group_mlp_max_ws_bytes(int(b), int(m), int(ns)))
    return torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=device), nbytes
