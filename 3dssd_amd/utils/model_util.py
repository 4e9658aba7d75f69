"""calc_square_dist of the reference's lib/utils/model_util.py:144-160 on torch-ROCm tensors (the only
function of that module on the SA hot path: it builds the F-FPS distance matrix,
lib/utils/layers_util.py:95,103)."""
import torch

from .tf_ops import _tensor as T
from . import _native as N


def calc_square_dist(a, b, norm=True):
    """a: [bs, npoint, c], b: [bs, ndataset, c] -> [bs, npoint, ndataset] = |a|^2 + |b|^2 - 2 a.b
    (channel sums as ascending fmaf chains; see csrc/sqdist.hip).  The default is the reference's (norm=True:
    sqrt / c, model_util.py:144,156-159), which is NOT provided -- the SA path always passes norm=False
    (layers_util.py:95,103) -- so a caller relying on the default is told instead of silently getting the
    unnormalised matrix."""
    T.require(not norm, "calc_square_dist: only norm=False is implemented; pass norm=False explicitly "
                        "(the reference's default norm=True is sqrt(dist)/c, which the SA path never uses)")
    a = T.f32_cuda(a, "a")
    b = T.f32_cuda(b, "b")
    T.require(a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.shape[2] == b.shape[2],
              "calc_square_dist expects a [bs,npoint,c] and b [bs,ndataset,c]")
    bs, n, c = a.shape
    m = b.shape[1]
    out = torch.empty((bs, n, m), dtype=torch.float32, device=a.device)
    lib = N.lib()
    sym = a.data_ptr() == b.data_ptr() and n == m
    ws = torch.empty((lib.sa_calc_square_dist_ws_bytes(bs, n, m, c, 1 if sym else 0) + 3) // 4, dtype=torch.float32, device=a.device)
    st = lib.sa_calc_square_dist_split_ws(bs, n, m, c, 0, a.data_ptr(), None, b.data_ptr(), None, out.data_ptr(),
                                          ws.data_ptr(), N.current_stream())
    N.check(st, "calc_square_dist")
    return out
