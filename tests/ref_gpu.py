"""ctypes binding of oracle/_ref/libtf_ops_ref_*.so -- the REFERENCE'S OWN device code (tf_sampling_g.cu,
tf_grouping_g.cu, tf_interpolate_g.cu, tf_points_pooling_g.cu), compiled unmodified for gfx950 by
`make -C oracle ref_gpu`.  TEST INFRASTRUCTURE ONLY: nothing under 3dssd_amd/ imports this.

The launchers are the C++ functions the reference's OpKernels declare (tf_sampling.cpp:131,164,235,
tf_grouping.cpp:270,363,446, ...); they are bound by their Itanium-mangled names, take raw device pointers,
launch on the default stream and check nothing -- so every call here is bracketed by a device synchronise.
Output buffers are zero-filled first: rows the reference leaves unwritten (empty balls, tf_grouping_g.cu:236-253)
then read as zeros, the oracle's convention D.

Function names / positional order follow the reference's Python operator API (scalars first, tensors last).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(os.path.dirname(_HERE), "oracle", "_ref")
VARIANTS = ("fma", "nofma", "hipdefault")

_i, _f, _p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
# launcher -> (mangled symbol, argtypes)
_SYMS = {
    "farthestpointsamplingLauncher": ("_Z29farthestpointsamplingLauncheriiiiPKfPfPi", [_i] * 4 + [_p] * 3),
    "farthestpointsamplingwithdistLauncher": ("_Z37farthestpointsamplingwithdistLauncheriiiPKfPfPi", [_i] * 3 + [_p] * 3),
    "farthestpointsamplingwithpreidxLauncher": ("_Z39farthestpointsamplingwithpreidxLauncheriiiiiPKfPKiPfPi",
                                                [_i] * 5 + [_p] * 4),
    "gatherpointLauncher": ("_Z19gatherpointLauncheriiiiPKfPKiPf", [_i] * 4 + [_p] * 3),
    "scatteraddpointLauncher": ("_Z23scatteraddpointLauncheriiiiPKfPKiPf", [_i] * 4 + [_p] * 3),
    "GatherByMaskLauncher": ("_Z20GatherByMaskLauncheriiiiPKfS0_Pf", [_i] * 4 + [_p] * 3),
    "queryBallPointLauncher": ("_Z22queryBallPointLauncheriiifiPKfS0_PiS1_", [_i] * 3 + [_f, _i] + [_p] * 4),
    "queryBallPointDilatedLauncher": ("_Z29queryBallPointDilatedLauncheriiiffiPKfS0_PiS1_",
                                      [_i] * 3 + [_f, _f, _i] + [_p] * 4),
    "queryBallPointWithidxLauncher": ("_Z29queryBallPointWithidxLauncheriiifiPKfS0_PKiPiS3_",
                                      [_i] * 3 + [_f, _i] + [_p] * 5),
    "groupPointLauncher": ("_Z18groupPointLauncheriiiiiPKfPKiPf", [_i] * 5 + [_p] * 3),
    "groupPointGradLauncher": ("_Z22groupPointGradLauncheriiiiiPKfPKiPf", [_i] * 5 + [_p] * 3),
    "selectionSortLauncher": ("_Z21selectionSortLauncheriiiiPKfPiPf", [_i] * 4 + [_p] * 3),
    "queryBoxes3dPointsLauncher": ("_Z26queryBoxes3dPointsLauncheriiiiPKfS0_PiS1_", [_i] * 4 + [_p] * 4),
    "queryBoxes3dMaskLauncher": ("_Z24queryBoxes3dMaskLauncheriiiPKfS0_Pi", [_i] * 3 + [_p] * 3),
    "queryPointsIouLauncher": ("_Z22queryPointsIouLauncheriiiiPKfS0_S0_S0_Pf", [_i] * 4 + [_p] * 5),
    "ThreeNNLauncher": ("_Z15ThreeNNLauncheriiiPKfS0_PfPi", [_i] * 3 + [_p] * 4),
    "ThreeInterpolateLauncher": ("_Z24ThreeInterpolateLauncheriiiiPKfPKiS0_Pf", [_i] * 4 + [_p] * 4),
    "ThreeInterpolateGradLauncher": ("_Z28ThreeInterpolateGradLauncheriiiiPKfPKiS0_Pf", [_i] * 4 + [_p] * 4),
    "KInterpolateLauncher": ("_Z20KInterpolateLauncheriiiiiPKfPKiS0_Pf", [_i] * 5 + [_p] * 4),
    "KInterpolateGradLauncher": ("_Z24KInterpolateGradLauncheriiiiiPKfPKiS0_Pf", [_i] * 5 + [_p] * 4),
    "pointsPoolingLauncher": ("_Z21pointsPoolingLauncheriiiiiiiiPKfS0_S0_PfPiS2_S1_", [_i] * 8 + [_p] * 7),
    "pointsPoolingGradLauncher": ("_Z25pointsPoolingGradLauncheriiiiiiiiPKfPKiS2_S0_Pf", [_i] * 8 + [_p] * 5),
}


def so_path(variant="fma"):
    return os.path.join(REF_DIR, "libtf_ops_ref_%s.so" % variant)


def available(variant="fma"):
    return os.path.exists(so_path(variant))


class RefOps:
    """One build of the reference's device code.  Tensors in and out are torch CUDA tensors."""

    def __init__(self, variant="fma"):
        assert variant in VARIANTS
        self.variant = variant
        self._h = ctypes.CDLL(so_path(variant))
        for name, (sym, argtypes) in _SYMS.items():
            fn = getattr(self._h, sym)
            fn.argtypes = argtypes
            fn.restype = None
            setattr(self, "_" + name, fn)

    # -- plumbing
    @staticmethod
    def _f32(t):
        assert t.is_cuda
        return t.to(torch.float32).contiguous()

    @staticmethod
    def _i32(t):
        assert t.is_cuda
        return t.to(torch.int32).contiguous()

    def _run(self, name, *args):
        torch.cuda.synchronize()
        getattr(self, "_" + name)(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args])
        torch.cuda.synchronize()   # surfaces a fault of the unchecked launch as an exception

    # -- sampling (tf_sampling.py)
    def farthest_point_sample(self, npoint, inp):
        inp = self._f32(inp)
        b, n, c = inp.shape
        temp = torch.zeros((b, n), dtype=torch.float32, device=inp.device)       # allocate_temp, tf_sampling.cpp:149-155
        out = torch.zeros((b, npoint), dtype=torch.int32, device=inp.device)
        self._run("farthestpointsamplingLauncher", b, n, c, npoint, inp, temp, out)
        return out

    def farthest_point_sample_with_distance(self, npoint, dist):
        dist = self._f32(dist)
        b, n, _ = dist.shape
        temp = torch.zeros((b, n), dtype=torch.float32, device=dist.device)
        out = torch.zeros((b, npoint), dtype=torch.int32, device=dist.device)
        self._run("farthestpointsamplingwithdistLauncher", b, n, npoint, dist, temp, out)
        return out

    def farthest_point_sample_with_preidx(self, npoint, inp, preidx):
        inp, preidx = self._f32(inp), self._i32(preidx)
        b, n, c = inp.shape
        m1 = preidx.shape[1]
        temp = torch.zeros((b, n), dtype=torch.float32, device=inp.device)
        out = torch.zeros((b, npoint), dtype=torch.int32, device=inp.device)
        self._run("farthestpointsamplingwithpreidxLauncher", b, n, c, npoint, m1, inp, preidx, temp, out)
        return out

    def gather_point(self, inp, idx):
        inp, idx = self._f32(inp), self._i32(idx)
        b, n, c = inp.shape
        m = idx.shape[1]
        out = torch.zeros((b, m, c), dtype=torch.float32, device=inp.device)
        self._run("gatherpointLauncher", b, n, m, c, inp, idx, out)
        return out

    def gather_point_grad(self, inp, idx, out_g):
        inp, idx, out_g = self._f32(inp), self._i32(idx), self._f32(out_g)
        b, n, c = inp.shape
        m = idx.shape[1]
        inp_g = torch.zeros((b, n, c), dtype=torch.float32, device=inp.device)   # cudaMemset, tf_sampling.cpp:286
        self._run("scatteraddpointLauncher", b, n, m, c, out_g, idx, inp_g)
        return inp_g

    def gather_by_mask(self, proposal_num, inp, mask):
        inp, mask = self._f32(inp), self._f32(mask)
        b, n, c = inp.shape
        out = torch.zeros((b, proposal_num, c), dtype=torch.float32, device=inp.device)
        self._run("GatherByMaskLauncher", b, n, c, proposal_num, inp, mask, out)
        return out

    # -- grouping (tf_grouping.py)
    def query_ball_point(self, radius, nsample, xyz1, xyz2):
        xyz1, xyz2 = self._f32(xyz1), self._f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz1.device)
        cnt = torch.zeros((b, m), dtype=torch.int32, device=xyz1.device)
        self._run("queryBallPointLauncher", b, n, m, float(radius), nsample, xyz1, xyz2, idx, cnt)
        return idx, cnt

    def query_ball_point_dilated(self, min_radius, max_radius, nsample, xyz1, xyz2):
        xyz1, xyz2 = self._f32(xyz1), self._f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz1.device)
        cnt = torch.zeros((b, m), dtype=torch.int32, device=xyz1.device)
        self._run("queryBallPointDilatedLauncher", b, n, m, float(min_radius), float(max_radius), nsample,
                  xyz1, xyz2, idx, cnt)
        return idx, cnt

    def query_ball_point_withidx(self, radius, nsample, xyz1, xyz2, sort_idx):
        xyz1, xyz2, sort_idx = self._f32(xyz1), self._f32(xyz2), self._i32(sort_idx)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz1.device)
        cnt = torch.zeros((b, m), dtype=torch.int32, device=xyz1.device)
        self._run("queryBallPointWithidxLauncher", b, n, m, float(radius), nsample, xyz1, xyz2, sort_idx, idx, cnt)
        return idx, cnt

    def group_point(self, points, idx):
        points, idx = self._f32(points), self._i32(idx)
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = torch.zeros((b, m, ns, c), dtype=torch.float32, device=points.device)
        self._run("groupPointLauncher", b, n, c, m, ns, points, idx, out)
        return out

    def group_point_grad(self, points, idx, grad_out):
        points, idx, grad_out = self._f32(points), self._i32(idx), self._f32(grad_out)
        b, n, c = points.shape
        _, m, ns = idx.shape
        g = torch.zeros((b, n, c), dtype=torch.float32, device=points.device)   # cudaMemset, tf_grouping.cpp:510
        self._run("groupPointGradLauncher", b, n, c, m, ns, grad_out, idx, g)
        return g

    def select_top_k(self, k, dist):
        dist = self._f32(dist)
        b, m, n = dist.shape
        outi = torch.zeros((b, m, n), dtype=torch.int32, device=dist.device)
        out = torch.zeros((b, m, n), dtype=torch.float32, device=dist.device)
        self._run("selectionSortLauncher", b, n, m, k, dist, outi, out)
        return outi, out

    def query_boxes_3d_points(self, nsample, xyz, proposals):
        xyz, proposals = self._f32(xyz), self._f32(proposals)
        b, n, _ = xyz.shape
        m = proposals.shape[1]
        idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz.device)
        cnt = torch.zeros((b, m), dtype=torch.int32, device=xyz.device)
        self._run("queryBoxes3dPointsLauncher", b, n, m, nsample, xyz, proposals, idx, cnt)
        return idx, cnt

    def query_boxes_3d_mask(self, xyz, boxes_3d):
        xyz, boxes_3d = self._f32(xyz), self._f32(boxes_3d)
        b, n, _ = xyz.shape
        m = boxes_3d.shape[1]
        mask = torch.zeros((b, m, n), dtype=torch.int32, device=xyz.device)
        self._run("queryBoxes3dMaskLauncher", b, n, m, xyz, boxes_3d, mask)
        return mask

    def query_points_iou(self, xyz, anchors_3d, gt_boxes_3d, iou_matrix):
        xyz, anchors_3d = self._f32(xyz), self._f32(anchors_3d)
        gt_boxes_3d, iou_matrix = self._f32(gt_boxes_3d), self._f32(iou_matrix)
        b, n, _ = xyz.shape
        a, g = anchors_3d.shape[1], gt_boxes_3d.shape[1]
        out = torch.zeros((b, a, g), dtype=torch.float32, device=xyz.device)
        self._run("queryPointsIouLauncher", b, n, a, g, xyz, anchors_3d, gt_boxes_3d, iou_matrix, out)
        return out

    # -- interpolation (tf_interpolate.py)
    def three_nn(self, xyz1, xyz2):
        xyz1, xyz2 = self._f32(xyz1), self._f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist = torch.zeros((b, n, 3), dtype=torch.float32, device=xyz1.device)
        idx = torch.zeros((b, n, 3), dtype=torch.int32, device=xyz1.device)
        self._run("ThreeNNLauncher", b, n, m, xyz1, xyz2, dist, idx)
        return dist, idx

    def three_interpolate(self, points, idx, weight):
        points, idx, weight = self._f32(points), self._i32(idx), self._f32(weight)
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.zeros((b, n, c), dtype=torch.float32, device=points.device)
        self._run("ThreeInterpolateLauncher", b, m, c, n, points, idx, weight, out)
        return out

    def k_interpolate(self, points, idx, weight):
        points, idx, weight = self._f32(points), self._i32(idx), self._f32(weight)
        b, m, c = points.shape
        _, n, k = idx.shape
        out = torch.zeros((b, n, c), dtype=torch.float32, device=points.device)
        self._run("KInterpolateLauncher", b, m, c, n, k, points, idx, weight, out)
        return out
