"""The SA backbone of the single-stage detector: the loop of
lib/modeling/single_stage_detector.py:115-125 (network_forward) over the ARCHITECTURE rows, without
the detection head (out of scope)."""
import torch

from .builder.layer_builder import LayerBuilder
from .utils import _native as N
from .utils.tf_ops import _tensor as T
from .utils.weights import VariableStore


class SABackbone:
    def __init__(self, arch, params, device="cuda:0", max_translate_range=(-3.0, -2.0, -3.0),
                 aggregation_sa_feature=True, precision=None, dfps_side_stream=None, ffps_fly=None, coop_capture=False):
        """precision: None = per-scale rule of utils/weights.py (fp16 one-pass on the wide scales, split bf16 elsewhere),
        "bf16x3" = split bf16 everywhere (~1e-5 of fp32, no range limit), "fp16" = one pass wherever the weights fit.
        dfps_side_stream: where the F-FPS || D-FPS launch of an 'FS' layer is issued (layers_util.DFPS_SIDE_STREAM; None =
        its default 6, a helper-stream branch; 5 = on the issuing stream, which keeps a captured graph linear).
        ffps_fly: F-FPS without the distance matrix where the shape allows it (layers_util.FFPS_FLY; csrc/ffps_fly.hip --
        all calls of a process on one stream at a time: direct single-stream callers only).
        coop_capture: frames of more than 16384 points may launch their multi-workgroup layer-1 sampler plainly under
        hipGraph capture (sa_fps_ex3 flag bit 0) -- only for a caller that keeps those launches on ONE stream (the staged
        executor sets it for its own network); default: a capture takes the single-workgroup kernels."""
        self.device = torch.device(device)
        self.variables = params if isinstance(params, VariableStore) else VariableStore(params, self.device, precision)
        # per instance (round 3 wrote them into layers_util's module attributes: two backbones shared the last value)
        self.settings = {"aggregation_sa_feature": bool(aggregation_sa_feature),
                         "max_translate_range": tuple(float(v) for v in max_translate_range),
                         "dfps_side_stream": dfps_side_stream, "ffps_fly": ffps_fly, "coop_capture": bool(coop_capture)}
        self.layers = [LayerBuilder(i, False, arch, variables=self.variables, settings=self.settings)
                       for i in range(len(arch))]

    def split_input(self, point_cloud):
        """The two tf.slice of single_stage_detector.py:117-118 in one launch: [B,n,3+C] -> xyz [B,n,3], features [B,n,C]."""
        point_cloud = T.f32_cuda(point_cloud, "point_cloud")      # device / dtype checks, fp32, contiguous
        T.require(point_cloud.dim() == 3 and point_cloud.shape[2] >= 3, "point_cloud must be [B, n, 3 + C]")
        bs, n, ch = point_cloud.shape
        l0_xyz = torch.empty((bs, n, 3), dtype=torch.float32, device=point_cloud.device)
        l0_points = torch.empty((bs, n, ch - 3), dtype=torch.float32, device=point_cloud.device)
        jobs = [(point_cloud[:, :, 0:3], l0_xyz, bs, n, 3)]
        if ch > 3:                                                # an xyz-only cloud has no feature block to copy
            jobs.append((point_cloud[:, :, 3:], l0_points, bs, n, ch - 3))
        N.copy_blocks(jobs)
        return l0_xyz, l0_points

    def forward(self, point_cloud):
        """point_cloud [B,n,3+C] fp32 on the GPU -> (xyz_list, feature_list, fps_idx_list); the backbone
        output is the last entry of xyz_list / feature_list."""
        l0_xyz, l0_points = self.split_input(point_cloud)
        xyz_list, feature_list, fps_idx_list = [l0_xyz], [l0_points], [None]
        out = {}
        for layer in self.layers:
            xyz_list, feature_list, fps_idx_list = layer.build_layer(xyz_list, feature_list, fps_idx_list,
                                                                     None, out)
        return xyz_list, feature_list, fps_idx_list

    __call__ = forward

    def forward_staged(self, point_cloud, split_layer2_sampler=False):
        """forward() as a generator with ONE yield, between the sampling half of the first row (input split + layer-1
        D-FPS + centres: a latency-bound chain on one CU per frame) and everything else: the staged executor
        (pipeline.py) enqueues / captures the two halves on different streams.  The generator's return value
        (StopIteration.value) is forward()'s result; the kernels and their order are exactly forward()'s.
        split_layer2_sampler: THREE yields -- also around the sampling half of the second row (the F-FPS || D-FPS of
        layer 2), so that an executor can keep every launch of the on-the-fly F-FPS (ffps_fly: multi-workgroup, its
        launches must never overlap) on one dedicated stream."""
        l0_xyz, l0_points = self.split_input(point_cloud)
        xyz_list, feature_list, fps_idx_list = [l0_xyz], [l0_points], [None]
        pre = self.layers[0].sample(xyz_list, feature_list, fps_idx_list)
        yield
        out = {}
        first = 0
        if split_layer2_sampler and len(self.layers) > 1:
            xyz_list, feature_list, fps_idx_list = self.layers[0].build_layer(xyz_list, feature_list, fps_idx_list, None, out,
                                                                              presampled=pre)
            yield
            pre = self.layers[1].sample(xyz_list, feature_list, fps_idx_list)
            yield
            first = 1
        for i in range(first, len(self.layers)):
            xyz_list, feature_list, fps_idx_list = self.layers[i].build_layer(xyz_list, feature_list, fps_idx_list, None, out,
                                                                              presampled=pre if i == first else None)
        return xyz_list, feature_list, fps_idx_list

    def raise_if_overflow(self):
        """The fp16 scales guard their operand range (csrc/mlp_act.h); forward() does not synchronise, so the check is
        explicit: call it after the results are complete (SAPipeline tickets do)."""
        self.variables.raise_if_overflow("SA backbone")
