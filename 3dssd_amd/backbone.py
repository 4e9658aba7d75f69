"""The SA backbone of the single-stage detector: the loop of
lib/modeling/single_stage_detector.py:115-125 (network_forward) over the ARCHITECTURE rows, without
the detection head (out of scope)."""
import torch

from .builder.layer_builder import LayerBuilder
from .utils import layers_util
from .utils import _native as N
from .utils.weights import VariableStore


class SABackbone:
    def __init__(self, arch, params, device="cuda:0", max_translate_range=(-3.0, -2.0, -3.0),
                 aggregation_sa_feature=True):
        self.device = torch.device(device)
        self.variables = VariableStore(params, self.device)
        layers_util.AGGREGATION_SA_FEATURE = bool(aggregation_sa_feature)
        layers_util.MAX_TRANSLATE_RANGE = tuple(max_translate_range)
        self.layers = [LayerBuilder(i, False, arch, variables=self.variables) for i in range(len(arch))]

    def split_input(self, point_cloud):
        """The two tf.slice of single_stage_detector.py:117-118 in one launch: [B,n,3+C] -> xyz [B,n,3], features [B,n,C]."""
        point_cloud = point_cloud.contiguous()
        bs, n, ch = point_cloud.shape
        l0_xyz = torch.empty((bs, n, 3), dtype=torch.float32, device=point_cloud.device)
        l0_points = torch.empty((bs, n, ch - 3), dtype=torch.float32, device=point_cloud.device)
        N.copy_blocks([(point_cloud[:, :, 0:3], l0_xyz, bs, n, 3), (point_cloud[:, :, 3:], l0_points, bs, n, ch - 3)])
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
