"""LayerBuilder with the reference's interface (lib/builder/layer_builder.py:11-102): decodes one
ARCHITECTURE row (lib/core/config.py:207-219) and calls the SA / Vote layer.  FP layers and
SA_Layer_SSG_Last belong to PointRCNN and are outside the 3DSSD SA path."""
from ..utils.layers_util import pointnet_sa_module_msg, sample_layer, vote_layer


class LayerBuilder:
    def __init__(self, layer_idx, is_training, layer_cfg, variables=None, settings=None):
        """settings: {"aggregation_sa_feature": bool, "max_translate_range": (x, y, z)} -- the two entries of the
        reference's global cfg the layers read (3dssd.yaml:39,44), per builder instead of per process; None: the
        defaults of utils/layers_util.py."""
        self.layer_idx = layer_idx
        self.settings = dict(settings or {})
        self.is_training = is_training
        self.variables = variables
        a = layer_cfg[self.layer_idx]
        self.layer_architecture = a
        self.xyz_index, self.feature_index = a[0], a[1]
        self.radius_list, self.nsample_list, self.mlp_list, self.bn = a[2], a[3], a[4], a[5]
        self.fps_sample_range_list, self.fps_method_list, self.npoint_list = a[6], a[7], a[8]
        assert len(self.fps_sample_range_list) == len(self.fps_method_list)
        assert len(self.fps_method_list) == len(self.npoint_list)
        self.former_fps_idx, self.use_attention = a[9], a[10]
        self.layer_type, self.scope = a[11], a[12]
        self.dilated_group, self.vote_ctr_index, self.aggregation_channel = a[13], a[14], a[15]
        if self.layer_type in ("SA_Layer", "Vote_Layer"):
            assert len(self.xyz_index) == 1
        else:
            raise Exception("Not Implementation Error!!!")

    def sample(self, xyz_list, feature_list, fps_idx_list):
        """The sampling half of an SA layer alone (layers_util.py:84-119) -> the `presampled` argument of build_layer.
        None for layers without a sampling half of their own (vote layers)."""
        if self.layer_type != "SA_Layer":
            return None
        former_fps_idx = fps_idx_list[self.former_fps_idx] if self.former_fps_idx != -1 else None
        vote_ctr = xyz_list[self.vote_ctr_index] if self.vote_ctr_index != -1 else None
        return sample_layer(xyz_list[self.xyz_index[0]], feature_list[self.feature_index[0]], self.fps_sample_range_list,
                            self.fps_method_list, self.npoint_list, former_fps_idx, vote_ctr, self.radius_list,
                            self.settings.get("dfps_side_stream"), self.settings.get("ffps_fly"),
                            bool(self.settings.get("coop_capture")))

    def build_layer(self, xyz_list, feature_list, fps_idx_list, bn_decay=None, output_dict=None, presampled=None):
        xyz_input = [xyz_list[i] for i in self.xyz_index]
        feature_input = [feature_list[i] for i in self.feature_index]
        former_fps_idx = fps_idx_list[self.former_fps_idx] if self.former_fps_idx != -1 else None
        vote_ctr = xyz_list[self.vote_ctr_index] if self.vote_ctr_index != -1 else None
        if self.layer_type == "SA_Layer":
            new_xyz, new_points, new_fps_idx = pointnet_sa_module_msg(
                xyz_input[0], feature_input[0], self.radius_list, self.nsample_list, self.mlp_list,
                self.is_training, bn_decay, self.bn, self.fps_sample_range_list, self.fps_method_list,
                self.npoint_list, former_fps_idx, bool(self.use_attention) and self.use_attention != -1,
                self.scope, self.dilated_group, vote_ctr, self.aggregation_channel,
                variables=self.variables, aggregation_sa_feature=self.settings.get("aggregation_sa_feature"),
                presampled=presampled, dfps_side_stream=self.settings.get("dfps_side_stream"),
                ffps_fly=self.settings.get("ffps_fly"), coop_capture=bool(self.settings.get("coop_capture")))
            xyz_list.append(new_xyz)
            feature_list.append(new_points)
            fps_idx_list.append(new_fps_idx)
        elif self.layer_type == "Vote_Layer":
            new_xyz, new_points, ctr_offsets = vote_layer(xyz_input[0], feature_input[0], self.mlp_list,
                                                          self.is_training, bn_decay, self.bn, self.scope,
                                                          variables=self.variables,
                                                          max_translate_range=self.settings.get("max_translate_range"))
            if output_dict is not None:
                output_dict.setdefault("pred_vote_base", []).append(xyz_input[0])
                output_dict.setdefault("pred_vote_offset", []).append(ctr_offsets)
            xyz_list.append(new_xyz)
            feature_list.append(new_points)
            fps_idx_list.append(None)
        return xyz_list, feature_list, fps_idx_list
