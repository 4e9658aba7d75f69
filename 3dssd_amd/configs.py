"""Architecture rows of the set-abstraction backbone (data only).

KITTI_3DSSD_ARCH restates MODEL.NETWORK.FIRST_STAGE.ARCHITECTURE of the reference's
configs/kitti/3dssd/3dssd.yaml:46-67.  Row format (lib/core/config.py:207-219, decoded by
lib/builder/layer_builder.py:16-37):

  0 xyz_index  1 feature_index  2 radius_list  3 nsample_list  4 mlp_list  5 bn
  6 fps_sample_range_list  7 fps_method_list  8 npoint_list  9 former_fps_idx
  10 use_attention  11 layer_type  12 scope  13 dilated_group  14 vote_ctr_index
  15 aggregation_channel
"""

KITTI_3DSSD_ARCH = [
    [[0], [0], [0.2, 0.4, 0.8], [32, 32, 64], [[16, 16, 32], [16, 16, 32], [32, 32, 64]], True,
     [-1], ["D-FPS"], [4096],
     -1, False, "SA_Layer", "layer1", True, -1, 64],
    [[1], [1], [0.4, 0.8, 1.6], [32, 32, 64], [[64, 64, 128], [64, 64, 128], [64, 96, 128]], True,
     [-1], ["FS"], [512],
     -1, False, "SA_Layer", "layer2", True, -1, 128],
    [[2], [2], [1.6, 3.2, 4.8], [32, 32, 32], [[128, 128, 256], [128, 192, 256], [128, 256, 256]],
     True,
     [512, -1], ["F-FPS", "D-FPS"], [256, 256],
     -1, False, "SA_Layer", "layer3", True, -1, 256],
    [[3], [3], [], [], [], True,
     [256, -1], ["F-FPS", "D-FPS"], [256, 0],
     -1, False, "SA_Layer", "vote", False, -1, 256],
    [[4], [4], -1, -1, [128], True,
     [-1], [-1], [-1],
     -1, -1, "Vote_Layer", "vote", False, -1, -1],
    [[3], [3], [4.8, 6.4], [16, 32], [[256, 256, 512], [256, 512, 1024]], True,
     [-1], ["D-FPS"], [256],
     -1, False, "SA_Layer", "layer4", False, 5, 512],
]

# configs/kitti/3dssd/3dssd.yaml:39 (MODEL.MAX_TRANSLATE_RANGE), :44 (AGGREGATION_SA_FEATURE),
# :35 (POINTS_NUM_FOR_TRAINING), :3 (POINT_CLOUD_RANGE).
KITTI_MAX_TRANSLATE_RANGE = (-3.0, -2.0, -3.0)
KITTI_AGGREGATION_SA_FEATURE = True
KITTI_POINTS_NUM = 16384
KITTI_POINT_CLOUD_RANGE = (-40.0, 40.0, -5.0, 3.0, 0.0, 70.0)
KITTI_INPUT_FEATURE_CHANNELS = 1  # intensity, lib/dataset/dataloader/kitti_dataloader.py:186,225

# BASELINE.json configs[0]: one SA layer on a 4096-point cloud, CPU plumbing/parity case.
CONFIG0_SINGLE_SA = [
    [[0], [0], [0.2], [32], [[16, 16, 32]], True,
     [-1], ["D-FPS"], [512],
     -1, False, "SA_Layer", "layer1", False, -1, -1],
]

# MODEL.NETWORK.FIRST_STAGE.HEAD of 3dssd.yaml:68 and the post-processing constants (:38, :70-71)
KITTI_3DSSD_HEAD = [[6], [6], "conv1d", [128], True, "Det", ""]
KITTI_ANGLE_CLS_NUM = 12
KITTI_MAX_OUTPUT_NUM = 100
KITTI_NMS_THRESH = 0.1
