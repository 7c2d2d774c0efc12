"""Constants of the detection task, with the attribute names the reference's
tools read (upstream:coperception/configs/Config.py; values per SURVEY.md
Appx A.1).  Only what the `--com disco` forward path consumes is kept."""
import math

import numpy as np


class Config:
    def __init__(self, split="train", binary=True, only_det=True, code_type="faf",
                 loss_type="faf_loss", savepath="", root="", is_cross_road=False, use_vis=False,
                 map_hw=256):
        self.split = split
        self.binary = binary
        self.only_det = only_det
        self.code_type = code_type
        self.loss_type = loss_type
        self.use_vis = use_vis
        self.use_map = False
        self.motion_state = False
        self.pred_len = 1
        self.box_code_size = 6
        self.category_num = 2

        self.voxel_size = (0.25, 0.25, 0.4)
        half = map_hw * self.voxel_size[0] / 2.0
        self.area_extents = np.array([[-half, half], [-half, half], [-3.0, 2.0]])
        self.map_dims = [
            int((self.area_extents[0][1] - self.area_extents[0][0]) / self.voxel_size[0]),
            int((self.area_extents[1][1] - self.area_extents[1][0]) / self.voxel_size[1]),
            int(math.ceil((self.area_extents[2][1] - self.area_extents[2][0]) / self.voxel_size[2])),
        ]
        # (w, l, yaw) per anchor; only len() is consumed by the forward path
        self.anchor_size = np.asarray(
            [[2.0, 4.0, 0.0], [2.0, 4.0, math.pi / 2.0], [2.0, 4.0, -math.pi / 4.0],
             [3.0, 12.0, 0.0], [3.0, 12.0, math.pi / 2.0], [3.0, 12.0, -math.pi / 4.0]])
