"""MeanVFE (pcdet/models/backbones_3d/vfe/mean_vfe.py:6-31): same constructor, attributes and
batch_dict contract.  On this path the voxeliser already produces the mean row
(csrc/voxelize.hip fuses it), so when `batch_dict` carries 'voxel_features' from
VoxelResBackBone8x.voxelize_batch this module is a no-op; given the reference's
(voxels, voxel_num_points) it evaluates the same expression with torch ops on the device.
"""
import torch
import torch.nn as nn


class MeanVFE(nn.Module):
    def __init__(self, model_cfg=None, num_point_features=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        if "voxels" not in batch_dict and "voxel_features" in batch_dict:
            return batch_dict  # produced by the fused GPU voxeliser
        voxel_features, voxel_num_points = batch_dict["voxels"], batch_dict["voxel_num_points"]
        points_mean = voxel_features[:, :, :].sum(dim=1, keepdim=False)
        normalizer = torch.clamp_min(voxel_num_points.view(-1, 1), min=1.0).type_as(voxel_features)
        points_mean = points_mean / normalizer
        batch_dict["voxel_features"] = points_mean.contiguous()
        return batch_dict
