"""Drop-in for pcdet/datasets/processor/data_processor.py: the voxel generator boundary (:17-62) and the
`DataProcessor` queue with the processors the two configs of the hot path list
(`transfusion_lidar.yaml:41-58`, `nuscenes_box_seeker_proposals.yaml:40-57`): mask_points_and_boxes_outside_range
(:80-94), shuffle_points (:96-106), transform_points_to_voxels (:255-302) and its placeholder (:228-236).

`VoxelGeneratorWrapper(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
max_num_voxels).generate(points)` returns (voxels (M,P,C) f32, coordinates (M,3) int32 [z,y,x],
num_points (M,) int32) with the reference's sequential first-come semantics.  Like the reference's it is a HOST
generator (fnp_host_voxelize): DataProcessor creates it lazily inside DataLoader workers, where no GPU context may
be created after a fork; pass device="cuda" for the device voxeliser (main process / spawn workers only).
"""
from functools import partial

import numpy as np

from ..spconv.utils import Point2VoxelCPU3d, Point2VoxelGPU3d


class VoxelGeneratorWrapper:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels, device=None):
        self.spconv_ver = 2
        kw = dict(vsize_xyz=vsize_xyz, coors_range_xyz=coors_range_xyz, num_point_features=num_point_features,
                  max_num_points_per_voxel=max_num_points_per_voxel, max_num_voxels=max_num_voxels)
        self._voxel_generator = Point2VoxelCPU3d(**kw) if device is None else Point2VoxelGPU3d(device=device, **kw)

    def generate(self, points):
        tv_voxels, tv_coordinates, tv_num_points = self._voxel_generator.point_to_voxel(points)
        return tv_voxels.numpy(), tv_coordinates.numpy(), tv_num_points.numpy()


def mask_points_by_range(points, limit_range):
    """common_utils.py:78-81: x and y only, both ends inclusive."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) \
        & (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


def _boxes_to_corners_xy(boxes):
    t = np.array([[1, 1], [1, -1], [-1, -1], [-1, 1], [1, 1], [1, -1], [-1, -1], [-1, 1]], boxes.dtype) / 2
    c = boxes[:, None, 3:5] * t[None]
    cs, sn = np.cos(boxes[:, 6]), np.sin(boxes[:, 6])
    x = c[..., 0] * cs[:, None] - c[..., 1] * sn[:, None]
    y = c[..., 0] * sn[:, None] + c[..., 1] * cs[:, None]
    return np.stack([x, y], -1) + boxes[:, None, 0:2]


def mask_boxes_outside_range_numpy(boxes, limit_range, min_num_corners=1, use_center_to_filter=True):
    """box_utils.py:93-114."""
    limit_range = np.asarray(limit_range)
    if boxes.shape[1] > 7:
        boxes = boxes[:, 0:7]
    if use_center_to_filter:
        return ((boxes[:, 0:3] >= limit_range[0:3]) & (boxes[:, 0:3] <= limit_range[3:6])).all(axis=-1)
    corners = _boxes_to_corners_xy(boxes)
    mask = ((corners >= limit_range[0:2]) & (corners <= limit_range[3:5])).all(axis=2)
    return mask.sum(axis=1) >= min_num_corners


def _cfg(config, key, default=None):
    if isinstance(config, dict):
        return config.get(key, default)
    return config.get(key, default) if hasattr(config, "get") else getattr(config, key, default)


class DataProcessor(object):
    """data_processor.py:64-78,405-420: the queue of bound processors, run in config order."""

    def __init__(self, processor_configs, point_cloud_range, training, num_point_features):
        self.point_cloud_range = np.asarray(point_cloud_range)
        self.training = training
        self.num_point_features = num_point_features
        self.mode = 'train' if training else 'test'
        self.grid_size = self.voxel_size = None
        self.data_processor_queue = []
        self.voxel_generator = None
        for cur_cfg in processor_configs:
            name = _cfg(cur_cfg, 'NAME')
            if not hasattr(self, name):
                raise NotImplementedError(f"DataProcessor.{name} is not on the hot path of this build")
            self.data_processor_queue.append(getattr(self, name)(config=cur_cfg))

    def mask_points_and_boxes_outside_range(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.mask_points_and_boxes_outside_range, config=config)
        if data_dict.get('points', None) is not None:
            mask = mask_points_by_range(data_dict['points'], self.point_cloud_range)
            data_dict['points'] = data_dict['points'][mask]
        if data_dict.get('gt_boxes', None) is not None and _cfg(config, 'REMOVE_OUTSIDE_BOXES') and self.training:
            mask = mask_boxes_outside_range_numpy(data_dict['gt_boxes'], self.point_cloud_range,
                                                  min_num_corners=_cfg(config, 'min_num_corners', 1),
                                                  use_center_to_filter=_cfg(config, 'USE_CENTER_TO_FILTER', True))
            data_dict['gt_boxes'] = data_dict['gt_boxes'][mask]
        return data_dict

    def shuffle_points(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.shuffle_points, config=config)
        if _cfg(config, 'SHUFFLE_ENABLED')[self.mode]:
            points = data_dict['points']
            data_dict['points'] = points[np.random.permutation(points.shape[0])]
        return data_dict

    def _bind_grid(self, config):
        grid_size = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(_cfg(config, 'VOXEL_SIZE'))
        self.grid_size = np.round(grid_size).astype(np.int64)
        self.voxel_size = _cfg(config, 'VOXEL_SIZE')

    def transform_points_to_voxels_placeholder(self, data_dict=None, config=None):
        if data_dict is None:
            self._bind_grid(config)
            return partial(self.transform_points_to_voxels_placeholder, config=config)
        return data_dict

    def transform_points_to_voxels(self, data_dict=None, config=None):
        if data_dict is None:
            self._bind_grid(config)
            return partial(self.transform_points_to_voxels, config=config)
        if self.voxel_generator is None:   # created lazily (pickling across dataloader workers, :266-273)
            self.voxel_generator = VoxelGeneratorWrapper(
                vsize_xyz=_cfg(config, 'VOXEL_SIZE'), coors_range_xyz=self.point_cloud_range,
                num_point_features=self.num_point_features, max_num_points_per_voxel=_cfg(config, 'MAX_POINTS_PER_VOXEL'),
                max_num_voxels=_cfg(config, 'MAX_NUMBER_OF_VOXELS')[self.mode])
        if _cfg(config, 'DOUBLE_FLIP', False):
            raise NotImplementedError("DOUBLE_FLIP test-time augmentation is not on the hot path of this build")
        voxels, coordinates, num_points = self.voxel_generator.generate(data_dict['points'])
        if not data_dict['use_lead_xyz']:
            voxels = voxels[..., 3:]
        data_dict['voxels'], data_dict['voxel_coords'], data_dict['voxel_num_points'] = voxels, coordinates, num_points
        return data_dict

    def forward(self, data_dict):
        for cur_processor in self.data_processor_queue:
            data_dict = cur_processor(data_dict=data_dict)
        return data_dict
