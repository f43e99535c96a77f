"""Drop-in for the voxelisation boundary of pcdet/datasets/processor/data_processor.py:17-62.

`VoxelGeneratorWrapper(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
max_num_voxels).generate(points)` returns (voxels (M,P,C) f32, coordinates (M,3) int32 [z,y,x],
num_points (M,) int32) with the reference's sequential first-come semantics, computed on the
MI355X.
"""
from ..spconv.utils import Point2VoxelCPU3d


class VoxelGeneratorWrapper:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        self.spconv_ver = 2
        self._voxel_generator = Point2VoxelCPU3d(
            vsize_xyz=vsize_xyz, coors_range_xyz=coors_range_xyz, num_point_features=num_point_features,
            max_num_points_per_voxel=max_num_points_per_voxel, max_num_voxels=max_num_voxels)

    def generate(self, points):
        tv_voxels, tv_coordinates, tv_num_points = self._voxel_generator.point_to_voxel(points)
        return tv_voxels.numpy(), tv_coordinates.numpy(), tv_num_points.numpy()
