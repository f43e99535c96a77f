"""spconv.utils voxel generators (call sites pcdet/datasets/processor/data_processor.py:17-62),
backed by the GPU voxeliser.  Inputs/outputs are host numpy arrays like the CPU originals, so
each call is H2D -> kernels -> D2H; the fused model path (backbones_3d.VoxelResBackBone8x
.forward_points) keeps everything on the device instead.
"""
import numpy as np
import torch

from .. import sparse as S


class _HostArray:
    """Minimal cumm.tensorview.Tensor look-alike: `.numpy()` / `.numpy_view()`."""

    def __init__(self, arr):
        self._arr = arr

    def numpy(self):
        return self._arr

    def numpy_view(self):
        return self._arr


def _as_numpy(points):
    if isinstance(points, _HostArray):
        return points.numpy()
    if isinstance(points, torch.Tensor):
        return points.detach().cpu().numpy()
    return np.asarray(points)


class Point2VoxelCPU3d:
    """spconv 2.x signature (data_processor.py:38-44)."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel):
        self.cfg = S.make_voxel_cfg(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
                                    max_num_voxels)
        self.grid_size = [self.cfg.grid[0], self.cfg.grid[1], self.cfg.grid[2]]
        self._device = torch.device("cuda", torch.cuda.current_device())

    def point_to_voxel(self, points):
        pts = np.ascontiguousarray(_as_numpy(points), dtype=np.float32)
        n = pts.shape[0]
        dev = self._device
        if n == 0:
            C, P = self.cfg.num_features, self.cfg.max_points
            return (_HostArray(np.zeros((0, P, C), np.float32)), _HostArray(np.zeros((0, 3), np.int32)),
                    _HostArray(np.zeros((0,), np.int32)))
        d_pts = torch.from_numpy(pts).to(dev)
        off = torch.tensor([0, n], dtype=torch.int32, device=dev)
        r = S.voxelize(d_pts, off, 1, self.cfg, want_voxels=True)
        m = int(r["n"].item())
        voxels = r["voxels"][:m].cpu().numpy()
        coords = r["coords"][:m, 1:].contiguous().cpu().numpy()  # [z, y, x]
        num = r["num_points"][:m].cpu().numpy()
        return _HostArray(voxels), _HostArray(coords), _HostArray(num)


class VoxelGenerator:
    """spconv 1.x signature (data_processor.py:31-36): generate(points) -> dict or tuple."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, num_point_features=None):
        self._args = (voxel_size, point_cloud_range, max_num_points, max_voxels)
        self._impl = None
        self._nfeat = num_point_features

    def generate(self, points, max_voxels=None):
        pts = _as_numpy(points)
        if self._impl is None or self._nfeat != pts.shape[1]:
            vs, rng, mp, mv = self._args
            self._nfeat = pts.shape[1]
            self._impl = Point2VoxelCPU3d(vs, rng, self._nfeat, max_voxels or mv, mp)
        v, c, n = self._impl.point_to_voxel(pts)
        return v.numpy(), c.numpy(), n.numpy()


VoxelGeneratorV2 = VoxelGenerator
