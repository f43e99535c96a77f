"""spconv.utils voxel generators (call sites pcdet/datasets/processor/data_processor.py:17-62).
Point2VoxelCPU3d / VoxelGenerator run the library's HOST voxeliser (safe in DataLoader workers, like the
originals); Point2VoxelGPU3d runs the device one.  Inputs/outputs are host numpy arrays in both cases; the fused
model path (backbones_3d.VoxelResBackBone8x.forward_points) keeps everything on the device instead.
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
    """spconv 2.x signature (data_processor.py:38-44).  HOST voxeliser (the library's fnp_host_voxelize): no GPU is
    touched, so it works inside forked DataLoader workers exactly like spconv's CPU class.  Results equal the GPU
    voxeliser's (Point2VoxelGPU3d / sparse.voxelize) bit for bit."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel):
        self.cfg = S.make_voxel_cfg(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
                                    max_num_voxels)
        self.grid_size = [self.cfg.grid[0], self.cfg.grid[1], self.cfg.grid[2]]

    def __getstate__(self):      # ctypes structs do not pickle (DataLoader workers under spawn)
        c = self.cfg
        return dict(vs=list(c.voxel_size), rng=list(c.range_min) + [c.range_min[d] + c.voxel_size[d] * c.grid[d] for d in range(3)],
                    nf=c.num_features, mp=c.max_points, mv=c.max_voxels)

    def __setstate__(self, st):
        self.__init__(st["vs"], st["rng"], st["nf"], st["mv"], st["mp"])

    def point_to_voxel(self, points):
        pts = np.ascontiguousarray(_as_numpy(points), dtype=np.float32)
        n = pts.shape[0]
        C, P = self.cfg.num_features, self.cfg.max_points
        assert pts.ndim == 2 and pts.shape[1] == C
        rows = max(min(n, self.cfg.max_voxels), 1)
        voxels = np.empty((rows, P, C), np.float32)
        coords = np.empty((rows, 3), np.int32)
        num = np.empty((rows,), np.int32)
        L = S._l.load()
        m = L.fnp_host_voxelize(pts.ctypes.data, n, self.cfg, voxels.ctypes.data, coords.ctypes.data, num.ctypes.data, rows)
        if m < 0:
            S._l.check(m, "fnp_host_voxelize")
        return _HostArray(voxels[:m]), _HostArray(coords[:m]), _HostArray(num[:m])


class Point2VoxelGPU3d:
    """The same generator on the device (spconv 2.x has this class too): host arrays in and out, H2D -> kernels -> D2H
    per call.  Needs a GPU context in the calling process (num_workers = 0 or the spawn start method); the fused model
    path (backbones_3d.VoxelResBackBone8x.forward_points) keeps everything on the device instead."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel, device=None):
        self.cfg = S.make_voxel_cfg(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
                                    max_num_voxels)
        self.grid_size = [self.cfg.grid[0], self.cfg.grid[1], self.cfg.grid[2]]
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

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
