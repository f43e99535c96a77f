"""SparseConvTensor (spconv_backbone.py:256-261 constructs it; height_compression.py:21 calls
.dense(); spconv_utils.py:32-38 calls .replace_feature)."""
import torch

from .. import sparse as S


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None,
                 indice_dict=None, benchmark=False, n_dev=None, rank_grid=None):
        assert features.dim() == 2 and indices.dim() == 2 and indices.shape[1] == 4
        assert indices.dtype == torch.int32, "indices must be int32 [b, z, y, x]"
        self.features = features
        self.indices = indices if indices.is_contiguous() else indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        # device-side row count + rank grid of `indices` (built lazily, shared by SubM outputs)
        self._n_dev = n_dev
        self._rank_grid = rank_grid

    # ---- spconv API -----------------------------------------------------------------------
    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def replace_feature(self, feature):
        t = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.grid, self.voxel_num,
                             self.indice_dict, self.benchmark, self._n_dev, self._rank_grid)
        return t

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        out = S.to_dense(self.features.contiguous(), self.indices, self.n_dev(), self.batch_size, self.spatial_shape)
        if not channels_first:
            out = out.permute(0, 2, 3, 4, 1).contiguous()
        return out

    # ---- device-side bookkeeping ----------------------------------------------------------
    @property
    def rows_ranked(self):
        """Are the rows in rank-grid order (the output of a strided convolution, and whatever SubM layers make of it)?  Such a
        tensor carries a rank grid without a row permutation."""
        return self._rank_grid is not None and self._rank_grid.perm is None

    def n_dev(self):
        if self._n_dev is None:
            self._n_dev = S.device_scalar(self.features.shape[0], self.features.device)
        return self._n_dev

    def rank_grid(self):
        if self._rank_grid is None:
            self._rank_grid = S.build_grid(self.indices, self.n_dev(), self.batch_size, self.spatial_shape,
                                           keep_order=True)
        return self._rank_grid
