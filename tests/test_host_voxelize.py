"""The dataloader-side voxeliser (fnp_host_voxelize behind spconv.utils.Point2VoxelCPU3d / VoxelGenerator /
processor.VoxelGeneratorWrapper; reference call sites data_processor.py:17-62,255-302): host code, no GPU — bit-equal to
the oracle's sequential restatement incl. the max_points / max_voxels cuts, usable inside FORKED DataLoader workers
(the GPU-backed generator could not be), picklable for spawn workers."""
import pickle

import numpy as np
import pytest
import torch

from findnpropagate_amd import synthetic as syn
from findnpropagate_amd.processor import DataProcessor, VoxelGeneratorWrapper


@pytest.mark.parametrize("max_points,max_voxels", [(10, 160000), (3, 160000), (10, 500), (1, 7)])
def test_matches_oracle(oracle, max_points, max_voxels):
    p = syn.make_scene(11)
    g = VoxelGeneratorWrapper(vsize_xyz=syn.VOXEL_SIZE, coors_range_xyz=syn.POINT_CLOUD_RANGE, num_point_features=5,
                              max_num_points_per_voxel=max_points, max_num_voxels=max_voxels)
    voxels, coords, num = g.generate(p)
    v, c, n = oracle.voxelize(p, syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, max_points, max_voxels)
    assert coords.dtype == np.int32 and coords.shape[1] == 3 and voxels.shape[1:] == (max_points, 5)
    assert np.array_equal(coords, c) and np.array_equal(num, n) and np.array_equal(voxels, v)
    assert len(c) <= max_voxels and num.max() <= max_points


def test_edge_inputs_and_spconv1_signature(oracle):
    from findnpropagate_amd.spconv.utils import VoxelGenerator
    rng = [0.0, 0.0, 0.0, 1.0, 1.0, 1.0]
    g = VoxelGeneratorWrapper([0.1, 0.1, 0.1], rng, 5, 4, 100)
    v, c, n = g.generate(np.zeros((0, 5), np.float32))
    assert v.shape == (0, 4, 5) and c.shape == (0, 3) and n.shape == (0,)
    # on-boundary points: x == max is outside, x == min is inside, tiny negative is outside
    p = np.array([[0.55, 0.25, 0.95, 7, 0], [1.0, 0.5, 0.5, 0, 0], [-1e-9, 0.5, 0.5, 0, 0], [0.55, 0.25, 0.95, 9, 1], [0.0, 0.0, 0.0, 1, 1]], np.float32)
    v, c, n = g.generate(p)
    assert c.tolist() == [[9, 2, 5], [0, 0, 0]] and n.tolist() == [2, 1] and v[0, :2, 3].tolist() == [7.0, 9.0]
    out = VoxelGenerator([0.1, 0.1, 0.1], rng, 4, max_voxels=100).generate(p)       # spconv 1.x: (voxels, coords, num)
    assert np.array_equal(out[1], c) and np.array_equal(out[0], v)


class _Scenes(torch.utils.data.Dataset):
    def __init__(self):
        cfgs = [{"NAME": "mask_points_and_boxes_outside_range", "REMOVE_OUTSIDE_BOXES": True},
                {"NAME": "transform_points_to_voxels", "VOXEL_SIZE": syn.VOXEL_SIZE, "MAX_POINTS_PER_VOXEL": 10,
                 "MAX_NUMBER_OF_VOXELS": {"train": 120000, "test": 160000}}]
        self.proc = DataProcessor(cfgs, syn.POINT_CLOUD_RANGE, training=False, num_point_features=5)

    def __len__(self):
        return 3

    def __getitem__(self, i):
        d = self.proc.forward({"points": syn.make_scene(60 + i, n_azimuth=200), "use_lead_xyz": True})
        return {k: torch.from_numpy(d[k]) for k in ("voxels", "voxel_coords", "voxel_num_points")}


def test_runs_inside_forked_dataloader_workers(oracle):
    """DataProcessor creates the generator lazily inside the worker (data_processor.py:260-271); with the fork start
    method a GPU-backed generator raises 'Cannot re-initialize CUDA in forked subprocess' — the host one just works."""
    ds = _Scenes()
    pickle.loads(pickle.dumps(VoxelGeneratorWrapper(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 1000)))   # spawn workers pickle it
    loader = torch.utils.data.DataLoader(ds, batch_size=None, num_workers=2, multiprocessing_context="fork")
    got = list(loader)
    assert len(got) == 3
    for i, g in enumerate(got):
        p = syn.make_scene(60 + i, n_azimuth=200)
        v, c, n = oracle.voxelize(p, syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 10, 160000)
        assert np.array_equal(g["voxel_coords"].numpy(), c) and np.array_equal(g["voxels"].numpy(), v)
        assert np.array_equal(g["voxel_num_points"].numpy(), n)
