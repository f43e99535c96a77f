"""MeanVFE pinned by the reference's own class: tests/golden/meanvfe_golden.npz is what
pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31 wrote into batch_dict['voxel_features'] (torch CPU) for a voxel block with full,
partial, single-point and empty voxels (tests/golden/make_meanvfe_golden.py).  oracle.mean_vfe must be that tensor bit for bit;
the GPU set holds the voxeliser's fused mean to the same fixture (tests/test_gpu_voxelize.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meanvfe_golden.npz")


def test_oracle_mean_vfe_equals_the_reference_class_bit_for_bit(oracle):
    d = np.load(GOLD)
    got = oracle.mean_vfe(d["voxels"], d["num_points"])
    assert got.dtype == np.float32 and np.array_equal(got, d["mean"])
    hist = np.bincount(d["num_points"], minlength=11)
    assert hist[0] >= 1 and hist[1] > 0 and hist[10] > 0            # the clamp, single-point and full voxels are in the fixture
    # the fixture separates torch's order from a plain slot-order sum (column 4 of 5-feature points: four interleaved partials)
    seq = np.zeros(d["mean"].shape, np.float32)
    for p in range(d["voxels"].shape[1]):
        seq = (seq + d["voxels"][:, p]).astype(np.float32)
    seq = seq / np.maximum(d["num_points"], 1)[:, None].astype(np.float32)
    wrong = np.nonzero(seq != d["mean"])
    assert len(wrong[0]) > 100 and set(wrong[1].tolist()) == {4}


@pytest.mark.parametrize("M,P,C", [(3000, 10, 5), (3000, 10, 4), (1, 10, 5), (2000, 5, 5), (2000, 10, 6), (2000, 12, 7), (2000, 20, 5),
                                   (1500, 32, 4), (1000, 64, 5), (800, 100, 7), (300, 300, 5), (2000, 3, 5), (2000, 1, 3)])
def test_oracle_mean_vfe_is_torchs_cpu_sum_for_every_shape_below_8_columns(oracle, rng, M, P, C):
    """oracle/fnp_oracle.c restates torch's cascade_sum (aten/src/ATen/native/cpu/SumKernel.cpp) for a contiguous (M, P, C < 8)
    tensor reduced over dim 1; torch itself (installed here and on the GPU box) is the checker."""
    import torch
    v = (rng.normal(size=(M, P, C)) * rng.choice([1e-3, 1.0, 100.0], size=(M, P, C))).astype(np.float32)
    n = rng.integers(0, P + 1, size=M).astype(np.int32)
    v[np.arange(P)[None, :] >= n[:, None]] = 0
    tv, tn = torch.from_numpy(v), torch.from_numpy(n)
    want = tv.sum(dim=1) / torch.clamp_min(tn.view(-1, 1), min=1.0).type_as(tv)      # mean_vfe.py:26-28
    assert np.array_equal(oracle.mean_vfe(v, n), want.numpy())
