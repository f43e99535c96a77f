"""Host logic of the dataloader processors on the hot path (data_processor.py:64-106,228-236): queue order,
inclusive x/y range mask (z untouched), centre / corner box filters, seeded shuffle, grid size.  The voxel
generator itself needs the GPU (tests/test_gpu_voxelize.py)."""
import numpy as np
import pytest

from findnpropagate_amd.processor import data_processor as DP

RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]


def test_queue_mask_and_shuffle(rng):
    cfgs = [{"NAME": "mask_points_and_boxes_outside_range", "REMOVE_OUTSIDE_BOXES": True},
            {"NAME": "shuffle_points", "SHUFFLE_ENABLED": {"train": True, "test": False}},
            {"NAME": "transform_points_to_voxels_placeholder", "VOXEL_SIZE": [0.075, 0.075, 0.2]}]
    pts = rng.uniform(-60, 60, (5000, 5)).astype(np.float32)
    pts[0, :2] = [54.0, -54.0]            # on the boundary: kept (inclusive)
    pts[1, :3] = [0.0, 0.0, 100.0]        # z is not tested
    pts[2, :2] = [54.0001, 0.0]
    boxes = np.array([[0, 0, 0, 4, 2, 1.5, 0.3, 1], [60, 0, 0, 4, 2, 1.5, 0.0, 2], [0, 0, 4.0, 1, 1, 1, 0, 3]], np.float32)
    want = (pts[:, 0] >= -54) & (pts[:, 0] <= 54) & (pts[:, 1] >= -54) & (pts[:, 1] <= 54)
    for training in (False, True):
        p = DP.DataProcessor(cfgs, RANGE, training=training, num_point_features=5)
        assert p.grid_size.tolist() == [1440, 1440, 40] and p.voxel_size == [0.075, 0.075, 0.2]
        np.random.seed(3)
        out = p.forward({"points": pts.copy(), "gt_boxes": boxes.copy()})
        assert out["points"].shape[0] == int(want.sum())
        if training:
            np.random.seed(3)
            assert np.array_equal(out["points"], pts[want][np.random.permutation(int(want.sum()))])
            assert out["gt_boxes"][:, 7].tolist() == [1.0]          # centre filter incl. z, training only
        else:
            assert np.array_equal(out["points"], pts[want]) and out["gt_boxes"].shape[0] == 3
    assert want[0] and want[1] and not want[2]
    with pytest.raises(NotImplementedError):
        DP.DataProcessor([{"NAME": "sample_points"}], RANGE, False, 5)


def test_box_corner_filter():
    b = np.array([[53.0, 0, 0, 4, 2, 1.5, 0.0], [53.0, 0, 0, 4, 2, 1.5, np.pi / 2], [0, 0, 0, 200, 200, 1, 0]], np.float32)
    m1 = DP.mask_boxes_outside_range_numpy(b, RANGE, min_num_corners=8, use_center_to_filter=False)
    m2 = DP.mask_boxes_outside_range_numpy(b, RANGE, min_num_corners=1, use_center_to_filter=False)
    assert m1.tolist() == [False, True, False] and m2.tolist() == [True, True, False]
