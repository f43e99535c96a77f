"""Host entry points of the library (no GPU): boxes_iou_bev_cpu / boxes_aligned_iou_bev_cpu
(iou3d_cpu.cpp:232-272, consumer pseudo_loader.py:29-55) against the oracle's restatement of the same
overlap algorithm and against the known answers of the reference's own code (overlap
5.68089008 / IoU 0.550521314 for the two probe boxes).  The reference's iou3d_cpu.cpp itself needs
cuda.h / cuda_runtime_api.h, which this image lacks: unbuildable here, so the host path is pinned
through the oracle (whose device twin is pinned against the reference .cu on the GPU box)."""
import numpy as np
import pytest
import torch

from findnpropagate_amd import synthetic as syn
from findnpropagate_amd.iou3d_nms import iou3d_nms_cuda, iou3d_nms_utils


def test_known_answer():
    a = np.array([[0, 0, 0, 4, 2, 1.5, 0.3]], np.float32)
    b = np.array([[0.5, 0.2, 0, 4, 2, 1.5, -0.2]], np.float32)
    iou = iou3d_nms_utils.boxes_bev_iou_cpu(a, b)
    assert isinstance(iou, np.ndarray) and iou.shape == (1, 1)
    assert iou[0, 0] == pytest.approx(0.550521314, rel=2e-6)


@pytest.mark.parametrize("n,m", [(1, 1), (7, 5), (64, 33), (0, 4), (3, 0)])
def test_matches_oracle(oracle, rng, n, m):
    a = syn.random_boxes(rng, n) if n else np.zeros((0, 7), np.float32)
    b = syn.random_boxes(rng, m) if m else np.zeros((0, 7), np.float32)
    if n and m:
        b[0] = a[0]                    # identical boxes
        if m > 1:
            b[1, :2] = a[0, :2] + 100  # disjoint
    got = iou3d_nms_utils.boxes_bev_iou_cpu(torch.from_numpy(a), torch.from_numpy(b))
    assert got.shape == (n, m) and got.dtype == torch.float32
    if n and m:
        exp = oracle.boxes_iou_bev(a, b)
        assert np.array_equal(got.numpy(), exp), np.abs(got.numpy() - exp).max()
        assert got[0, 0] == pytest.approx(1.0, abs=1e-5)
        if m > 1:
            assert got[0, 1] == 0.0


def test_aligned_and_conventions(oracle, rng):
    a, b = syn.random_boxes(rng, 40), syn.random_boxes(rng, 40)
    b[:10, :2] = a[:10, :2] + rng.uniform(-1, 1, (10, 2)).astype(np.float32)   # make some pairs overlap
    out = torch.zeros((40, 1))
    assert iou3d_nms_cuda.boxes_aligned_iou_bev_cpu(torch.from_numpy(a), torch.from_numpy(b), out) == 1
    exp = np.diagonal(oracle.boxes_iou_bev(a, b))
    assert np.array_equal(out.numpy()[:, 0], exp)
    assert (out.numpy()[:10] > 0).any()
    with pytest.raises(Exception):
        iou3d_nms_cuda.boxes_iou_bev_cpu(torch.from_numpy(a).double(), torch.from_numpy(b), torch.zeros((40, 40)))
