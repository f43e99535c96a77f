"""Host operators of the pseudo-label mixing (no GPU): bev_nms_cpu, the .pth reader, PseudoSampler's
points_in_boxes, against the oracle's statement-by-statement restatements of
pcdet/datasets/augmentor/pseudo_loader.py."""
import os
import tempfile

import numpy as np
import pytest
import torch

from findnpropagate_amd import extract as E, synthetic as syn
from findnpropagate_amd.augmentor import pseudo_loader as PL


@pytest.mark.parametrize("n,thresh", [(0, 0.5), (1, 0.5), (60, 0.5), (120, 0.1), (80, 0.9)])
def test_bev_nms_cpu(oracle, rng, n, thresh):
    boxes = syn.random_boxes(rng, n, centre_range=8.0) if n else np.zeros((0, 7), np.float32)
    scores = rng.permutation(max(n, 1))[:n].astype(np.float32) / max(n, 1)
    got = PL.bev_nms_cpu(torch.from_numpy(boxes), torch.from_numpy(scores), thresh)
    want = oracle.bev_nms_cpu(boxes, scores, thresh)
    assert np.array_equal(got.numpy(), want)
    if n == 120:
        assert len(want) < n


def test_remove_empty_and_reader_roundtrip(rng):
    b = syn.random_boxes(rng, 6)
    b[1, 3] = 0
    b[4, 4] = -1
    kept, mask = PL.remove_empty(b)
    assert mask.tolist() == [True, False, True, True, False, True] and kept.shape == (4, 7)
    with tempfile.TemporaryDirectory() as d:
        pd = dict(pred_boxes=torch.from_numpy(b), pred_scores=torch.rand(6), pred_labels=torch.randint(1, 11, (6,), dtype=torch.int32))
        E.save_frame(d, "n008-2018.pcd.bin", pd)
        got = PL.read_pseudo_file(d, "n008-2018.pcd.bin")
        assert np.array_equal(got[0], b) and np.array_equal(got[1], pd["pred_scores"].numpy()) and got[2].dtype == np.int32
        assert PL.read_pseudo_file(d, "missing.pcd.bin") is None
        open(os.path.join(d, "bad_pcd_bin.pth"), "wb").write(b"not a checkpoint")
        assert PL.read_pseudo_file(d, "bad.pcd.bin") is None


def test_points_in_boxes_frame(oracle, rng):
    boxes = syn.random_boxes(rng, 9, centre_range=10.0)
    pts = np.concatenate([rng.uniform(-14, 14, (4000, 2)), rng.uniform(-3, 2, (4000, 1)), rng.uniform(0, 1, (4000, 2))], 1).astype(np.float32)
    pts[:9, :3] = boxes[:, :3]                      # box centres are inside
    pts[9, :3] = boxes[0, :3] + np.array([0, 0, boxes[0, 5] / 2], np.float32)   # on the top face: inclusive
    got_in, got_p = PL.points_in_boxes(pts, np.concatenate([boxes, np.ones((9, 1), np.float32)], 1))
    want_in, want_p = oracle.pseudo_points_in_boxes(pts, boxes)
    assert got_in.shape == (9, 4000) and got_p.shape == (9, 4000, 5)
    np.testing.assert_allclose(got_p, want_p, rtol=0, atol=2e-5)
    diff = got_in != want_in
    # libm vs numpy trig may differ in the last bit: only points within 1e-4 of a face may flip
    if diff.any():
        t, i = np.nonzero(diff)
        d = np.abs(np.abs(want_p[t, i, :3]) - boxes[t, 3:6] / 2).min(axis=1)
        assert (d < 1e-4).all()
    assert got_in[np.arange(9), np.arange(9)].all() and got_in[0, 9]
    assert 0 < got_in.sum() < got_in.size
