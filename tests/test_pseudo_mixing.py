"""The stateful half of the pseudo-label mixing: PseudoLoader (load_pseudos with the EMA score
thresholds and per-class top-k, load_frustum_pseudos, load_selftrain_pseudos, copy_and_paste with the PseudoSampler
queue) and PseudoProcessor (__call__, save_predictions, undo_augmentations) against what the REFERENCE's own classes
produced on the same seeded scenario (tests/golden/make_pseudo_golden.py -> tests/golden/pseudo_golden.npz), step by
step over two epochs of five frames, two configurations.  np.random is consumed in the reference's order, so the
pasted samples are the same objects at the same places.  Host code (numpy + fnp_host_* entry points), CPU suite;
re-run in the GPU set by tests/test_gpu_host_abi.py."""
import os
import tempfile

import numpy as np
import pytest

import pseudo_scenario as SC
from findnpropagate_amd.augmentor.pseudo_loader import PseudoLoader
from findnpropagate_amd.dense_heads.pseudo_processor import PseudoProcessor

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pseudo_golden.npz")


def _compare(got, gold, prefix):
    keys = [k for k in gold.files if k.startswith(prefix)]
    assert keys and set(keys) == {k for k in got if k.startswith(prefix)}
    for k in keys:
        g, w = got[k], gold[k]
        assert g.shape == w.shape, (k, g.shape, w.shape)
        if w.dtype.kind in "biUS":
            assert np.array_equal(g, w), k
        elif k.endswith("points_sum"):
            # pasted points come from the box-frame points of the host point-in-box entry (f32, libm trig vs torch's)
            np.testing.assert_allclose(g, w, rtol=1e-6, err_msg=k)
        else:
            np.testing.assert_allclose(g, w, rtol=0, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("name", list(SC.LOADER_CONFIGS))
def test_loader_sequence_matches_reference(name):
    gold = np.load(GOLD)
    with tempfile.TemporaryDirectory() as fr, tempfile.TemporaryDirectory() as st:
        frames = SC.make_frames(fr, st)
        got = SC.run_loader(PseudoLoader, name, fr, st, frames)
    _compare(got, gold, name + "_")
    masks = [gold[k] for k in gold.files if k.startswith(name) and k.endswith("_mask")]
    assert sum(int(m.sum()) for m in masks) >= 5, "the scenario pastes samples"
    assert any(gold[k].max() >= 2 for k in gold.files if k.startswith(name) and k.endswith("_queue")), "queues fill up"


def test_processor_matches_reference():
    gold = np.load(GOLD)
    with tempfile.TemporaryDirectory() as st:
        got = SC.run_processor(PseudoProcessor, os.path.join(st, "selftrain"))
    _compare(got, gold, "proc_")
    assert gold["proc_e1_cons"].sum() > 0, "second epoch finds last round's boxes"
    assert gold["proc_e0_f0_pred_boxes"].shape[0] < 8, "predictions on pasted samples were dropped"
