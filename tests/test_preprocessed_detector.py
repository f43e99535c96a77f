"""Pre-computed 2D detection loaders (PreprocessedGLIP / PreprocessedDetector,
pcdet/models/preprocessed_detector.py:7-290) against tensors the reference's own classes returned on the same files
(tests/golden/make_detector_golden.py -> tests/golden/detector/).  CPU only; re-run in the GPU set by
tests/test_gpu_host_abi.py.  The GLIP file holds BoxList objects pickled under maskrcnn_benchmark's module path, which
does not exist here: loading it exercises the lenient unpickler."""
import os

import numpy as np
import pytest
import torch

from findnpropagate_amd.preprocessed_detector import PreprocessedDetector, PreprocessedGLIP, load_glip_predictions

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detector")
CAMS = ['CAM_BACK', 'CAM_BACK_LEFT', 'CAM_BACK_RIGHT', 'CAM_FRONT', 'CAM_FRONT_LEFT', 'CAM_FRONT_RIGHT']
NAMES = ("boxes", "labels", "scores", "idx", "cam")


def image_path(s, c):
    return f"../data/nuscenes/v1.0-trainval/samples/{CAMS[c]}/n015-2018-07-{s:02d}__{CAMS[c]}__15{s}{c}0000.jpg"


def batch(scenes):
    return {"batch_size": len(scenes), "image_paths": [[image_path(s, c) for c in range(6)] for s in scenes],
            "metadata": [{"token": f"token{s:04d}"} for s in scenes]}


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(D, "detector_golden.npz"))


def _same(got, gold, tag):
    for name, t in zip(NAMES, got):
        want = gold[f"{tag}_{name}"]
        assert isinstance(t, torch.Tensor) and not t.is_cuda
        assert t.numpy().dtype == want.dtype, (tag, name, t.dtype, want.dtype)
        assert np.array_equal(t.numpy(), want), (tag, name)


def test_glip_loader_matches_reference(gold):
    assert "maskrcnn_benchmark" not in __import__("sys").modules
    glip = PreprocessedGLIP(pred_pth=os.path.join(D, "glip_pred.pth"), meta_coco=os.path.join(D, "glip_meta.coco.json"))
    _same(glip(batch([2])), gold, "glip_b1")
    _same(glip(batch([0, 3, 1])), gold, "glip_b3")
    _same(glip(batch([2])), gold, "glip_b1")          # repeated calls: the tables are not consumed
    assert gold["glip_b3_boxes"].shape[0] > 20 and set(gold["glip_b3_idx"]) == {0, 1, 2}
    bad = batch([1])
    bad["metadata"][0]["token"] = "token0000"
    with pytest.raises(AssertionError):               # token / file-name alignment check (:72,77)
        glip(bad)
    with pytest.raises(TypeError):
        glip({"frame_id": ["000001"], "batch_size": 1})
    # plain-dict prediction files (no BoxList at all) load the same way
    raw = load_glip_predictions(os.path.join(D, "glip_pred.pth"))
    assert hasattr(raw[0], "bbox") and "scores" in raw[0].extra_fields
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        plain = [{"bbox": r.bbox, "scores": r.extra_fields["scores"], "labels": r.extra_fields["labels"]} for r in raw]
        torch.save(plain, os.path.join(td, "plain.pth"))
        g2 = PreprocessedGLIP(pred_pth=os.path.join(td, "plain.pth"), meta_coco=os.path.join(D, "glip_meta.coco.json"))
        _same(g2(batch([0, 3, 1])), gold, "glip_b3")


@pytest.mark.parametrize("kind,names", [("cam", []), ("cam1based", []), ("gt", ["car", "pedestrian", "bicycle"])])
def test_coco_loader_matches_reference(gold, kind, names):
    det = PreprocessedDetector([os.path.join(D, f"{kind}_{c}.json") for c in range(6)], class_names=names)
    _same(det(batch([1])), gold, f"{kind}_b1")
    _same(det(batch([3, 0])), gold, f"{kind}_b2")
    if kind == "gt":
        assert set(gold["gt_b2_labels"]) <= {1, 2, 3} and (gold["gt_b2_scores"] == 1.0).all()
        assert 0 not in gold["gt_b2_idx"], "scene 3 is unknown to the ground-truth files: skipped, not an error"


def test_coco_loader_kitti_form(gold):
    one = PreprocessedDetector([os.path.join(D, "cam_3.json")], class_names=[])
    one.incl_ext = False
    one._rows = {os.path.splitext(k)[0]: v for k, v in one._rows.items()}
    stems = [os.path.splitext(os.path.basename(image_path(s, 3)))[0] for s in (0, 2)]
    _same(one({"batch_size": 2, "frame_id": stems}), gold, "kitti")
    with pytest.raises(ValueError):
        one({"batch_size": 1, "frame_id": ["missing"]})
    with pytest.raises(ValueError):
        PreprocessedDetector([os.path.join(D, "cam_3.json")], class_names=["unicorn"])
