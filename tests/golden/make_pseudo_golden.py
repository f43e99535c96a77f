#!/usr/bin/env python3
"""Golden vectors for the pseudo-label mixing by RUNNING THE REFERENCE's
pcdet/datasets/augmentor/pseudo_loader.py (PseudoLoader, PseudoSampler, ObjectSample) and
pcdet/models/dense_heads/pseudo_processor.py (PseudoProcessor) on the seeded scenario of tests/pseudo_scenario.py.

Runs in the build container only (needs /root/reference).  The reference modules are imported from where they lie; the
two compiled ops they call are stood in by the oracle (pinned in tests/test_oracle_ops.py): boxes_bev_iou_cpu ->
oracle rotated BEV IoU.  Output: tests/golden/pseudo_golden.npz (arrays only)."""
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FNP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
import pseudo_scenario as SC  # noqa: E402


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """pcdet/ops/iou3d_nms/iou3d_nms_utils.py:12-28 on top of the oracle's overlap (numpy in -> numpy out)"""
    is_numpy = isinstance(boxes_a, np.ndarray)
    a = torch.from_numpy(boxes_a).float() if isinstance(boxes_a, np.ndarray) else boxes_a
    b = torch.from_numpy(boxes_b).float() if isinstance(boxes_b, np.ndarray) else boxes_b
    assert a.shape[1] == 7 and b.shape[1] == 7
    if a.shape[0] == 0 or b.shape[0] == 0:
        ans = a.new_zeros((a.shape[0], b.shape[0]))
    else:
        ans = torch.from_numpy(O.boxes_iou_bev(a.contiguous().numpy(), b.contiguous().numpy()))
    return ans.numpy() if is_numpy else ans


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def shell(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def load_reference():
    class _Any:
        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    for name in ("SharedArray", "cv2", "easydict", "spconv", "spconv.pytorch", "cumm", "cumm.tensorview", "numba", "tqdm"):
        stub(name).__getattr__ = lambda k: _Any()  # type: ignore
    p = os.path.join(REF, "pcdet")
    shell("pcdet", p)
    shell("pcdet.utils", os.path.join(p, "utils"))
    shell("pcdet.ops", os.path.join(p, "ops"))
    shell("pcdet.ops.iou3d_nms", os.path.join(p, "ops", "iou3d_nms"))
    shell("pcdet.ops.roiaware_pool3d", os.path.join(p, "ops", "roiaware_pool3d"))
    shell("pcdet.datasets", os.path.join(p, "datasets"))
    shell("pcdet.datasets.augmentor", os.path.join(p, "datasets", "augmentor"))
    shell("pcdet.models", os.path.join(p, "models"))
    shell("pcdet.models.dense_heads", os.path.join(p, "models", "dense_heads"))
    shell("pcdet.models.dense_heads.target_assigner", os.path.join(p, "models", "dense_heads", "target_assigner"))
    sys.modules["pcdet.ops.iou3d_nms"].iou3d_nms_utils = stub("pcdet.ops.iou3d_nms.iou3d_nms_utils", boxes_bev_iou_cpu=boxes_bev_iou_cpu)
    sys.modules["pcdet.ops.roiaware_pool3d"].roiaware_pool3d_utils = stub("pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils")
    stub("pcdet.models.dense_heads.target_assigner.hungarian_assigner", HungarianAssigner3D=lambda *a, **k: None)
    pl = importlib.import_module("pcdet.datasets.augmentor.pseudo_loader")
    pp = importlib.import_module("pcdet.models.dense_heads.pseudo_processor")
    return pl, pp


def main():
    _load = torch.load     # the reference predates torch's weights_only default
    torch.load = lambda *a, **k: _load(*a, **{**k, "weights_only": k.get("weights_only", False)})
    pl, pp = load_reference()
    save = {}
    for name in SC.LOADER_CONFIGS:
        with tempfile.TemporaryDirectory() as fr, tempfile.TemporaryDirectory() as st:
            frames = SC.make_frames(fr, st)
            save.update(SC.run_loader(pl.PseudoLoader, name, fr, st, frames))
    with tempfile.TemporaryDirectory() as st:
        save.update(SC.run_processor(pp.PseudoProcessor, os.path.join(st, "selftrain")))
    np.savez_compressed(os.path.join(HERE, "pseudo_golden.npz"), **save)
    n_pasted = sum(int(v.sum()) for k, v in save.items() if k.endswith("_mask"))
    print(len(save), "arrays;", n_pasted, "pasted samples over the run")


if __name__ == "__main__":
    main()
