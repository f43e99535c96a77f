#!/usr/bin/env python3
"""Golden vectors for the pre-computed 2D detection loaders by RUNNING THE REFERENCE's
pcdet/models/preprocessed_detector.py (PreprocessedGLIP, PreprocessedDetector) on synthetic prediction files.

Runs in the build container only (needs /root/reference).  Committed outputs (tests/golden/detector/): the synthetic
INPUT files — glip_pred.pth (a list of BoxList objects pickled under maskrcnn_benchmark's module path, exactly the form
of the GLIP predictions the reference loads), glip_meta.coco.json, cam_<i>.json / cam1based_<i>.json / gt_<i>.json
(COCO files per camera) — and detector_golden.npz, the tensors the reference classes returned.  Data only.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("FNP_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "detector")
CAMS = ['CAM_BACK', 'CAM_BACK_LEFT', 'CAM_BACK_RIGHT', 'CAM_FRONT', 'CAM_FRONT_LEFT', 'CAM_FRONT_RIGHT']
CLASSES = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle', 'bicycle', 'pedestrian', 'traffic_cone']
N_SCENES = 4


def boxlist_module():
    """maskrcnn_benchmark.structures.bounding_box.BoxList as far as pickling goes: bbox, size, mode, extra_fields."""
    pkg = types.ModuleType("maskrcnn_benchmark")
    sub = types.ModuleType("maskrcnn_benchmark.structures")
    mod = types.ModuleType("maskrcnn_benchmark.structures.bounding_box")

    class BoxList(object):
        def __init__(self, bbox, image_size, mode="xyxy"):
            self.bbox, self.size, self.mode, self.extra_fields = bbox, image_size, mode, {}

    BoxList.__module__ = mod.__name__
    BoxList.__qualname__ = "BoxList"
    mod.BoxList = BoxList
    sys.modules[pkg.__name__], sys.modules[sub.__name__], sys.modules[mod.__name__] = pkg, sub, mod
    return BoxList


def image_path(s, c):
    return f"../data/nuscenes/v1.0-trainval/samples/{CAMS[c]}/n015-2018-07-{s:02d}__{CAMS[c]}__15{s}{c}0000.jpg"


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)
    BoxList = boxlist_module()
    preds, images = [], []
    for s in range(N_SCENES):
        for c in range(6):
            n = int(rng.integers(0, 9)) if not (s == 1 and c == 2) else 0      # one image without detections
            xy = rng.uniform(0, 1400, size=(n, 2))
            wh = rng.uniform(10, 200, size=(n, 2))
            bl = BoxList(torch.tensor(np.concatenate([xy, xy + wh], 1), dtype=torch.float32), (1600, 900))
            bl.extra_fields["scores"] = torch.tensor(rng.uniform(0.05, 0.99, size=n), dtype=torch.float32)
            bl.extra_fields["labels"] = torch.tensor(rng.integers(1, 11, size=n), dtype=torch.int64)
            preds.append(bl)
            images.append({"id": len(images), "token": f"token{s:04d}", "file_name": image_path(s, c), "width": 1600, "height": 900})
    torch.save(preds, os.path.join(OUT, "glip_pred.pth"))
    meta = {"images": images, "categories": [{"id": i, "name": n} for i, n in enumerate(CLASSES)], "annotations": []}
    json.dump(meta, open(os.path.join(OUT, "glip_meta.coco.json"), "w"))

    # COCO jsons per camera: predictions (0-based category ids), a 1-based variant, ground truth without scores
    cats = [{"id": i, "name": n} for i, n in enumerate(CLASSES)]
    for kind in ("cam", "cam1based", "gt"):
        for c in range(6):
            imgs, anns = [], []
            for s in range(N_SCENES):
                if kind == "gt" and s == 3:
                    continue                       # an image the file does not know (skipped by infer_nusc)
                imgs.append({"id": 100 * c + s, "file_name": image_path(s, c)})
                for _ in range(int(rng.integers(0, 6))):
                    x, y, w, h = rng.uniform(0, 1300), rng.uniform(0, 700), rng.uniform(8, 250), rng.uniform(8, 180)
                    a = {"id": len(anns), "image_id": 100 * c + s, "bbox": [float(x), float(y), float(x + w), float(y + h)],
                         "category_id": int(rng.integers(0, 10)) + (1 if kind == "cam1based" else 0)}
                    if kind != "gt":
                        a["score"] = float(rng.uniform(0.1, 1.0))
                    anns.append(a)
            if kind == "cam1based" and not any(a["category_id"] == 10 for a in anns):
                anns.append({"id": len(anns), "image_id": 100 * c, "bbox": [1.0, 2.0, 30.0, 40.0], "category_id": 10, "score": 0.5})
            json.dump({"images": imgs, "annotations": anns, "categories": cats}, open(os.path.join(OUT, f"{kind}_{c}.json"), "w"))

    # the reference predates torch's weights_only default: its plain torch.load(path) is the pickle-everything form
    _load = torch.load
    torch.load = lambda *a, **k: _load(*a, **{**k, "weights_only": k.get("weights_only", False)})
    spec = importlib.util.spec_from_file_location("ref_preprocessed_detector", os.path.join(REF, "pcdet/models/preprocessed_detector.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    def batch(scenes):
        return {"batch_size": len(scenes), "image_paths": [[image_path(s, c) for c in range(6)] for s in scenes],
                "metadata": [{"token": f"token{s:04d}"} for s in scenes]}

    save = {}
    glip = ref.PreprocessedGLIP(pred_pth=os.path.join(OUT, "glip_pred.pth"), meta_coco=os.path.join(OUT, "glip_meta.coco.json"))
    for tag, scenes in (("glip_b1", [2]), ("glip_b3", [0, 3, 1])):
        for name, t in zip(("boxes", "labels", "scores", "idx", "cam"), glip(batch(scenes))):
            save[f"{tag}_{name}"] = t.numpy()
    for kind, names in (("cam", None), ("cam1based", None), ("gt", ["car", "pedestrian", "bicycle"])):
        det = ref.PreprocessedDetector([os.path.join(OUT, f"{kind}_{c}.json") for c in range(6)], class_names=names or [])
        for tag, scenes in (("b1", [1]), ("b2", [3, 0])):
            for name, t in zip(("boxes", "labels", "scores", "idx", "cam"), det(batch(scenes))):
                save[f"{kind}_{tag}_{name}"] = t.numpy()
    one = ref.PreprocessedDetector([os.path.join(OUT, "cam_3.json")], class_names=[])      # KITTI-style: frame_id, one camera
    stems = [os.path.splitext(os.path.basename(image_path(s, 3)))[0] for s in (0, 2)]
    one.incl_ext = False
    one.name_to_anns = {os.path.splitext(k)[0]: v for k, v in one.name_to_anns.items()}
    for name, t in zip(("boxes", "labels", "scores", "idx", "cam"), one({"batch_size": 2, "frame_id": stems})):
        save[f"kitti_{name}"] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "detector_golden.npz"), **save)
    print({k: v.shape for k, v in save.items() if k.endswith("boxes")})


if __name__ == "__main__":
    main()
