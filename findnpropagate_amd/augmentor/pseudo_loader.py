"""Host-side operators of the pseudo-label mixing (pcdet/datasets/augmentor/pseudo_loader.py), the consumer
of the `.pth` files the extraction writes.  Same names, arguments and returns for:

  remove_empty          :14-27    boxes with dx > 0 and dy > 0
  bev_nms_cpu           :29-55    greedy rotated-BEV NMS on HOST tensors (IoU matrix from the library's host
                                  entry point instead of the reference's iou3d_cpu.cpp; vectorised sweep
                                  instead of the O(N^2) Python double loop, same kept set and order)
  read_pseudo_file      :574-603  the `.pth` reader of PseudoLoader.load_pseudos (format written by
                                  tools/extract_pseudo_labels.py:134-137 / findnpropagate_amd.extract.save_frame)
  points_in_boxes       :270-316  PseudoSampler.points_in_boxes: dense (T, N) membership + box-frame points

The stateful policy around them (EMA score thresholds, per-class quotas, the copy-paste queue) stays the
reference's own Python: it is dataset bookkeeping, not an operator."""
import os
from pathlib import Path

import numpy as np
import torch

from .. import lib as _l
from ..iou3d_nms import iou3d_nms_utils


def remove_empty(pseudo_boxes):
    non_empty = np.bitwise_and(pseudo_boxes[:, 3] > 0, pseudo_boxes[:, 4] > 0)
    return pseudo_boxes[non_empty], non_empty


def bev_nms_cpu(boxes, scores, thresh=0.5):
    """boxes (N,7) host tensor, scores (N,) -> kept indices in descending-score order (order[keep])."""
    N = boxes.shape[0]
    order = torch.argsort(-scores)
    if N == 0:
        return order
    ious = iou3d_nms_utils.boxes_bev_iou_cpu(boxes.float().contiguous(), boxes.float().contiguous())
    sup = (ious[order][:, order] > thresh).numpy()            # in sorted positions
    keep = np.ones((N,), bool)
    for i in range(N):                                        # position i suppresses every later position it overlaps
        if keep[i]:
            keep[i + 1:] &= ~sup[i, i + 1:]
    return order[torch.from_numpy(keep)]


def read_pseudo_file(folder, frame_id):
    """-> (pred_boxes (M,7+), pred_scores (M,), pred_labels (M,)) numpy, or None when the file is missing or
    unreadable (load_pseudos returns empty arrays in both cases)."""
    path = Path(folder) / f"{str(frame_id).replace('.', '_')}.pth"
    if not os.path.exists(path):
        return None
    try:
        preds = torch.load(path, map_location="cpu")
    except Exception:
        return None
    assert len(preds) == 1 or isinstance(preds, dict), f"preds dict should have len==1, got {len(preds)} {type(preds)}"
    d = preds if isinstance(preds, dict) else preds[0]
    return d["pred_boxes"].numpy(), d["pred_scores"].numpy(), d["pred_labels"].numpy()


def points_in_boxes(points, boxes3d):
    """points (N,5) numpy, boxes3d (T,7+) -> (in_box (T,N) bool, points in every box frame (T,N,5) f32)."""
    points = np.ascontiguousarray(points, np.float32)
    boxes = np.ascontiguousarray(boxes3d[:, :7], np.float32)
    assert points.shape[-1] == 5
    T, N, C = boxes.shape[0], points.shape[0], points.shape[1]
    in_box = np.zeros((T, N), np.uint8)
    out = np.empty((T, N, C), np.float32)
    rc = _l.load().fnp_host_points_in_boxes_frame(points.ctypes.data, N, C, boxes.ctypes.data, T, in_box.ctypes.data, out.ctypes.data)
    _l.check(rc, "fnp_host_points_in_boxes_frame")
    return in_box.astype(bool), out
