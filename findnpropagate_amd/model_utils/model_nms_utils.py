"""pcdet/models/model_utils/model_nms_utils.py:6-66 with the same names, arguments and returns, on the
library's NMS (device-side sweep, `iou3d_nms_utils.nms_gpu / nms_normal_gpu`).

nms_config: object or dict with NMS_TYPE, NMS_THRESH, NMS_PRE_MAXSIZE, NMS_POST_MAXSIZE (the reference
passes the whole EasyDict as **kwargs to the nms function; only `pre_maxsize` is read there)."""
import torch

from ..iou3d_nms import iou3d_nms_utils


def _cfg(nms_config, key):
    return nms_config[key] if isinstance(nms_config, dict) else getattr(nms_config, key)


def _select(box_scores, box_preds, nms_config):
    """top-k by score -> NMS -> first NMS_POST_MAXSIZE kept indices into (box_scores, box_preds)."""
    if box_scores.shape[0] == 0:
        return box_scores.new_zeros((0,), dtype=torch.long)
    box_scores_nms, indices = torch.topk(box_scores, k=min(_cfg(nms_config, "NMS_PRE_MAXSIZE"), box_scores.shape[0]))
    boxes_for_nms = box_preds[indices]
    keep_idx, _ = getattr(iou3d_nms_utils, _cfg(nms_config, "NMS_TYPE"))(
        boxes_for_nms[:, 0:7], box_scores_nms, _cfg(nms_config, "NMS_THRESH"))
    return indices[keep_idx[:_cfg(nms_config, "NMS_POST_MAXSIZE")]]


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.py:6-27: -> (selected indices into the inputs, their scores)."""
    src_box_scores = box_scores
    if score_thresh is not None:
        scores_mask = box_scores >= score_thresh
        box_scores = box_scores[scores_mask]
        box_preds = box_preds[scores_mask]
    selected = _select(box_scores, box_preds, nms_config)
    if score_thresh is not None:
        original_idxs = scores_mask.nonzero().view(-1)
        selected = original_idxs[selected]
    return selected, src_box_scores[selected]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.py:30-66: cls_scores (N, num_class), box_preds (N, 7+C) ->
    (pred_scores, pred_labels (0-based class column), pred_boxes), classes concatenated in order.

    All classes at once (round 5): the score threshold becomes -inf entries, ONE stable sort of the class-major score matrix
    orders every class, ONE launch pair runs the NMS of all classes (fnp_nms_batched, the per-class counts stay on the
    device), and the host reads the kept counts once — the reference (and rounds 1-4 here) walked the classes in Python
    with a top-k, an NMS and a host synchronisation per class."""
    from .. import lib as _l
    n, num_class = cls_scores.shape
    dev = cls_scores.device
    if n == 0 or num_class == 0:
        return (cls_scores.new_zeros((0,)), torch.zeros((0,), dtype=torch.long, device=dev), box_preds.new_zeros((0, box_preds.shape[1])))
    _l.require_device(cls_scores, box_preds)
    rotated = {"nms_gpu": 1, "nms_normal_gpu": 0}[_cfg(nms_config, "NMS_TYPE")]
    per_class = cls_scores.t().float()                                        # (num_class, N)
    # which boxes take part is an explicit mask, not a sentinel score (with -inf as the sentinel a genuine -inf score
    # was dropped when no threshold is given, and a NaN score — first in a descending sort, never counted — pushed valid boxes off
    # the counted prefix).  A NaN never passes a threshold (`>=` is false, as in the reference's mask); without a threshold every
    # box takes part, NaN ones last.
    valid = (per_class >= score_thresh) if score_thresh is not None else torch.ones_like(per_class, dtype=torch.bool)
    key = torch.where(valid & ~torch.isnan(per_class), per_class, per_class.new_full((), float("-inf")))
    cap = min(int(_cfg(nms_config, "NMS_PRE_MAXSIZE")), n)
    # (a STABLE descending sort, not top-k: equal scores keep their index order whatever the backend's top-k does.  Boxes that do
    #  not take part sort behind every box that does: the sort key of an invalid box is -inf AND it is ordered after the valid -inf
    #  ones by sorting on (invalid, -key) — one stable sort on the validity after the stable sort on the score)
    _, idx1 = torch.sort(key, dim=1, descending=True, stable=True)
    _, idx2 = torch.sort((~valid).gather(1, idx1).to(torch.uint8), dim=1, stable=True)
    top_idx = idx1.gather(1, idx2)[:, :cap]
    counts = valid.sum(dim=1).clamp(max=cap).to(torch.int32)                  # boxes of each class that take part (<= cap)
    boxes = box_preds[top_idx.reshape(-1), 0:7].float().contiguous()          # (num_class * cap, 7)
    L = _l.load()
    ws = torch.empty((max(int(L.fnp_nms_batched_workspace_bytes(num_class, cap)), 8),), dtype=torch.uint8, device=dev)
    keep = torch.empty((num_class, cap), dtype=torch.int64, device=dev)
    num_keep = torch.zeros((num_class,), dtype=torch.int32, device=dev)
    rc = L.fnp_nms_batched(_l.ptr(boxes), _l.ptr(counts), num_class, cap, float(_cfg(nms_config, "NMS_THRESH")), rotated, _l.ptr(ws), _l.ptr(keep),
                           _l.ptr(num_keep), _l.stream())
    _l.check(rc, "fnp_nms_batched")
    post = int(_cfg(nms_config, "NMS_POST_MAXSIZE"))
    kept = [min(int(v), post) for v in num_keep.tolist()]                     # the one host synchronisation
    sel = torch.cat([top_idx[k, keep[k, :m]] for k, m in enumerate(kept)]) if sum(kept) else top_idx.new_zeros((0,))
    labels = torch.repeat_interleave(torch.arange(num_class, device=dev), torch.tensor(kept, device=dev))
    return cls_scores[sel, labels], labels, box_preds[sel]
