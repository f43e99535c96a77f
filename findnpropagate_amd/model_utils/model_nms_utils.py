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
    (pred_scores, pred_labels (0-based class column), pred_boxes), classes concatenated in order."""
    pred_scores, pred_labels, pred_boxes = [], [], []
    for k in range(cls_scores.shape[1]):
        if score_thresh is not None:
            scores_mask = cls_scores[:, k] >= score_thresh
            box_scores = cls_scores[scores_mask, k]
            cur_box_preds = box_preds[scores_mask]
        else:
            box_scores = cls_scores[:, k]
            cur_box_preds = box_preds
        selected = _select(box_scores, cur_box_preds, nms_config)
        pred_scores.append(box_scores[selected])
        pred_labels.append(box_scores.new_ones(len(selected)).long() * k)
        pred_boxes.append(cur_box_preds[selected])
    return torch.cat(pred_scores, dim=0), torch.cat(pred_labels, dim=0), torch.cat(pred_boxes, dim=0)
