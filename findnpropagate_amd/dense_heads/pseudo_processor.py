"""PseudoProcessor (pcdet/models/dense_heads/pseudo_processor.py:110-400): the model-side half of
the pseudo-label mixing in the self-training step (tools/train_st.py).

  __call__            :374-400  relabel the known-class ground truth to the full 10-class ids and append the pseudo boxes
                                of the unknown classes: gt_boxes (B, G, C) + pseudo_boxes (B, P, 8) -> (B, <= G+P, C)
  save_predictions    :277-372  per frame: drop predictions that overlap pasted samples (rotated BEV IoU on the host),
                                undo the world augmentations, count the boxes consistent with last round's file,
                                write `<frame>.pth` = dict(pred_boxes, pred_scores, pred_labels, epoch)
  undo_augmentations  :241-275  translation, scaling, rotation, flip undone in reverse order

Same names, arguments, return values, on-disk format and statistics (`forward_pseudo_stats`).  The per-box Python loops
of the reference (`.item()` per ground-truth row, :171-184; per-box class counting, :22-29) are array operations here;
the rotated BEV IoU runs through the library's host entry point.  Two quirks of the reference are kept because the
files it writes depend on them: undoing the scaling divides the YAW by the scale too (:96), and only x, y, z — not
the box size — are rescaled.
"""
import os
from pathlib import Path

import numpy as np
import torch

from ..augmentor.pseudo_loader import ALL_CLASS_NAMES, rotate_points_along_z
from ..iou3d_nms import iou3d_nms_utils


def valid_boxes(cur_gt_bboxes_3d):
    """rows with dx > 0 and dy > 0 (:12-19)"""
    return cur_gt_bboxes_3d[(cur_gt_bboxes_3d[:, 3] > 0) & (cur_gt_bboxes_3d[:, 4] > 0)]


def count_classes(curr_boxes, curr_stats, all_class_names):
    if curr_boxes.shape[0]:
        for lbl, n in zip(*torch.unique(curr_boxes[..., -1].long() - 1, return_counts=True)):
            curr_stats[f'num_per_class_{all_class_names[int(lbl)]}'] += int(n)
    return curr_stats


def single_batch_apply(batch_dict, preds_dict, required_keys, apply_fn, b):
    assert all(k in batch_dict.keys() for k in required_keys)
    preds_dict['pred_boxes'] = apply_fn({k: batch_dict[k][b] for k in required_keys}, preds_dict['pred_boxes'])
    return preds_dict


class AugReverse:
    """Inverse of each world augmentation of the data augmentor, applied to predicted boxes (:56-108)."""

    @staticmethod
    def random_world_flip(data, preds_dict, b):
        def apply_fn(d, boxes):
            if d['flip_x']:
                boxes[..., 1] = -boxes[..., 1]
                boxes[..., 6] = -boxes[..., 6]
            if d['flip_y']:
                boxes[..., 0] = -boxes[..., 0]
                boxes[..., 6] = -(boxes[..., 6] + np.pi)
            return boxes
        return single_batch_apply(data, preds_dict, ['flip_x', 'flip_y'], apply_fn, b)

    @staticmethod
    def random_world_rotation(data, preds_dict, b):
        def apply_fn(d, boxes):
            rot = d['noise_rot'].to(boxes.device)
            boxes[:, 0:3] = rotate_points_along_z(boxes[None, :, 0:3], -rot[None])[0]
            boxes[:, 6] -= rot
            return boxes
        return single_batch_apply(data, preds_dict, ['noise_rot'], apply_fn, b)

    @staticmethod
    def random_world_scaling(data, preds_dict, b):
        def apply_fn(d, boxes):
            scale = d['noise_scale'].to(boxes.device)
            boxes[:, 0:3] /= scale
            boxes[:, 6] /= scale          # (sic, :96)
            return boxes
        return single_batch_apply(data, preds_dict, ['noise_scale'], apply_fn, b)

    @staticmethod
    def random_world_translation(data, preds_dict, b):
        def apply_fn(d, boxes):
            boxes[:, 0:3] -= d['noise_translate'].to(boxes.device)
            return boxes
        return single_batch_apply(data, preds_dict, ['noise_translate'], apply_fn, b)


class PseudoProcessor(object):
    sample_iou_thresh = 0.01
    point_cloud_range = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    cons_iou_thresh = 0.3

    def __init__(self, known_class_names, self_training_folder=None, all_class_names=None):
        self.all_class_names = list(ALL_CLASS_NAMES) if all_class_names is None else all_class_names
        self.known_class_names = known_class_names
        self.num_classes = len(self.all_class_names)
        self.self_training = self_training_folder is not None      # predictions are saved live
        self.is_known = {i: (c in known_class_names) for i, c in enumerate(self.all_class_names)}
        self.training = set(known_class_names) != set(self.all_class_names)   # open vocabulary: fewer classes in training
        # ground-truth labels are 1-based indices into the KNOWN class list
        self.gt_known_to_full_labels = {(i + 1): (j + 1) for i, kn in enumerate(known_class_names)
                                        for j, an in enumerate(self.all_class_names) if kn == an}
        self.full_labels_to_gt_known = {v: k for k, v in self.gt_known_to_full_labels.items()}
        self.unknown_labels = [i + 1 for i, c in enumerate(self.all_class_names) if c not in known_class_names]
        self.all_labels = [i + 1 for i in range(self.num_classes)]
        self.pseudos_missing = set()
        if self.self_training:
            self.self_training_folder = self_training_folder
            parent = Path(self_training_folder).parent
            assert os.path.exists(parent), f'self training folder parent should exist! {parent}'
            os.makedirs(self_training_folder, exist_ok=True)
        self.forward_pseudo_stats = {}
        # label -> label table of relabel_gt_boxes (identity for everything that is not a known-list index)
        self._relabel_max = max(list(self.gt_known_to_full_labels) + [0])

    def relabel_gt_boxes(self, gt_boxes):
        """Known-list class index -> index in the full 10-class list, in place (:162-183).  One table lookup over the
        label column instead of a `.item()` per ground-truth row (B x G host syncs on a device tensor)."""
        labels = gt_boxes[..., -1]
        as_int = labels.to(torch.int64)       # int(x.item()) truncates toward zero
        lut = torch.arange(self._relabel_max + 1, dtype=labels.dtype, device=labels.device)
        for k, v in self.gt_known_to_full_labels.items():
            lut[k] = v
        hit = (as_int >= 0) & (as_int <= self._relabel_max)
        hit &= torch.isin(as_int, torch.tensor(list(self.gt_known_to_full_labels), dtype=torch.int64, device=labels.device))
        gt_boxes[..., -1] = torch.where(hit, lut[as_int.clamp(0, self._relabel_max)], labels)
        return gt_boxes

    def combine_gt_with_pseudos(self, gt_boxes, pseudo_boxes):
        """gt (B, N, C) + pseudo (B, M, 8) -> (B, max_b(valid gt + valid pseudo), C): the pseudo box fills the first 7
        columns and its label the last (:185-239)."""
        stats = {'num_gt': 0, 'num_pseudo': 0}
        for c in self.all_class_names:
            stats[f'num_per_class_{c}'] = 0
        B = gt_boxes.shape[0]
        assert B == pseudo_boxes.shape[0], f'batch size should be the same, gt:{gt_boxes.shape} pseudo:{pseudo_boxes.shape}'
        out = torch.zeros((B, gt_boxes.shape[1] + pseudo_boxes.shape[1], gt_boxes.shape[-1]), device=gt_boxes.device)
        longest = 0
        for b in range(B):
            g, p = valid_boxes(gt_boxes[b]), valid_boxes(pseudo_boxes[b])
            ng, np_ = g.shape[0], p.shape[0]
            stats['num_gt'] += ng
            stats['num_pseudo'] += np_
            stats = count_classes(g, stats, self.all_class_names)
            stats = count_classes(p, stats, self.all_class_names)
            longest = max(longest, ng + np_)
            out[b, :ng] = g
            out[b, ng:ng + np_, :(p.shape[-1] - 1)] = p[..., :-1]
            out[b, ng:ng + np_, -1] = p[..., -1]
        for k in stats:
            self.forward_pseudo_stats[k] = stats[k] / max(B, 1)
        return out[:, :longest].contiguous()

    def undo_augmentations(self, batch_dict, preds_dict, b):
        keys = {'random_world_flip': ['flip_x', 'flip_y'], 'random_world_rotation': ['noise_rot'],
                'random_world_scaling': ['noise_scale'], 'random_world_translation': ['noise_translate']}
        for aug in ('random_world_translation', 'random_world_scaling', 'random_world_rotation', 'random_world_flip'):
            data = {k: batch_dict[k] for k in keys[aug] if k in batch_dict}
            if data:
                preds_dict = getattr(AugReverse, aug)(data, preds_dict, b)
        return preds_dict

    def save_predictions(self, batch_dict, preds_dicts, epoch=0):
        pseudo_boxes = batch_dict.get('pseudo_boxes', None)
        sample_mask = batch_dict.get('pseudo_samples_mask', None)
        pred_keys = ('pred_boxes', 'pred_scores', 'pred_labels')
        consistent = {l: 0.0 for l in self.all_labels}
        for b, (frame_id, preds_dict) in enumerate(zip(batch_dict['frame_id'], preds_dicts)):
            for k in pred_keys:
                if k in preds_dict:
                    preds_dict[k] = preds_dict[k].detach().clone().cpu()
            if pseudo_boxes is not None:
                sample_mask = sample_mask.to(dtype=torch.bool)
                pasted = pseudo_boxes[b][sample_mask[b]].cpu()
                if pasted.shape[0] > 0 and preds_dict['pred_boxes'].shape[0] > 0:
                    # predictions on top of copy-pasted objects are not pseudo labels of this frame
                    ious = iou3d_nms_utils.boxes_bev_iou_cpu(preds_dict['pred_boxes'][:, :7].contiguous(), pasted[:, :7].contiguous())
                    keep = ious.max(dim=1).values < self.sample_iou_thresh
                    for k in pred_keys:
                        if k in preds_dict:
                            preds_dict[k] = preds_dict[k][keep]
            preds_dict = self.undo_augmentations(batch_dict, preds_dict, b)
            path = Path(self.self_training_folder) / f"{frame_id.replace('.', '_')}.pth"
            if path.exists():       # how many of this round's boxes were already there last round, per class
                counts = {l: 0 for l in self.all_labels}
                try:
                    old = torch.load(path, map_location='cpu')
                    ov = iou3d_nms_utils.boxes_bev_iou_cpu(preds_dict['pred_boxes'][:, :7].contiguous(), old['pred_boxes'][:, :7].contiguous())
                    same = ov.max(dim=1).values >= self.cons_iou_thresh
                    for lbl in preds_dict['pred_labels'][same].tolist():
                        counts[int(lbl)] += 1
                except Exception as e:
                    print('Exception when trying to calculate consistency with hungarian assigner =>', e)
                for l in self.all_labels:
                    consistent[l] += counts[l]
            preds_dict['epoch'] = epoch
            torch.save(preds_dict, path)
        for l in self.all_labels:
            self.forward_pseudo_stats[f'mean_consistent_{self.all_class_names[int(l - 1)]}'] = consistent[l] / batch_dict['batch_size']

    def __call__(self, batch_dict):
        if not self.training:
            return batch_dict
        gt_boxes = self.relabel_gt_boxes(batch_dict['gt_boxes'])
        gt_boxes = self.combine_gt_with_pseudos(gt_boxes, batch_dict['pseudo_boxes'])
        if 'pseudo_samples_mask' in batch_dict:
            self.forward_pseudo_stats['mean_samples'] = batch_dict['pseudo_samples_mask'].sum(dim=1).mean()
        else:
            self.forward_pseudo_stats['mean_samples'] = 0
        batch_dict['gt_boxes'] = gt_boxes
        return batch_dict
