"""The recall bookkeeping of pcdet/models/detectors/detector3d_template.py:314-399, the part of
Detector3DTemplate that tools/extract_pseudo_labels.py:124 calls directly on the Box Seeker's output.

Same static-method signature, same dictionary keys and counts.  The reference synchronises ~6 times per
IoU threshold (`.item()` after every masked sum) plus once per ground-truth box for the known/unknown
masks; here one fused IoU launch per box set (`boxes_iou3d_gpu`) is followed by device-side reductions
into a single small counter vector that crosses to the host once."""
import torch

from ..iou3d_nms import iou3d_nms_utils

# detector3d_template.py:15-22
all_class_names = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer',
                   'barrier', 'motorcycle', 'bicycle', 'pedestrian', 'traffic_cone']
knowns3_names = ['car', 'bicycle', 'pedestrian']
knowns6_names = ['car', 'construction_vehicle', 'trailer', 'barrier', 'bicycle', 'pedestrian']
known3_labels = [all_class_names.index(x) + 1 for x in knowns3_names]
known6_labels = [all_class_names.index(x) + 1 for x in knowns6_names]


class Detector3DTemplate:
    """Only the static helpers that sit on the extraction path; the module plumbing (build_networks,
    post_processing, checkpoint loading) stays the reference's own."""

    @staticmethod
    def recall_counter_vector(box_preds, gt_boxes, thresh_list, rois=None, pred_count=None):
        """The counters one frame adds to the recall record (detector3d_template.py:342-397) as ONE device vector,
        no host synchronisation: [gt, num_3known, num_6known, num_4unknown, num_7unknown] then per threshold
        [roi, rcnn, rcnn_3known, rcnn_6known, rcnn_4unknown, rcnn_7unknown] (int64).  box_preds (K,7+); gt_boxes
        (G, >=8) with the class label last; pred_count: optional device scalar, rows >= it are padding."""
        dev = gt_boxes.device
        T = len(thresh_list)
        out = torch.zeros((5 + 6 * T,), dtype=torch.int64, device=dev)
        if gt_boxes.shape[0] == 0:
            return out
        # the all-zero padding rows at the end (:342-346) are masked out on the device instead of sliced off
        nonzero = (gt_boxes.sum(dim=1) != 0).to(torch.int32)
        valid = torch.flip(torch.cummax(torch.flip(nonzero, [0]), 0)[0], [0]).bool()    # row i: some non-zero row at or after i
        labels = gt_boxes[:, -1].long()
        known3 = torch.isin(labels, torch.tensor(known3_labels, device=dev)) & valid
        known6 = torch.isin(labels, torch.tensor(known6_labels, device=dev)) & valid
        unk3, unk6 = valid & ~known3, valid & ~known6
        th = torch.tensor([float(t) for t in thresh_list], dtype=torch.float32, device=dev)
        out[0:5] = torch.stack([valid.sum(), known3.sum(), known6.sum(), unk6.sum(), unk3.sum()])
        per = torch.zeros((T, 6), dtype=torch.int64, device=dev)
        gt7 = gt_boxes[:, 0:7].contiguous().float()
        if box_preds.shape[0] > 0:
            iou = iou3d_nms_utils.boxes_iou3d_gpu(box_preds[:, 0:7].contiguous().float(), gt7)
            if pred_count is not None:
                live = torch.arange(box_preds.shape[0], device=dev)[:, None] < pred_count.reshape(1, 1)
                iou = torch.where(live, iou, torch.zeros_like(iou))
            hit = (iou.max(dim=0)[0][None, :] > th[:, None]) & valid[None, :]    # (T, G)
            per[:, 1:6] = torch.stack([hit.sum(1), (hit & known3).sum(1), (hit & known6).sum(1), (hit & unk6).sum(1),
                                       (hit & unk3).sum(1)], dim=1)
        if rois is not None:
            iou_roi = iou3d_nms_utils.boxes_iou3d_gpu(rois[:, 0:7].contiguous().float(), gt7)
            per[:, 0] = ((iou_roi.max(dim=0)[0][None, :] > th[:, None]) & valid[None, :]).sum(1)
        out[5:] = per.reshape(-1)
        return out

    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        if 'gt_boxes' not in data_dict:
            return recall_dict
        rois = data_dict['rois'][batch_index] if 'rois' in data_dict else None
        gt_boxes = data_dict['gt_boxes'][batch_index]

        if recall_dict.__len__() == 0:
            recall_dict = {'gt': 0, 'num_3known': 0, 'num_6known': 0, 'num_4unknown': 0, 'num_7unknown': 0}
            for cur_thresh in thresh_list:
                for stem in ('roi_%s', 'rcnn_%s', 'rcnn_3known_%s', 'rcnn_6known_%s', 'rcnn_4unknown_%s', 'rcnn_7unknown_%s'):
                    recall_dict[stem % str(cur_thresh)] = 0
        if gt_boxes.shape[0] == 0:
            return recall_dict
        vec = Detector3DTemplate.recall_counter_vector(box_preds, gt_boxes, thresh_list, rois=rois).cpu().tolist()   # the one host sync
        keys = ['gt', 'num_3known', 'num_6known', 'num_4unknown', 'num_7unknown']
        for cur_thresh in thresh_list:
            keys += [stem % str(cur_thresh) for stem in ('roi_%s', 'rcnn_%s', 'rcnn_3known_%s', 'rcnn_6known_%s',
                                                          'rcnn_4unknown_%s', 'rcnn_7unknown_%s')]
        for k, v in zip(keys, vec):
            recall_dict[k] += v
        return recall_dict
