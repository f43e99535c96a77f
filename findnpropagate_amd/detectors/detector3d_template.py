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

        # the all-zero padding rows at the end (detector3d_template.py:342-346) are masked out on the device
        # instead of sliced off (slicing needs their count on the host: two more synchronisations per frame)
        if gt_boxes.shape[0] == 0:
            return recall_dict
        dev = gt_boxes.device
        nonzero = (gt_boxes.sum(dim=1) != 0).to(torch.int32)
        valid = torch.flip(torch.cummax(torch.flip(nonzero, [0]), 0)[0], [0]).bool()    # row i: some non-zero row at or after i
        cur_gt = gt_boxes
        labels = cur_gt[:, -1].long()
        known3 = torch.isin(labels, torch.tensor(known3_labels, device=dev)) & valid
        known6 = torch.isin(labels, torch.tensor(known6_labels, device=dev)) & valid
        unk3, unk6 = valid & ~known3, valid & ~known6
        th = torch.tensor([float(t) for t in thresh_list], dtype=torch.float32, device=dev)

        counters = [valid.sum(), known3.sum(), known6.sum(), unk3.sum(), unk6.sum()]
        if box_preds.shape[0] > 0:
            iou3d_rcnn = iou3d_nms_utils.boxes_iou3d_gpu(box_preds[:, 0:7].contiguous().float(), cur_gt[:, 0:7].contiguous().float())
            hit = (iou3d_rcnn.max(dim=0)[0][None, :] > th[:, None]) & valid[None, :]    # (T, G)
            counters += [hit.sum(1), (hit & known3).sum(1), (hit & known6).sum(1), (hit & unk3).sum(1), (hit & unk6).sum(1)]
        if rois is not None:
            iou3d_roi = iou3d_nms_utils.boxes_iou3d_gpu(rois[:, 0:7].contiguous().float(), cur_gt[:, 0:7].contiguous().float())
            counters.append(((iou3d_roi.max(dim=0)[0][None, :] > th[:, None]) & valid[None, :]).sum(1))
        flat = torch.cat([c.reshape(-1).long() for c in counters]).cpu().tolist()   # the one host sync
        n_gt, flat = flat[0], flat[1:]
        if n_gt == 0:
            return recall_dict

        recall_dict['num_3known'] += flat[0]
        recall_dict['num_6known'] += flat[1]
        recall_dict['num_7unknown'] += flat[2]
        recall_dict['num_4unknown'] += flat[3]
        pos, T = 4, len(thresh_list)
        if box_preds.shape[0] > 0:
            for name in ('rcnn_%s', 'rcnn_3known_%s', 'rcnn_6known_%s', 'rcnn_7unknown_%s', 'rcnn_4unknown_%s'):
                for i, cur_thresh in enumerate(thresh_list):
                    recall_dict[name % str(cur_thresh)] += flat[pos + i]
                pos += T
        if rois is not None:
            for i, cur_thresh in enumerate(thresh_list):
                recall_dict['roi_%s' % str(cur_thresh)] += flat[pos + i]
        recall_dict['gt'] += n_gt
        return recall_dict
