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
    def recall_counter_vector(box_preds, gt_boxes, thresh_list, rois=None, pred_count=None, out=None):
        """The counters one frame adds to the recall record (detector3d_template.py:342-397) as ONE device vector and
        ONE launch (fnp_recall_counters), no host synchronisation: [gt, num_3known, num_6known, num_4unknown,
        num_7unknown] then per threshold [roi, rcnn, rcnn_3known, rcnn_6known, rcnn_4unknown, rcnn_7unknown] (int64).
        box_preds (K, >=7) rows may be strided views (e.g. the body of an extraction record); gt_boxes (G, >=8) with
        the class label last; pred_count: optional device f32 scalar, rows >= it are padding; out: vector to add into."""
        import ctypes

        from .. import lib as _l

        L = _l.load()
        dev = gt_boxes.device
        T = len(thresh_list)
        if out is None:
            out = torch.zeros((5 + 6 * T,), dtype=torch.int64, device=dev)
        if gt_boxes.shape[0] == 0:
            return out
        _l.require_device(gt_boxes)
        gt = gt_boxes if (gt_boxes.dtype == torch.float32 and gt_boxes.is_contiguous()) else gt_boxes.float().contiguous()

        def rows(t):
            """(pointer, row stride in floats, rows) of a 2-D f32 tensor whose rows are contiguous"""
            if t is None or t.shape[0] == 0:
                return None, 7, 0
            if t.dtype != torch.float32 or t.stride(1) != 1:
                t = t.float().contiguous()
            keep.append(t)
            return t.data_ptr(), t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 7), t.shape[0]

        keep = []
        p_ptr, p_stride, p_n = rows(box_preds)
        r_ptr, r_stride, r_n = rows(rois)
        th = (ctypes.c_float * T)(*[float(t) for t in thresh_list])
        bits = lambda labels: sum(1 << int(l) for l in labels)
        rc = L.fnp_recall_counters(p_ptr, p_stride, p_n, None if pred_count is None else pred_count.data_ptr(),
                                   gt.data_ptr(), gt.shape[0], gt.shape[1], r_ptr, r_n, r_stride,
                                   ctypes.cast(th, ctypes.c_void_p), T, bits(known3_labels), bits(known6_labels),
                                   out.data_ptr(), _l.stream())
        _l.check(rc, "fnp_recall_counters")
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
