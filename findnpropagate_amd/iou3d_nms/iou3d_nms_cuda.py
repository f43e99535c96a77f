"""Module object with the seven pybind names of pcdet.ops.iou3d_nms.iou3d_nms_cuda
(iou3d_nms_api.cpp:11-19); target_assigner/hungarian_assigner.py:3,39 imports it directly.

Calling convention kept: caller-allocated outputs, `keep` is a CPU int64 tensor, return value is
1 or the number of boxes kept.  Inputs that are not device/contiguous raise (the reference
prints and exit(-1)s, iou3d_nms.cpp:14-26).
"""
import torch

from .. import lib as _l


def _check(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _l.FnpError("must be a device tensor")
        if not t.is_contiguous():
            raise _l.FnpError("must be contiguous tensor")


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _check(boxes_a, boxes_b, ans_overlap)
    rc = _l.load().fnp_boxes_overlap_bev(_l.ptr(boxes_a), boxes_a.shape[0], _l.ptr(boxes_b), boxes_b.shape[0],
                                         _l.ptr(ans_overlap), _l.stream())
    _l.check(rc, "fnp_boxes_overlap_bev")
    return 1


def boxes_aligned_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _check(boxes_a, boxes_b, ans_overlap)
    assert boxes_a.shape[0] == boxes_b.shape[0]
    rc = _l.load().fnp_boxes_aligned_overlap_bev(_l.ptr(boxes_a), _l.ptr(boxes_b), boxes_a.shape[0],
                                                 _l.ptr(ans_overlap), _l.stream())
    _l.check(rc, "fnp_boxes_aligned_overlap_bev")
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    _check(boxes_a, boxes_b, ans_iou)
    rc = _l.load().fnp_boxes_iou_bev(_l.ptr(boxes_a), boxes_a.shape[0], _l.ptr(boxes_b), boxes_b.shape[0],
                                     _l.ptr(ans_iou), _l.stream())
    _l.check(rc, "fnp_boxes_iou_bev")
    return 1


def boxes_iou3d_gpu(boxes_a, boxes_b, ans_iou):
    """Extension: fused iou3d_nms_utils.boxes_iou3d_gpu (one launch instead of ~12 torch ops)."""
    _check(boxes_a, boxes_b, ans_iou)
    rc = _l.load().fnp_boxes_iou3d(_l.ptr(boxes_a), boxes_a.shape[0], _l.ptr(boxes_b), boxes_b.shape[0],
                                   _l.ptr(ans_iou), _l.stream())
    _l.check(rc, "fnp_boxes_iou3d")
    return 1


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b, ans_iou):
    """Extension: fused iou3d_nms_utils.boxes_aligned_iou3d_gpu, ans_iou (N,1)."""
    _check(boxes_a, boxes_b, ans_iou)
    assert boxes_a.shape[0] == boxes_b.shape[0] == ans_iou.numel()
    rc = _l.load().fnp_boxes_aligned_iou3d(_l.ptr(boxes_a), _l.ptr(boxes_b), boxes_a.shape[0], _l.ptr(ans_iou), _l.stream())
    _l.check(rc, "fnp_boxes_aligned_iou3d")
    return 1


def _nms_device(boxes, thresh, rotated):
    """boxes sorted by score desc (device) -> (keep int64 device (N,), num_keep int32 device (1,))."""
    L = _l.load()
    n = boxes.shape[0]
    ws = torch.empty((max(int(L.fnp_nms_workspace_bytes(n)), 8),), dtype=torch.uint8, device=boxes.device)
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=boxes.device)
    num = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    fn = L.fnp_nms_rotated if rotated else L.fnp_nms_normal
    rc = fn(_l.ptr(boxes), n, float(thresh), _l.ptr(ws), _l.ptr(keep), _l.ptr(num), _l.stream())
    _l.check(rc, "fnp_nms")
    return keep, num


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """boxes (N,7) device, pre-sorted; keep (N,) CPU int64 — iou3d_nms.cpp:113-159."""
    _check(boxes)
    k, num = _nms_device(boxes, nms_overlap_thresh, True)
    n = int(num.item())
    keep[:n] = k[:n].cpu()
    return n


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:162-209."""
    _check(boxes)
    k, num = _nms_device(boxes, nms_overlap_thresh, False)
    n = int(num.item())
    keep[:n] = k[:n].cpu()
    return n


def _check_host(*ts):
    for t in ts:
        if t.is_cuda or t.dtype != torch.float32:
            raise _l.FnpError("must be a float32 host tensor")
        if not t.is_contiguous():
            raise _l.FnpError("must be contiguous tensor")


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:232-252: HOST tensors (N,7),(M,7) -> ans_iou (N,M), rotated BEV IoU (pseudo-label
    mixing in dataloader workers).  Runs the library's host entry point; no GPU is touched."""
    _check_host(boxes_a, boxes_b, ans_iou)
    assert ans_iou.shape[0] == boxes_a.shape[0] and ans_iou.shape[1] == boxes_b.shape[0]
    rc = _l.load().fnp_host_boxes_iou_bev(_l.ptr(boxes_a), boxes_a.shape[0], _l.ptr(boxes_b), boxes_b.shape[0],
                                          _l.ptr(ans_iou))
    _l.check(rc, "fnp_host_boxes_iou_bev")
    return 1


def boxes_aligned_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:254-272: HOST tensors (N,7),(N,7) -> ans_iou (N,1)."""
    _check_host(boxes_a, boxes_b, ans_iou)
    assert boxes_a.shape[0] == boxes_b.shape[0] and ans_iou.numel() == boxes_a.shape[0]
    rc = _l.load().fnp_host_boxes_aligned_iou_bev(_l.ptr(boxes_a), _l.ptr(boxes_b), boxes_a.shape[0], _l.ptr(ans_iou))
    _l.check(rc, "fnp_host_boxes_aligned_iou_bev")
    return 1
