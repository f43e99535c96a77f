"""pcdet/ops/iou3d_nms/iou3d_nms_utils.py:12-152 with the same names, arguments and returns."""
import torch

from . import iou3d_nms_cuda


def boxes_iou_bev(boxes_a, boxes_b):
    """boxes (N,7),(M,7) [x,y,z,dx,dy,dz,heading] -> (N,M) rotated BEV IoU."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) 3D IoU (z = box centre), iou3d_nms_utils.py:48-81, one fused launch."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou3d_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans)
    return ans


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(N,7) -> (N,1) 3D IoU of pair i (iou3d_nms_utils.py:83-117), one fused launch."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans = torch.zeros((boxes_a.shape[0], 1), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_aligned_iou3d_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans)
    return ans


def _nms(boxes, scores, thresh, rotated, pre_maxsize=None):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    order = order.to(boxes.device)
    boxes = boxes[order].contiguous()
    keep, num = iou3d_nms_cuda._nms_device(boxes, thresh, rotated)
    n = int(num.item())  # the reference synchronises here too (cudaMemcpy D2H, iou3d_nms.cpp:131)
    return order[keep[:n]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """(N,7) boxes, (N,) scores -> (kept indices in score order, None), iou3d_nms_utils.py:120-135."""
    return _nms(boxes, scores, thresh, True, pre_maxsize)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned BEV NMS (heading ignored), iou3d_nms_utils.py:138-152."""
    return _nms(boxes, scores, thresh, False)


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:12-28: (N,7),(M,7) host tensors or numpy arrays -> (N,M) rotated BEV IoU of
    the same kind (numpy in -> numpy out).  Device tensors are refused like in the reference."""
    import numpy as np
    is_numpy = isinstance(boxes_a, np.ndarray)
    if isinstance(boxes_a, np.ndarray):
        boxes_a = torch.from_numpy(boxes_a).float()
    if isinstance(boxes_b, np.ndarray):
        boxes_b = torch.from_numpy(boxes_b).float()
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    assert boxes_a.shape[1] == 7 and boxes_b.shape[1] == 7
    ans_iou = boxes_a.new_zeros(torch.Size((boxes_a.shape[0], boxes_b.shape[0])))
    iou3d_nms_cuda.boxes_iou_bev_cpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou.numpy() if is_numpy else ans_iou
