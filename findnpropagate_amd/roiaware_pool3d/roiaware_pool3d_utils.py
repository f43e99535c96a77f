"""pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:9-41 with the same names and signatures."""
import numpy as np
import torch

from . import roiaware_pool3d_cuda


def _to_device(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float().cuda(), True
    return x, False


def points_in_boxes_cpu(points, boxes):
    """
    Args:
        points: (num_points, 3)
        boxes: [x, y, z, dx, dy, dz, heading], (x, y, z) is the box center, each box DO NOT overlaps
    Returns:
        point_indices: (N, num_points)
    The test (MARGIN 1e-2) is the reference's CPU one; it is evaluated on the GPU and returned in
    the container type it came in (numpy in -> numpy out, tensor in -> tensor on its device).
    """
    assert boxes.shape[1] == 7
    assert points.shape[1] == 3
    points_d, is_numpy = _to_device(points)
    boxes_d, _ = _to_device(boxes)
    src_device = None if is_numpy else points.device
    points_d = points_d.float().cuda().contiguous()
    boxes_d = boxes_d.float().cuda().contiguous()
    point_indices = torch.zeros((boxes_d.shape[0], points_d.shape[0]), dtype=torch.int, device=points_d.device)
    roiaware_pool3d_cuda.points_in_boxes_cpu(boxes_d, points_d, point_indices)
    if is_numpy:
        return point_indices.cpu().numpy()
    return point_indices.to(src_device)


def points_in_boxes_gpu(points, boxes):
    """
    :param points: (B, M, 3)
    :param boxes: (B, T, 7), num_valid_boxes <= T
    :return box_idxs_of_pts: (B, M), default background = -1
    """
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    batch_size, num_points, _ = points.shape
    box_idxs_of_pts = points.new_zeros((batch_size, num_points), dtype=torch.int).fill_(-1)
    roiaware_pool3d_cuda.points_in_boxes_gpu(boxes.contiguous(), points.contiguous(), box_idxs_of_pts)
    return box_idxs_of_pts


def points_in_boxes_count(points, boxes):
    """points (M,3), boxes (T,7) -> (T,) int32 point counts per box (GPU margin 1e-5): one launch
    for the Box Seeker's per-candidate loop (frustum_proposals_v1.py:930-932)."""
    assert boxes.shape[1] == 7 and points.shape[1] == 3
    counts = torch.zeros((boxes.shape[0],), dtype=torch.int, device=points.device)
    roiaware_pool3d_cuda.points_in_boxes_count(boxes.contiguous(), points.contiguous(), counts)
    return counts
