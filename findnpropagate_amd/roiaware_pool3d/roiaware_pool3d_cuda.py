"""Module object with the pybind names of pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda
(roiaware_pool3d.cpp:171-177), calling libfnp_hip.so through the C ABI.

Same calling convention as the extension: the caller allocates the output tensor, the function
fills it and returns 1.  Errors raise FnpError instead of exit(-1)
(roiaware_pool3d_kernel.cu:350-354).
"""
import torch

from .. import lib as _l


def points_in_boxes_gpu(boxes_tensor, pts_tensor, box_idx_of_points_tensor):
    """boxes (B,T,7), pts (B,M,3), box_idx_of_points (B,M) int32 — roiaware_pool3d.cpp:98-118."""
    L = _l.load()
    _l.require_device(boxes_tensor, pts_tensor, box_idx_of_points_tensor)
    assert boxes_tensor.dtype == torch.float32 and pts_tensor.dtype == torch.float32
    assert box_idx_of_points_tensor.dtype == torch.int32
    B, T = boxes_tensor.shape[0], boxes_tensor.shape[1]
    M = pts_tensor.shape[1]
    rc = L.fnp_points_in_boxes(_l.ptr(boxes_tensor), _l.ptr(pts_tensor), _l.ptr(box_idx_of_points_tensor), B, T, M,
                               _l.stream())
    _l.check(rc, "fnp_points_in_boxes")
    return 1


def points_in_boxes_cpu(boxes_tensor, pts_tensor, pts_indices_tensor):
    """boxes (T,7), pts (M,3), pts_indices (T,M) int32, MARGIN 1e-2 — roiaware_pool3d.cpp:143-168.
    The reference runs this on the host; here the tensors must live on the device."""
    L = _l.load()
    _l.require_device(boxes_tensor, pts_tensor, pts_indices_tensor)
    T, M = boxes_tensor.shape[0], pts_tensor.shape[0]
    rc = L.fnp_points_in_boxes_dense(_l.ptr(boxes_tensor), _l.ptr(pts_tensor), _l.ptr(pts_indices_tensor), T, M,
                                     _l.stream())
    _l.check(rc, "fnp_points_in_boxes_dense")
    return 1


def points_in_boxes_count(boxes_tensor, pts_tensor, counts_tensor):
    """Extension (not in the reference module): boxes (T,7), pts (M,3) -> counts (T,) int32, the
    batched form of frustum_proposals_v1.py:930-932."""
    L = _l.load()
    _l.require_device(boxes_tensor, pts_tensor, counts_tensor)
    rc = L.fnp_points_in_boxes_count(_l.ptr(boxes_tensor), _l.ptr(pts_tensor), _l.ptr(counts_tensor),
                                     boxes_tensor.shape[0], pts_tensor.shape[0], _l.stream())
    _l.check(rc, "fnp_points_in_boxes_count")
    return 1


def forward(*args, **kwargs):
    raise NotImplementedError("RoI-aware pooling (Part-A2) is outside the hot path (SURVEY.md §2.2)")


def backward(*args, **kwargs):
    raise NotImplementedError("RoI-aware pooling (Part-A2) is outside the hot path (SURVEY.md §2.2)")
