"""The reference's own point-in-box kernel on the GPU (oracle/_ref/libref_pib_gpu.so: roiaware_pool3d_kernel.cu:16-36,313-336
compiled by hipcc, oracle/Makefile) as the checker of fnp_points_in_boxes, fnp_points_in_boxes_count and the Box Seeker's
per-candidate counts: same GPU, same ocml, same IEEE operations -> array_equal, face-grazing points included."""
import ctypes

import torch


def lib_or_none():
    from oracle import ref_loader
    return ref_loader.pib_gpu_lib()


def points_in_boxes(lib, boxes, pts):
    """boxes (B, T, 7), pts (B, M, 3) f32 cuda -> (B, M) int32: index of the first box that holds the point, or -1
    (what roiaware_pool3d_utils.points_in_boxes_gpu returns, :28-41)."""
    boxes, pts = boxes.float().contiguous(), pts.float().contiguous()
    B, T, M = boxes.shape[0], boxes.shape[1], pts.shape[1]
    out = torch.full((B, M), -1, dtype=torch.int32, device=pts.device)
    if B and M:
        rc = lib.ref_points_in_boxes(B, T, M, ctypes.c_void_p(boxes.data_ptr()), ctypes.c_void_p(pts.data_ptr()),
                                     ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    return out


def counts(lib, pts, boxes):
    """points of `pts` (m, 3) inside each of `boxes` (n, 7), one reference launch per box batched as B = n, T = 1 — the
    reference's per-candidate loop (frustum_proposals_v1.py:930-932: points_in_boxes_gpu(pts[None], box[None, None]) >= 0).sum())"""
    n, m = boxes.shape[0], pts.shape[0]
    if n == 0 or m == 0:
        return torch.zeros((n,), dtype=torch.int64, device=boxes.device)
    idx = points_in_boxes(lib, boxes.reshape(n, 1, 7), pts[None].expand(n, m, 3).contiguous())
    return (idx >= 0).sum(1)
