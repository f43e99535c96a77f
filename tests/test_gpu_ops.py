"""Parity of the HIP operators (through the C ABI and the reference-named python wrappers) with
the CPU oracle and, where it could be built, with the reference's own kernels compiled by hipcc
(oracle/_ref/libref_iou3d_gpu.so).  Integer outputs must be bit-exact."""
import ctypes

import numpy as np
import pytest
import torch

from findnpropagate_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _boundary_mask(pts, boxes, margin, eps=2e-6):
    """(T,M) pairs whose decision is within float-rounding distance of a face: the only pairs where
    libm (CPU oracle) and ocml (GPU) cos/sin ulps may legitimately flip the flag."""
    out = np.zeros((boxes.shape[0], pts.shape[0]), bool)
    for t, b in enumerate(boxes.astype(np.float64)):
        d = pts.astype(np.float64) - b[:3]
        c, s = np.cos(-b[6]), np.sin(-b[6])
        lx, ly = d[:, 0] * c - d[:, 1] * s, d[:, 0] * s + d[:, 1] * c
        scale = np.abs(d[:, :2]).sum(1) + 1.0
        out[t] = (np.abs(np.abs(lx) - (b[3] / 2 + margin)) < eps * scale) | (np.abs(np.abs(ly) - (b[4] / 2 + margin)) < eps * scale)
    return out


def test_points_in_boxes_gpu_bit_exact(cuda, oracle, rng):
    from findnpropagate_amd.roiaware_pool3d import roiaware_pool3d_utils as U

    B, T, M = 3, 37, 20011   # ragged sizes: M not a multiple of 256, T not of the LDS tile
    boxes = np.stack([syn.random_boxes(rng, T, 8.0) for _ in range(B)])
    pts = rng.uniform(-10, 10, size=(B, M, 3)).astype(np.float32)
    pts[..., 2] = rng.uniform(-4, 3, size=(B, M))
    boxes[:, 0, 6] = 0.0     # exact trig on box 0: face-touching points are decided identically
    for b in range(B):
        bx = boxes[b, 0]
        pts[b, 0] = [bx[0] + bx[3] / 2, bx[1], bx[2]]
        pts[b, 1] = [bx[0], bx[1], bx[2] + bx[5] / 2]
    d_pts, d_boxes = torch.from_numpy(pts).to(cuda), torch.from_numpy(boxes).to(cuda)
    got = U.points_in_boxes_gpu(d_pts, d_boxes).cpu().numpy()
    assert (got >= 0).sum() > 100 and got.dtype == np.int32
    import ref_pib
    ref = ref_pib.lib_or_none()
    if ref is not None:
        # the reference's OWN kernel on this GPU (roiaware_pool3d_kernel.cu:16-36,313-336 compiled by hipcc): bit-exact, no allowance
        assert np.array_equal(got, ref_pib.points_in_boxes(ref, d_boxes, d_pts).cpu().numpy())
        return
    # without it (oracle/_ref not built): the libm oracle, where only face-grazing pairs may differ (ocml vs glibc cos / sin)
    want = oracle.points_in_boxes(pts, boxes)
    diff = got != want
    if diff.any():
        for b in range(B):
            near = _boundary_mask(pts[b], boxes[b], 1e-5).any(0)
            assert not (diff[b] & ~near).any()
    assert diff.sum() <= 2


def test_reference_point_in_box_kernel_is_the_checker_when_the_reference_build_travelled(cuda):
    """oracle/_ref is built in the container (where /root/reference is) and travels with the snapshot: when the reference's iou3d
    build is there, its point-in-box build must be too — otherwise the array_equal checks above silently fall back to the libm
    oracle and its face-grazing allowance."""
    import os
    import ref_pib
    ref_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    if not os.path.exists(os.path.join(ref_dir, "libref_iou3d_gpu.so")):
        pytest.skip("oracle/_ref not built")
    lib = ref_pib.lib_or_none()
    assert lib is not None, "oracle/_ref/libref_pib_gpu.so missing: run `make -C oracle ref` where /root/reference is"
    boxes = torch.tensor([[[0.0, 0.0, 0.0, 4.0, 2.0, 1.5, 0.3]]], device=cuda)
    pts = torch.tensor([[[0.5, 0.2, 0.0], [3.0, 0.0, 0.0], [0.0, 0.0, 0.76]]], device=cuda)
    assert ref_pib.points_in_boxes(lib, boxes, pts).cpu().tolist() == [[0, -1, -1]]


def test_points_in_boxes_empty_and_large_T(cuda, oracle, rng):
    from findnpropagate_amd.roiaware_pool3d import roiaware_pool3d_utils as U

    pts = torch.zeros((1, 0, 3), device=cuda)
    assert U.points_in_boxes_gpu(pts, torch.zeros((1, 4, 7), device=cuda)).shape == (1, 0)
    pts = rng.uniform(-5, 5, size=(1, 1000, 3)).astype(np.float32)
    got = U.points_in_boxes_gpu(torch.from_numpy(pts).to(cuda), torch.zeros((1, 0, 7), device=cuda)).cpu().numpy()
    assert (got == -1).all()
    boxes = syn.random_boxes(rng, 300, 5.0)[None]   # more boxes than one LDS tile
    got = U.points_in_boxes_gpu(torch.from_numpy(pts).to(cuda), torch.from_numpy(boxes).to(cuda)).cpu().numpy()
    assert np.array_equal(got, oracle.points_in_boxes(pts, boxes))


def test_points_in_boxes_count_and_dense(cuda, oracle, rng):
    from findnpropagate_amd.roiaware_pool3d import roiaware_pool3d_utils as U

    boxes = syn.random_boxes(rng, 60, 4.0)
    boxes[:, 3:6] *= 1.5
    pts = rng.uniform(-8, 8, size=(3001, 3)).astype(np.float32)
    cnt = U.points_in_boxes_count(torch.from_numpy(pts).to(cuda), torch.from_numpy(boxes).to(cuda)).cpu().numpy()
    assert cnt.sum() > 0
    import ref_pib
    ref = ref_pib.lib_or_none()
    if ref is not None:   # the reference kernel, one box per launch like the Box Seeker's loop (:930-932): equal counts, no allowance
        assert np.array_equal(cnt, ref_pib.counts(ref, torch.from_numpy(pts).to(cuda), torch.from_numpy(boxes).to(cuda)).cpu().numpy())
    else:                 # libm oracle: a face-grazing point may flip
        want = oracle.points_in_boxes_count(pts, boxes)
        assert np.abs(cnt - want).max() <= 1 and (cnt != want).sum() <= 1
    # per-candidate loop of the reference (one launch each) gives the same counts
    d_pts, d_boxes = torch.from_numpy(pts).to(cuda), torch.from_numpy(boxes).to(cuda)
    loop = [int((U.points_in_boxes_gpu(d_pts[None], d_boxes[[i]][None]) >= 0).sum()) for i in range(0, 60, 7)]
    assert loop == cnt[0:60:7].tolist()
    dense = U.points_in_boxes_cpu(d_pts, d_boxes).cpu().numpy()
    wd = oracle.points_in_boxes_dense(pts, boxes)
    assert (dense != wd).sum() <= 1
    assert U.points_in_boxes_cpu(pts, boxes).shape == (60, 3001)   # numpy in -> numpy out


def test_cpu_tensor_is_rejected_not_silently_computed(cuda):
    from findnpropagate_amd.lib import FnpError
    from findnpropagate_amd.roiaware_pool3d import roiaware_pool3d_cuda as C

    with pytest.raises(FnpError):
        C.points_in_boxes_gpu(torch.zeros((1, 1, 7)), torch.zeros((1, 1, 3)), torch.zeros((1, 1), dtype=torch.int32))


def _ref_gpu():
    from oracle import ref_loader

    lib = ref_loader.iou3d_gpu_lib()
    if lib is None:
        pytest.skip("oracle/_ref/libref_iou3d_gpu.so not built")
    return lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_rotated_iou_bit_exact_vs_reference_kernels(cuda, oracle, rng):
    """Same GPU, same ocml, same IEEE op sequence -> bit-identical to the reference's kernels."""
    from findnpropagate_amd.iou3d_nms import iou3d_nms_cuda as C, iou3d_nms_utils as U

    ref = _ref_gpu()
    A = syn.random_boxes(rng, 83, 6.0)
    Bx = syn.random_boxes(rng, 57, 6.0)
    A[:5] = Bx[:5]                                   # identical boxes
    A[5, :] = [0, 0, 0, 4, 2, 1, 0.0]
    Bx[5, :] = [4.0, 0, 0, 4, 2, 1, 0.0]             # touching edge
    Bx[6, :] = [0.5, -0.3, 0, 1.5, 0.7, 1, 1.1]
    A[6, :] = [0, 0, 0, 10, 10, 1, 0.2]              # nested
    A[7, :] = [0, 0, 0, 4, 2, 1, 0.0]
    Bx[7, :] = [0, 0, 0, 4, 2, 1, np.pi / 2]         # 90 degrees
    A[8, 3:5] = 1e-3                                  # tiny
    a, b = torch.from_numpy(A).to(cuda), torch.from_numpy(Bx).to(cuda)
    ov = torch.zeros((83, 57), device=cuda)
    C.boxes_overlap_bev_gpu(a, b, ov)
    ov_ref = torch.zeros_like(ov)
    ref.ref_boxes_overlap(83, _p(a), 57, _p(b), _p(ov_ref))
    torch.cuda.synchronize()
    assert torch.equal(ov, ov_ref), f"max diff {(ov - ov_ref).abs().max().item()}"
    iou = U.boxes_iou_bev(a, b)
    iou_ref = torch.zeros_like(iou)
    ref.ref_boxes_iou_bev(83, _p(a), 57, _p(b), _p(iou_ref))
    torch.cuda.synchronize()
    assert torch.equal(iou, iou_ref)
    n = 57
    al = torch.zeros((n, 1), device=cuda)
    C.boxes_aligned_overlap_bev_gpu(a[:n].contiguous(), b, al)
    al_ref = torch.zeros_like(al)
    ref.ref_boxes_aligned_overlap(n, _p(a[:n].contiguous()), _p(b), _p(al_ref))
    torch.cuda.synchronize()
    assert torch.equal(al, al_ref)
    # and within float tolerance of the CPU oracle (libm vs ocml trig)
    np.testing.assert_allclose(ov.cpu().numpy(), oracle.boxes_overlap_bev(A, Bx), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(iou.cpu().numpy(), oracle.boxes_iou_bev(A, Bx), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(U.boxes_iou3d_gpu(a, b).cpu().numpy(), oracle.boxes_iou3d(A, Bx), rtol=1e-4, atol=1e-5)


def _sweep(mask, n):
    """Host greedy sweep of the reference (iou3d_nms.cpp:139-155) over its u64 mask."""
    cb = (n + 63) // 64
    remv = [0] * cb
    keep = []
    for i in range(n):
        if not (remv[i // 64] >> (i % 64)) & 1:
            keep.append(i)
            for j in range(i // 64, cb):
                remv[j] |= int(mask[i * cb + j])
    return keep


@pytest.mark.parametrize("n", [1, 60, 64, 65, 200, 700])
@pytest.mark.parametrize("rotated", [False, True])
def test_nms_keep_lists_bit_exact(cuda, oracle, rng, n, rotated):
    from findnpropagate_amd.iou3d_nms import iou3d_nms_cuda as C, iou3d_nms_utils as U

    ref = _ref_gpu()
    boxes = syn.random_boxes(rng, n, 6.0 if n < 300 else 14.0)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    d_boxes, d_scores = torch.from_numpy(boxes).to(cuda), torch.from_numpy(scores).to(cuda)
    order = torch.sort(d_scores, descending=True)[1]
    sb = d_boxes[order].contiguous()
    for thresh in (0.1, 0.45, 1.0):
        keep, num = C._nms_device(sb, thresh, rotated)
        got = keep[: int(num.item())].cpu().tolist()
        cb = (n + 63) // 64
        mask = torch.zeros((n * cb,), dtype=torch.int64, device=cuda)
        (ref.ref_nms_mask if rotated else ref.ref_nms_normal_mask)(_p(sb), _p(mask), n, ctypes.c_float(thresh))
        torch.cuda.synchronize()
        m = mask.cpu().numpy().astype(np.uint64)
        assert got == _sweep(m, n), "keep list differs from the reference kernel + reference sweep"
        want = oracle.nms(sb.cpu().numpy(), thresh, rotated).tolist()
        assert got == want, "keep list differs from the CPU oracle"
        # pybind-shaped entry point: CPU int64 keep + count
        k_cpu = torch.zeros((n,), dtype=torch.int64)
        cnt = (C.nms_gpu if rotated else C.nms_normal_gpu)(sb, k_cpu, thresh)
        assert k_cpu[:cnt].tolist() == got
        # wrapper with the reference's name/signature
        sel, _ = (U.nms_gpu if rotated else U.nms_normal_gpu)(d_boxes, d_scores, thresh)
        assert sel.cpu().tolist() == order[torch.tensor(got, dtype=torch.long, device=cuda)].cpu().tolist()
        if thresh == 1.0:
            assert len(got) == n


def test_nms_empty(cuda):
    from findnpropagate_amd.iou3d_nms import iou3d_nms_utils as U

    sel, _ = U.nms_normal_gpu(torch.zeros((0, 7), device=cuda), torch.zeros((0,), device=cuda), 0.5)
    assert sel.numel() == 0
