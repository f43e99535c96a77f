"""Pins the CPU oracle's operator restatements (oracle/fnp_oracle.c) against the reference's own
code (oracle/_ref, built from /root/reference by oracle/Makefile), known-answer values and
independent numpy restatements.  CPU only."""
import numpy as np
import pytest
import torch

from findnpropagate_amd import synthetic as syn


def test_points_in_boxes_dense_matches_reference_cpp(oracle, rng):
    """oracle vs the reference's roiaware_pool3d.cpp:143-168 compiled as is."""
    from oracle import ref_loader

    ref = ref_loader.roiaware_cpu_module()
    if ref is None:
        pytest.skip("oracle/_ref/roiaware_pool3d_ref.so not built")
    for trial in range(4):
        boxes = syn.random_boxes(rng, 17, centre_range=6.0)
        pts = rng.uniform(-9, 9, size=(20000, 3)).astype(np.float32)
        pts[:, 2] = rng.uniform(-4, 4, size=20000)
        # points exactly on / next to the faces of box 0 (heading 0 -> exact trig)
        boxes[0, 6] = 0.0
        b = boxes[0]
        edge = np.array([[b[0] + b[3] / 2 + d, b[1], b[2]] for d in (-1e-2, 0.0, 9e-3, 1e-2, 1.1e-2)], np.float32)
        pts[:5] = edge
        out = torch.zeros((boxes.shape[0], pts.shape[0]), dtype=torch.int32)
        ref.points_in_boxes_cpu(torch.from_numpy(boxes), torch.from_numpy(pts), out)
        mine = oracle.points_in_boxes_dense(pts, boxes)
        assert np.array_equal(out.numpy(), mine)
        assert mine.sum() > 0


def test_points_in_boxes_gpu_variant_margin_and_first_box(oracle):
    """GPU variant (roiaware_pool3d_kernel.cu:23-36): MARGIN 1e-5, strict z test, first box wins."""
    boxes = np.array([[[0, 0, 0, 4, 2, 2, 0.0], [0, 0, 0, 8, 8, 8, 0.0]]], np.float32)
    pts = np.array([[
        [0, 0, 0],            # in both -> box 0
        [2.0, 0, 0],          # |lx| = dx/2 < dx/2 + 1e-5 -> box 0
        [2.0 + 2e-5, 0, 0],   # outside box 0 margin, inside box 1
        [0, 0, 1.0],          # |z| = dz/2 is NOT > dz/2 -> box 0
        [0, 0, 1.0001],       # above box 0, inside box 1
        [9, 9, 9],            # nowhere
    ]], np.float32)
    assert oracle.points_in_boxes(pts, boxes).tolist() == [[0, 0, 1, 0, 1, -1]]
    assert oracle.points_in_boxes_count(pts[0], boxes[0]).tolist() == [3, 5]


def test_points_in_boxes_count_matches_numpy(oracle, rng):
    boxes = syn.random_boxes(rng, 30, centre_range=5.0)
    pts = rng.uniform(-8, 8, size=(4000, 3)).astype(np.float32)
    cnt = oracle.points_in_boxes_count(pts, boxes)
    # independent double-precision restatement; margins are 1e-5 so ties are practically absent
    for t in range(boxes.shape[0]):
        b = boxes[t].astype(np.float64)
        d = pts.astype(np.float64) - b[:3]
        c, s = np.cos(-b[6]), np.sin(-b[6])
        lx, ly = d[:, 0] * c - d[:, 1] * s, d[:, 0] * s + d[:, 1] * c
        inside = (np.abs(d[:, 2]) <= b[5] / 2) & (np.abs(lx) < b[3] / 2 + 1e-5) & (np.abs(ly) < b[4] / 2 + 1e-5)
        assert abs(int(inside.sum()) - int(cnt[t])) <= 1


def test_rotated_overlap_known_answers(oracle):
    """Values printed by the reference's own box_overlap / iou_bev / iou_normal compiled during the
    survey."""
    a = np.array([[0, 0, 0, 4, 2, 1.5, 0.3]], np.float32)
    b = np.array([[0.5, 0.2, 0, 4, 2, 1.5, -0.2]], np.float32)
    assert oracle.boxes_overlap_bev(a, b)[0, 0] == pytest.approx(5.68089008, rel=2e-6)
    assert oracle.boxes_iou_bev(a, b)[0, 0] == pytest.approx(0.550521314, rel=2e-6)
    assert oracle.iou_normal(a[0], b[0]) == pytest.approx(0.649484456, rel=2e-6)


def test_rotated_overlap_geometry(oracle, rng):
    sq = np.array([[0, 0, 0, 2, 2, 1, 0.0]], np.float32)
    assert oracle.boxes_overlap_bev(sq, sq)[0, 0] == pytest.approx(4.0, rel=1e-5)
    far = np.array([[10, 10, 0, 2, 2, 1, 0.7]], np.float32)
    assert oracle.boxes_overlap_bev(sq, far)[0, 0] == 0.0
    # nested: small rotated box inside a large one -> overlap = small area
    big = np.array([[0, 0, 0, 10, 10, 1, 0.2]], np.float32)
    small = np.array([[0.5, -0.3, 0, 1.5, 0.7, 1, 1.1]], np.float32)
    assert oracle.boxes_overlap_bev(big, small)[0, 0] == pytest.approx(1.5 * 0.7, rel=1e-4)
    # 90 degree rotation of a 4x2 box about its centre over itself: overlap 2x2
    r = np.array([[0, 0, 0, 4, 2, 1, 0.0]], np.float32)
    r90 = np.array([[0, 0, 0, 4, 2, 1, np.pi / 2]], np.float32)
    assert oracle.boxes_overlap_bev(r, r90)[0, 0] == pytest.approx(4.0, rel=1e-4)
    # against a polygon-clipping restatement on random pairs (loose: the reference's 1e-2 corner margin)
    A, B = syn.random_boxes(rng, 40, 6.0), syn.random_boxes(rng, 40, 6.0)
    ov = oracle.boxes_overlap_bev(A, B)
    for i in range(0, 40, 3):
        for j in range(0, 40, 3):
            assert ov[i, j] == pytest.approx(_clip_area(A[i], B[j]), abs=0.15)
    al = oracle.boxes_aligned_overlap_bev(A, B)
    assert np.array_equal(al, np.diag(ov))


def _corners(b):
    c, s = np.cos(b[6]), np.sin(b[6])
    hx, hy = b[3] / 2, b[4] / 2
    loc = np.array([[-hx, -hy], [hx, -hy], [hx, hy], [-hx, hy]], np.float64)
    R = np.array([[c, -s], [s, c]])
    return loc @ R.T + b[:2].astype(np.float64)


def _clip_area(a, b):
    """Sutherland-Hodgman clip of rectangle a by rectangle b (float64)."""
    poly = _corners(a).tolist()
    clip = _corners(b)
    for i in range(4):
        p0, p1 = clip[i], clip[(i + 1) % 4]
        e = p1 - p0
        out = []
        for k in range(len(poly)):
            q0, q1 = np.array(poly[k]), np.array(poly[(k + 1) % len(poly)])
            s0 = e[0] * (q0[1] - p0[1]) - e[1] * (q0[0] - p0[0])
            s1 = e[0] * (q1[1] - p0[1]) - e[1] * (q1[0] - p0[0])
            if s0 >= 0:
                out.append(q0.tolist())
            if (s0 >= 0) != (s1 >= 0):
                t = s0 / (s0 - s1)
                out.append((q0 + t * (q1 - q0)).tolist())
        poly = out
        if not poly:
            return 0.0
    p = np.array(poly)
    return 0.5 * abs(np.dot(p[:, 0], np.roll(p[:, 1], -1)) - np.dot(p[:, 1], np.roll(p[:, 0], -1)))


def test_iou3d_matches_torch_formula(oracle, rng):
    """orc_boxes_iou3d vs the torch expression of iou3d_nms_utils.py:48-81 fed with oracle overlaps."""
    A, B = syn.random_boxes(rng, 25, 5.0), syn.random_boxes(rng, 31, 5.0)
    bev = torch.from_numpy(oracle.boxes_overlap_bev(A, B))
    a, b = torch.from_numpy(A), torch.from_numpy(B)
    a_max, a_min = (a[:, 2] + a[:, 5] / 2).view(-1, 1), (a[:, 2] - a[:, 5] / 2).view(-1, 1)
    b_max, b_min = (b[:, 2] + b[:, 5] / 2).view(1, -1), (b[:, 2] - b[:, 5] / 2).view(1, -1)
    oh = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    o3 = bev * oh
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1), (b[:, 3] * b[:, 4] * b[:, 5]).view(1, -1)
    want = (o3 / torch.clamp(va + vb - o3, min=1e-6)).numpy()
    assert np.array_equal(oracle.boxes_iou3d(A, B), want)


@pytest.mark.parametrize("rotated", [False, True])
def test_nms_sweep_matches_python_greedy(oracle, rng, rotated):
    n = 150
    boxes = syn.random_boxes(rng, n, centre_range=8.0)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    sb = boxes[order]
    iou = oracle.boxes_iou_bev(sb, sb) if rotated else np.array(
        [[oracle.iou_normal(sb[i], sb[j]) for j in range(n)] for i in range(n)], np.float32)
    for thresh in (0.1, 0.5, 1.0):
        keep, removed = [], np.zeros(n, bool)
        for i in range(n):
            if removed[i]:
                continue
            keep.append(i)
            removed |= (iou[i] > thresh) & (np.arange(n) > i)
        got = oracle.nms(sb, thresh, rotated)
        assert got.tolist() == keep
        if thresh == 1.0:   # Box Seeker config (nms_normal 1.0): nothing is suppressed
            assert len(keep) == n
        assert oracle.nms_gpu(boxes, scores, thresh, rotated).tolist() == order[keep].tolist()


def test_nms_empty_and_single(oracle):
    assert oracle.nms(np.zeros((0, 7), np.float32), 0.5, True).tolist() == []
    assert oracle.nms(np.array([[0, 0, 0, 1, 1, 1, 0]], np.float32), 0.5, False).tolist() == [0]
