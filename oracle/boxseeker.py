"""CPU restatement (numpy, float32) of the Greedy Box Seeker: FrustumProposerOG.get_proposals of
pcdet/models/dense_heads/frustum_proposals_v1.py for the shipped configuration
(tools/cfgs/nuscenes_box_seeker_proposals.yaml:83) and for the options no shipped configuration sets (topk > 1 through
the 3D NMS, search_depth, rand_center, MULTICAM_IOU, occl_w / OCCL_MULT, BOX_FORMAT xywh, num_mags 0), without its
debug/Blender branches.  aln_w is not restated: the reference's own code path raises (IndexError, :988) as soon as a
candidate holds more than three points (fixture tests/golden/boxseeker_seed26.npz records it).

TEST INFRASTRUCTURE ONLY (see oracle/fnp_oracle.c).  Pinned stage by stage against
tests/golden/boxseeker_seed*.npz, which were produced by running the reference's own
get_proposals (tests/golden/make_boxseeker_golden.py).
"""
import numpy as np

from . import oracle as O

F = np.float32
IMAGE_ORDER = [2, 0, 1, 5, 3, 4]                       # frustum_proposals_v1.py:201
IMAGE_H, IMAGE_W = 900, 1600                           # :205
ANCHORS = np.array([[4.63, 1.97, 1.74], [6.93, 2.51, 2.84], [6.37, 2.85, 3.19], [10.5, 2.94, 3.47], [12.29, 2.90, 3.87],
                    [0.50, 2.53, 0.98], [2.11, 0.77, 1.47], [1.70, 0.60, 1.28], [0.73, 0.67, 1.77], [0.41, 0.41, 1.07]], F)  # :270-281

# MULT (model_cfg key, :997-1000) and ego_w (:1017-1021) are accepted in `params` too
DEFAULT_PARAMS = dict(MULT=False, ego_w=0.0, lq=0.0, uq=0.25, cq=1.0, iou_w=1.0, nms_normal=1.0, dst_w=0.0, dns_w=1.0, min_cam_iou=0.3,
                      score_thr=0.45, nms_2d=0.4, nms_3d=0.0, clamp_bottom=1, num_sizes=1, num_mags=6, num_rotations=10,
                      size_min=0.957, size_max=1.2, ry_min=0.0, ry_max=float(np.pi), max_dist=50.0, topk=1,
                      # options outside the shipped yaml (:154-196); model_cfg keys MULTICAM_IOU / OCCL_MULT / BOX_FORMAT included
                      search_depth=None, rand_center=False, occl_w=0.0, OCCL_MULT=False, MULTICAM_IOU=False, BOX_FORMAT="xyxy")


def occl_scores(boxes, xyz):
    """calc_occl_scores (:408-477).  Meant as: per candidate the number of the frustum's points that lie beyond the
    candidate's nearest corner (range from the sensor) without being inside it.  What the reference computes: `mags` is
    (N, 1) (norm with keepdim, :1008) and the in-box mask (N,), so `(mags > m1) & ~mask` broadcasts to (N, N) and its sum is
    the PRODUCT of the two counts — (points beyond the nearest corner) x (points outside the box).  Restated as it runs."""
    mags = np.sqrt((xyz * xyz).sum(1)).astype(F)
    corners = boxes_to_corners_3d(boxes)
    m1 = np.sqrt((corners * corners).sum(2)).astype(F).min(1)
    out = np.zeros((boxes.shape[0],), F)
    for i in range(boxes.shape[0]):
        inside = O.points_in_boxes(xyz[None], boxes[i][None, None])[0] >= 0
        out[i] = F(int((mags > m1[i]).sum()) * int((~inside).sum()))
    return out


def linspace_f32(a, b, n):
    """torch.linspace in float32: first half from the start, second half from the end."""
    a, b = F(a), F(b)
    if n == 1:
        return np.array([a], F)
    step = F((b - a) / F(n - 1))
    i = np.arange(n)
    return np.where(i < n // 2, a + step * i.astype(F), b - step * (n - 1 - i).astype(F)).astype(F)


def boxes_to_corners_3d(boxes):
    """box_utils.boxes_to_corners_3d (box_utils.py:28-53) + rotate_points_along_z (common_utils.py:35-57)."""
    boxes = boxes.astype(F)
    t = (np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]], F) / F(2))
    c = boxes[:, None, 3:6] * t[None]
    cosa, sina = np.cos(boxes[:, 6]).astype(F), np.sin(boxes[:, 6]).astype(F)
    # points @ [[cos, sin, 0], [-sin, cos, 0], [0, 0, 1]]
    x = c[..., 0] * cosa[:, None] + c[..., 1] * (-sina[:, None])
    y = c[..., 0] * sina[:, None] + c[..., 1] * cosa[:, None]
    return (np.stack([x, y, c[..., 2]], -1) + boxes[:, None, 0:3]).astype(F)


def base_proposals(p=DEFAULT_PARAMS):
    """ctor block frustum_proposals_v1.py:284-298 -> base_boxes (10, R*S, 7), base_corners (10, R*S, 8, 3)."""
    sizes = linspace_f32(p["size_min"], p["size_max"], p["num_sizes"])
    rots = linspace_f32(p["ry_min"], p["ry_max"], p["num_rotations"])
    bb = np.zeros((10, p["num_rotations"], p["num_sizes"], 7), F)
    bb[..., 3:6] = ANCHORS[:, None, None, :]
    bb[..., 6] = rots[None, :, None]
    bb[..., 3:6] = bb[..., 3:6] * sizes[None, None, :, None]
    corners = boxes_to_corners_3d(bb.reshape(-1, 7)).reshape(10, -1, 8, 3)
    return bb.reshape(10, -1, 7), corners


def project_to_camera(points, lidar_aug, lidar2image, img_aug=None):
    """:1431-1475 (img_aug: the camera's 4x4 img_aug_matrix or None).  Returns coords (N,3) [u,v,depth], on_img (N,)."""
    cur = points.astype(F).copy()
    cur = cur - lidar_aug[:3, 3]
    cur = np.linalg.inv(lidar_aug[:3, :3].astype(F)).astype(F) @ cur.T
    cur = lidar2image[:3, :3].astype(F) @ cur
    cur = cur + lidar2image[:3, 3].reshape(3, 1)
    cur[2] = np.clip(cur[2], F(1e-5), F(1e5))
    cur[:2] = cur[:2] / cur[2:3]
    if img_aug is not None:                                                  # :1456-1458
        cur = img_aug[:3, :3].astype(F) @ cur
        cur = cur + img_aug[:3, 3].astype(F).reshape(3, 1)
    cur = cur.T
    on = (cur[:, 1] < IMAGE_H) & (cur[:, 1] >= 0) & (cur[:, 0] < IMAGE_W) & (cur[:, 0] >= 0)
    return cur.astype(F), on


def geometry_at_image_coords(image_coords, c2l, intrins, lidar_aug, img_aug=None):
    """:1509-1545: (u,v,d) -> lidar xyz (img_aug: the camera's img_aug_matrix; its post-transformation is undone first)."""
    pts = image_coords.astype(F)
    if img_aug is not None:                                                  # :1525-1527
        pts = pts - img_aug[:3, 3].astype(F)
        pts = (np.linalg.inv(img_aug[:3, :3].astype(F)).astype(F) @ pts.T).T
    pts = np.concatenate([pts[:, :2] * pts[:, 2:3], pts[:, 2:3]], -1)
    combine = (c2l[:3, :3].astype(F) @ np.linalg.inv(intrins[:3, :3].astype(F)).astype(F)).astype(F)
    pts = (combine @ pts.T).T + c2l[:3, 3].astype(F)
    pts = (lidar_aug[:3, :3].astype(F) @ pts.T).T
    pts = pts + lidar_aug[:3, 3].astype(F)
    return pts.astype(F)


def cam_frustum(xyzxyz):
    """get_cam_frustum :128-140."""
    whl = xyzxyz[3:] - xyzxyz[:3]
    center = (xyzxyz[3:] + xyzxyz[:3]) / F(2)
    t = np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]], F) / F(2)
    return (whl[None, :] * t + center).astype(F)


def quantile(x, q):
    """torch.quantile(x, q), interpolation='linear'."""
    s = np.sort(x.astype(F))
    n = s.shape[0]
    rank = F(q) * F(n - 1)
    lo = int(np.floor(rank))
    hi = int(np.ceil(rank))
    w = F(rank - F(lo))
    a, b = s[lo], s[hi]
    return F(a + w * (b - a)) if w < 0.5 else F(b - (b - a) * (F(1) - w))


def box_iou_2d(b1, b2):
    """torchvision.ops.box_iou."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = np.maximum(b1[:, None, :2], b2[None, :, :2])
    rb = np.minimum(b1[:, None, 2:], b2[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return (inter / (a1[:, None] + a2[None, :] - inter)).astype(F)


def batched_nms_2d(boxes, scores, labels, thr):
    """torchvision.ops.batched_nms (coordinate trick): kept indices, score descending."""
    if boxes.shape[0] == 0:
        return np.zeros((0,), np.int64)
    off = labels.astype(F) * (boxes.max() + F(1))
    b = (boxes + off[:, None]).astype(F)
    order = np.argsort(-scores, kind="stable")
    iou = box_iou_2d(b[order], b[order])
    keep, removed = [], np.zeros(len(order), bool)
    for i in range(len(order)):
        if removed[i]:
            continue
        keep.append(order[i])
        removed |= (iou[i] > thr) & (np.arange(len(order)) > i)
    return np.array(keep, np.int64)


def calc_iou(corners, cam_box, lidar_aug, lidar2image, img_aug=None):
    """:1392-1411: 2D IoU of the clamped projected corner hull box vs the detection."""
    pos, _ = project_to_camera(corners.reshape(-1, 3), lidar_aug, lidar2image, img_aug)
    pos = pos[:, :2].reshape(-1, 8, 2)
    pos[..., 0] = np.clip(pos[..., 0], 0, IMAGE_W)
    pos[..., 1] = np.clip(pos[..., 1], 0, IMAGE_H)
    proj = np.concatenate([pos.min(1), pos.max(1)], -1).astype(F)
    return box_iou_2d(proj, cam_box.reshape(1, 4).astype(F)).reshape(-1), proj


def get_proposals(scene, params=None, trace=None, noise=None):
    """scene: dict(points (N,6) [b,x,y,z,..], camera_intrinsics/camera2lidar/lidar2image (1,6,4,4),
    lidar_aug_matrix (1,4,4), dets = (boxes, labels, scores, batch_idx, cam_idx)), batch size 1.
    Returns boxes (K,7), labels (K,), scores (K,) like :1055-1067.  `trace` (list) receives one
    dict per frustum with the intermediate values.  noise: rand_center's draws, one (num_mags, 3) array per frustum that
    reaches :847, in order (the reference takes them from torch.randn)."""
    p = dict(DEFAULT_PARAMS)
    if params:
        p.update(params)
    base_boxes, base_corners = base_proposals(p)
    mags = linspace_f32(0.0, 1.0, p["num_mags"]) if p["num_mags"] > 0 else np.zeros((1,), F)   # :831-834
    noise = list(noise) if noise is not None else None
    det_boxes, det_labels, det_scores, det_b, det_c = scene["dets"]
    pts = scene["points"][scene["points"][:, 0] == 0][:, 1:4].astype(F)
    aug = scene["lidar_aug_matrix"][0].astype(F)
    iaug = (lambda c: scene["img_aug_matrix"][0, c].astype(F)) if "img_aug_matrix" in scene else (lambda c: None)
    out_boxes, out_labels, out_scores = [], [], []
    frusts = []
    for c in IMAGE_ORDER:                                                   # :582
        m = det_c == c
        cb, cl, cs = det_boxes[m], det_labels[m], det_scores[m]
        if cb.shape[0] > 0:
            sel = batched_nms_2d(cb, cs, cl, p["nms_2d"])                    # :587
            cb, cl, cs = cb[sel], cl[sel], cs[sel]
        L = scene["lidar2image"][0, c].astype(F)
        cam_points, on = project_to_camera(pts, aug, L, iaug(c))             # :590
        cam_points = cam_points[on]
        for box, label, score in zip(cb, cl, cs):                            # :593
            if score < p["score_thr"]:
                continue
            if p["BOX_FORMAT"] != "xyxy":                                    # :599-601 (in place: the box stays xyxy from here on)
                box = box.copy()
                box[2:] = box[2:] + box[0:2]
            x1, y1, x2, y2 = box
            on_box = (cam_points[:, 1] < y2) & (cam_points[:, 1] >= y1) & (cam_points[:, 0] < x2) & (cam_points[:, 0] >= x1)
            bp = cam_points[on_box]
            if bp.shape[0] == 0:
                continue
            fmin = quantile(bp[:, 2], p["lq"])                               # :616-629
            fmax = quantile(bp[:, 2], p["uq"]) if p["search_depth"] is None else F(fmin + F(p["search_depth"]))   # :617-622
            cz = quantile(bp[:, 2], p["cq"])
            wc_cam = np.array([[(x1 + x2) / F(2), (y1 + y2) / F(2), cz]], F)   # :630
            wc_xyz = geometry_at_image_coords(wc_cam, scene["camera2lidar"][0, c], scene["camera_intrinsics"][0, c], aug, iaug(c))
            fmax = min(fmax, F(p["max_dist"]))                               # :647-648
            fmin = max(fmin, F(2.0))
            xyzxyz = np.array([box[0], box[1], fmin, box[2], box[3], fmax], F)
            fr = geometry_at_image_coords(cam_frustum(xyzxyz), scene["camera2lidar"][0, c], scene["camera_intrinsics"][0, c], aug, iaug(c))
            frusts.append((fr, c, box.astype(F), bp, int(label), F(score), wc_xyz))
    for fi, (fr, c, box, bp, label, score, wc_xyz) in enumerate(frusts):   # :805
        c2l, K_, L = scene["camera2lidar"][0, c], scene["camera_intrinsics"][0, c], scene["lidar2image"][0, c].astype(F)
        xyz = geometry_at_image_coords(bp, c2l, K_, aug, iaug(c))            # :812-815
        fr = fr.copy()
        if p["clamp_bottom"] > 0:                                            # :817-826
            for d in range(3):
                f1 = max(xyz[:, d].min(), fr[:, d].min())
                f2 = min(xyz[:, d].max(), fr[:, d].max())
                fr[:, d] = np.clip(fr[:, d], f1, f2) if f1 <= f2 else np.minimum(np.maximum(fr[:, d], f1), f2)
        bev = np.stack([(fr[2 * i] + fr[2 * i + 1]) / F(2) for i in range(4)]).astype(F)   # :828
        close = (bev[0] + bev[1]) / F(2)
        far = (bev[2] + bev[3]) / F(2)
        vec = far - close
        if p["search_depth"] is not None:                                    # :841-842
            with np.errstate(invalid="ignore", divide="ignore"):             # (a collapsed frustum: 0 / 0 = NaN there too)
                vec = (vec / np.sqrt((vec * vec).sum()).astype(F)).astype(F) * F(p["search_depth"])
        if not p["rand_center"]:
            bev_pts = (close[None, :] + vec[None, :] * mags[:, None]).astype(F)  # :845
        else:
            bev_pts = (wc_xyz.reshape(1, 3) + np.asarray(noise.pop(0), F)).astype(F)   # :847
        corners = (base_corners[label - 1][None] + bev_pts[:, None, None, :]).reshape(-1, 8, 3).astype(F)
        boxes = np.repeat(base_boxes[label - 1][None], bev_pts.shape[0], 0).copy()
        boxes[..., 0:3] = boxes[..., 0:3] + bev_pts[:, None, :]
        boxes = boxes.reshape(-1, 7).astype(F)
        nrm = np.sqrt((corners * corners).sum(2)).astype(F)                  # :863 softmin over the 8 corners
        e = np.exp(-nrm - (-nrm).max(1, keepdims=True)).astype(F)
        rank = (e / e.sum(1, keepdims=True)).astype(F)
        wfc = (rank[..., None] * corners).sum(1).astype(F)
        f2c = boxes[:, 0:3] - wfc
        boxes[:, 0:3] = boxes[:, 0:3] + f2c
        corners = corners + f2c[:, None, :]
        dist = np.sqrt((wfc * wfc).sum(1)).astype(F)                         # :871-879
        valid = dist < p["max_dist"]
        t = dict(frustum=fi, cam=c, label=label, n_points=bp.shape[0], frust=fr, cand_boxes=boxes.copy(), valid_dist=valid.copy())
        if valid.sum() == 0:
            if trace is not None:
                trace.append(t)
            continue
        idx = np.nonzero(valid)[0]
        if not p["MULTICAM_IOU"]:
            ious, proj = calc_iou(corners[idx], box, aug, L, iaug(c))        # :883
        else:                                                                # :885-886, multicam_ious :1413-1429
            per = [calc_iou(corners[idx], b2, aug, scene["lidar2image"][0, c2].astype(F), iaug(c2))[0]
                   for (_, c2, b2, _, l2, _, _) in frusts if l2 == label]
            tot = np.zeros_like(per[0])
            for v in per:
                tot = (tot + v).astype(F)
            nz = np.sum([(v > 0) for v in per], axis=0)
            ious = (tot / (nz.astype(F) + F(1e-6))).astype(F)
        dd = np.sqrt(((wfc[idx] - wc_xyz.reshape(1, 3)) ** 2).sum(1)).astype(F)   # :886-893
        dists_ranked = (F(1) - (dd - dd.min()) / (dd.max() - dd.min() + F(1e-8))).astype(F)
        keep = ious > p["min_cam_iou"]                                       # :904
        t.update(ious=ious.copy(), idx_after_dist=idx.copy())
        idx, ious, dists_ranked = idx[keep], ious[keep], dists_ranked[keep]
        if idx.shape[0] == 0:
            if trace is not None:
                trace.append(t)
            continue
        counts = O.points_in_boxes_count(xyz, boxes[idx]).astype(F)          # :930-932
        soft = counts / (counts.max() + F(1e-8))                             # :994
        if not p["MULT"]:
            s2 = (soft * F(p["dns_w"]) + ious * F(p["iou_w"]) + dists_ranked * F(p["dst_w"])).astype(F)   # :997
        else:
            s2 = (soft * F(p["dns_w"]) * ious * F(p["iou_w"]) * dists_ranked * F(p["dst_w"])).astype(F)   # :999
        if p["occl_w"] > 0:                                                  # :1007-1014
            occl = occl_scores(boxes[idx], xyz)
            s2 = (s2 + F(p["occl_w"]) * (F(1) - occl / (occl.max() + F(1e-6)))).astype(F)
        if p["ego_w"] > 0:                                                   # :1017-1021
            ego = np.sqrt((boxes[idx, :3] ** 2).sum(1)).astype(F)
            s2 = (s2 + F(p["ego_w"]) * (ego / ego.max())).astype(F)
        if p["OCCL_MULT"]:                                                   # :1022-1027: replaces the score
            occl = occl_scores(boxes[idx], xyz)
            s2 = (soft * ious * occl).astype(F)
        # 3D NMS on the axis-aligned BEV footprints in score order, then the first topk (:1030-1045); topk 1 and threshold
        # 1.0 (the shipped values) reduce to the first maximum of a stable descending sort
        sel = O.nms_gpu(boxes[idx], s2, float(p["nms_normal"]), rotated=False)[: max(int(p["topk"]), 0)]
        best = int(sel[0])
        t.update(idx_final=idx.copy(), counts=counts.copy(), scores=s2.copy(), best=int(idx[best]), selected=idx[sel].copy())
        if trace is not None:
            trace.append(t)
        for b_ in sel:
            out_boxes.append(boxes[idx[int(b_)]])
            out_labels.append(label)
            out_scores.append(score)
    if not out_boxes:
        return np.zeros((0, 7), F), np.zeros((0,), np.int64), np.zeros((0,), F)
    return np.stack(out_boxes).astype(F), np.array(out_labels, np.int64), np.array(out_scores, F)
