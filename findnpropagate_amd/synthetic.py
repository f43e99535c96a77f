"""Seeded synthetic nuScenes-like inputs: 32-beam spinning lidar scenes, pinhole
cameras and 2D detections.  numpy only; used by bench.py, __graft_entry__.smoke() and tests so
that every measurement and parity case runs on the same, reproducible data.
"""
import numpy as np

# class anchors (l, w, h) of the Box Seeker, frustum_proposals_v1.py:270-281
ANCHORS = np.array([
    [4.63, 1.97, 1.74], [6.93, 2.51, 2.84], [6.37, 2.85, 3.19], [10.5, 2.94, 3.47], [12.29, 2.90, 3.87],
    [0.50, 2.53, 0.98], [2.11, 0.77, 1.47], [1.70, 0.60, 1.28], [0.73, 0.67, 1.77], [0.41, 0.41, 1.07]],
    dtype=np.float32)

POINT_CLOUD_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]   # transfusion_lidar.yaml:6
VOXEL_SIZE = [0.075, 0.075, 0.2]                              # transfusion_lidar.yaml:54
MAX_POINTS_PER_VOXEL = 10
MAX_VOXELS_TEST = 160000                                      # transfusion_lidar.yaml:56-59
GROUND_Z = -1.84


def make_boxes(rng, n_boxes=20):
    """(n,7) [x,y,z,dx,dy,dz,yaw] cuboids standing on the ground + their 0-based class."""
    cls = rng.integers(0, ANCHORS.shape[0], size=n_boxes)
    size = ANCHORS[cls] * rng.uniform(0.9, 1.1, size=(n_boxes, 1)).astype(np.float32)
    xy = rng.uniform(-40.0, 40.0, size=(n_boxes, 2))
    # keep the ego vehicle clear
    near = np.linalg.norm(xy, axis=1) < 4.0
    xy[near] += np.sign(xy[near] + 1e-3) * 4.0
    yaw = rng.uniform(-np.pi, np.pi, size=n_boxes)
    boxes = np.zeros((n_boxes, 7), np.float32)
    boxes[:, 0:2] = xy
    boxes[:, 3:6] = size
    boxes[:, 2] = GROUND_Z + size[:, 2] / 2
    boxes[:, 6] = yaw
    return boxes, cls.astype(np.int64)


def _ray_boxes(origin, dirs, boxes):
    """Slab test of rays against rotated boxes.  dirs (R,3) unit; returns (R,) nearest hit t."""
    t_best = np.full(dirs.shape[0], np.inf)
    for b in boxes:
        c, s = np.cos(-b[6]), np.sin(-b[6])
        o = origin - b[:3]
        ox, oy = o[0] * c - o[1] * s, o[0] * s + o[1] * c
        dx, dy = dirs[:, 0] * c - dirs[:, 1] * s, dirs[:, 0] * s + dirs[:, 1] * c
        o_l = np.array([ox, oy, o[2]])
        d_l = np.stack([dx, dy, dirs[:, 2]], axis=1)
        half = b[3:6] / 2
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (-half - o_l) / d_l
            t2 = (half - o_l) / d_l
        tmin = np.nanmax(np.minimum(t1, t2), axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tmax >= tmin) & (tmax > 0)
        t = np.where(tmin > 0, tmin, tmax)
        t_best = np.where(hit & (t < t_best), t, t_best)
    return t_best


def make_scene(seed, n_beams=32, n_azimuth=1084, n_boxes=20, return_boxes=False):
    """One sweep: (N,5) f32 [x,y,z,intensity,t=0] after the ±54 m range mask (≈30k points)."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-30.67, 10.67, n_beams))
    azim = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False)
    az, el = np.meshgrid(azim, elev, indexing="ij")        # azimuth-major firing order
    az, el = az.ravel(), el.ravel()
    dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
    origin = np.array([0.0, 0.0, 0.0])
    # ground plane
    with np.errstate(divide="ignore"):
        t_ground = np.where(dirs[:, 2] < 0, (GROUND_Z - origin[2]) / dirs[:, 2], np.inf)
    # 64 azimuth sectors, one vertical wall each at a horizontal range U(8, 60)
    wall_r = rng.uniform(8.0, 60.0, size=64)
    sector = np.floor((az + np.pi) / (2 * np.pi) * 64).astype(int) % 64
    t_wall = wall_r[sector] / np.maximum(np.cos(el), 1e-6)
    boxes, cls = make_boxes(rng, n_boxes)
    t_box = _ray_boxes(origin, dirs, boxes)
    t = np.minimum(np.minimum(t_ground, t_wall), t_box)
    ok = np.isfinite(t) & (t > 1.0) & (t < 100.0)
    ok &= rng.random(t.shape[0]) > 0.11                     # no-return rays (real sweeps keep ~30k of 34.7k)
    t = t * (1.0 + rng.normal(0.0, 0.002, size=t.shape))   # range noise N(0, 0.002 r)
    xyz = origin[None, :] + dirs * t[:, None]
    pts = np.zeros((xyz.shape[0], 5), np.float32)
    pts[:, :3] = xyz
    pts[:, 3] = rng.uniform(0.0, 255.0, size=xyz.shape[0])
    pts = pts[ok]
    r = POINT_CLOUD_RANGE                                   # common_utils.mask_points_by_range :78-81 (x,y only)
    m = (pts[:, 0] >= r[0]) & (pts[:, 0] <= r[3]) & (pts[:, 1] >= r[1]) & (pts[:, 1] <= r[4])
    pts = np.ascontiguousarray(pts[m])
    if return_boxes:
        return pts, boxes, cls
    return pts


def make_sweeps_scene(seed, sweeps=10):
    """A 10-sweep-sized input (tools/cfgs/dataset_configs/nuscenes_dataset.yaml:5 MAX_SWEEPS 10 under
    nuscenes_models/transfusion_lidar.yaml: ~250-300 k points and up to the 120 k / 160 k voxel caps per sample): `sweeps`
    synthetic sweeps of the same seeded static world from ego positions ~0.5 m apart, the time channel = sweep age, points of
    all sweeps concatenated oldest last (as the reference's get_lidar_with_sweeps appends them) and cut to the x / y range."""
    parts = []
    for j in range(sweeps):
        p = make_scene(seed).copy()
        p[:, 0] += 0.5 * j + 0.013 * j * j
        p[:, 1] += 0.07 * j
        p[:, 4] = 0.05 * j
        parts.append(p)
    p = np.concatenate(parts, 0)
    r = POINT_CLOUD_RANGE
    return np.ascontiguousarray(p[(p[:, 0] >= r[0]) & (p[:, 0] <= r[3]) & (p[:, 1] >= r[1]) & (p[:, 1] <= r[4])])


def make_sweeps_batch(seeds, sweeps=10):
    scenes = [make_sweeps_scene(s, sweeps) for s in seeds]
    off = np.zeros(len(scenes) + 1, np.int32)
    off[1:] = np.cumsum([s.shape[0] for s in scenes])
    return np.concatenate(scenes, 0), off


def make_batch(seeds):
    """Concatenate scenes: points (N,5) f32, batch_offsets (B+1,) int32."""
    scenes = [make_scene(s) for s in seeds]
    off = np.zeros(len(scenes) + 1, np.int32)
    off[1:] = np.cumsum([s.shape[0] for s in scenes])
    return np.concatenate(scenes, axis=0), off


def random_boxes(rng, n, centre_range=20.0):
    """Random (n,7) boxes for operator tests."""
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-centre_range, centre_range, size=(n, 2))
    b[:, 2] = rng.uniform(-2, 1, size=n)
    b[:, 3:6] = rng.uniform(0.5, 8.0, size=(n, 3))
    b[:, 6] = rng.uniform(-np.pi, np.pi, size=n)
    return b


def init_backbone_weights(module, seed=0):
    """Seeded kaiming-normal conv weights + randomised BN affine/statistics so the fused epilogue
    is exercised."""
    import torch

    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() == 5:  # (Cout,kD,kH,kW,Cin)
                fan_in = p.shape[1] * p.shape[2] * p.shape[3] * p.shape[4]
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(1.0 + 0.1 * torch.rand(b.shape, generator=g))
    return module


def calibrate_batchnorm(module, batch_dict):
    """Give a seeded backbone the BatchNorm statistics of a TRAINED one: one forward in train mode with momentum 1 over
    `batch_dict` (voxel_features, voxel_coords, batch_size — device tensors), so that every running_mean / running_var is the
    statistic of the rows that layer actually sees and every stage output is O(1) in eval mode, as SURVEY.md Appendix A.6 describes
    the reference's activations ("values of O(1) after BN").  The seeded statistics of init_backbone_weights leave the kaiming
    weights' growth in (feature scales of 200-300 at the last stages): fine for relative error bounds, useless for BASELINE.json's
    ABSOLUTE 1e-4.  Returns the module in eval mode."""
    import torch

    bns = [m for m in module.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    old = [m.momentum for m in bns]
    was_training = module.training
    try:
        for m in bns:
            m.momentum = 1.0
        module.train()
        with torch.no_grad():
            module(dict(batch_dict))
    finally:
        for m, mom in zip(bns, old):
            m.momentum = mom
        module.train(was_training)
    return module.eval()


# ------------------------------------------------------------------------------------------
# Cameras and 2D detections for the Greedy Box Seeker
# ------------------------------------------------------------------------------------------
IMAGE_SIZE = (900, 1600)                       # (H, W), frustum_proposals_v1.py:205
CAM_YAWS_DEG = [0.0, -55.0, 55.0, 180.0, 110.0, -110.0]


def make_cameras(batch_size=1):
    """6 pinhole cameras per scene: camera_intrinsics, camera2lidar, lidar2image (B,6,4,4) f32 and
    lidar_aug_matrix (B,4,4) = I  (keys of batch_dict read at frustum_proposals_v1.py:535-538)."""
    K = np.eye(4, dtype=np.float64)
    K[0, 0] = K[1, 1] = 1266.0
    K[0, 2], K[1, 2] = 816.0, 491.0
    intr, c2l, l2i = [], [], []
    for yaw in np.deg2rad(CAM_YAWS_DEG):
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        T = np.eye(4)
        T[:3, 0], T[:3, 1], T[:3, 2] = right, down, fwd      # camera x (right), y (down), z (forward) in lidar
        T[:3, 3] = [0.8 * np.cos(yaw), 0.8 * np.sin(yaw), -0.3]
        intr.append(K)
        c2l.append(T)
        l2i.append(K @ np.linalg.inv(T))
    rep = lambda a: np.repeat(np.stack(a)[None].astype(np.float32), batch_size, axis=0)
    return {"camera_intrinsics": rep(intr), "camera2lidar": rep(c2l), "lidar2image": rep(l2i),
            "lidar_aug_matrix": np.repeat(np.eye(4, dtype=np.float32)[None], batch_size, axis=0)}


def box_corners(boxes):
    """(N,7) -> (N,8,3), corner order of box_utils.boxes_to_corners_3d (box_utils.py:28-53)."""
    t = np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]) / 2.0
    c = boxes[:, None, 3:6] * t[None]
    cs, sn = np.cos(boxes[:, 6]), np.sin(boxes[:, 6])
    x = c[..., 0] * cs[:, None] - c[..., 1] * sn[:, None]
    y = c[..., 0] * sn[:, None] + c[..., 1] * cs[:, None]
    return np.stack([x, y, c[..., 2]], -1) + boxes[:, None, 0:3]


def make_img_aug(rng):
    """(6,4,4) img_aug_matrix: per camera a mild resize + rotation + shift of the image plane (the form
    ImageAug3D-style pipelines accumulate: [[s R, t], [0, 1]] acting on (u, v, depth))."""
    out = np.tile(np.eye(4, dtype=np.float64), (6, 1, 1))
    for c in range(6):
        sc, th = rng.uniform(0.85, 1.0), np.deg2rad(rng.uniform(-3.0, 3.0))
        out[c, :2, :2] = sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        out[c, :2, 3] = rng.uniform(-40.0, 40.0, size=2)
    return out.astype(np.float32)


def make_detections(boxes, cls, cams, rng, batch_idx=0, duplicates=True, img_aug=None):
    """Synthetic GLIP-like 2D detections: clipped projection of each cuboid into every camera that
    sees it, jittered +-5 px, score U(0.3, 0.95), label = class + 1 (1-based like
    preprocessed_detector.py:83-85).  Returns float32 (D,4) xyxy, int64 labels, f32 scores,
    int64 batch idx, int64 cam idx — the tuple PreprocessedGLIP.__call__ returns."""
    H, W = IMAGE_SIZE
    corners = box_corners(boxes.astype(np.float64))
    out = []
    for c in range(6):
        L = cams["lidar2image"][batch_idx, c].astype(np.float64)
        for i in range(boxes.shape[0]):
            p = corners[i] @ L[:3, :3].T + L[:3, 3]
            if (p[:, 2] < 1.0).any():
                continue
            uv = p[:, :2] / p[:, 2:3]
            if img_aug is not None:      # detections live in the augmented image
                A = img_aug[c].astype(np.float64)
                uv = uv @ A[:2, :2].T + A[:2, 3]
            x1, y1 = np.clip(uv[:, 0].min(), 0, W), np.clip(uv[:, 1].min(), 0, H)
            x2, y2 = np.clip(uv[:, 0].max(), 0, W), np.clip(uv[:, 1].max(), 0, H)
            if (x2 - x1) < 12 or (y2 - y1) < 12:
                continue
            j = rng.uniform(-5, 5, size=4)
            b = np.array([max(x1 + j[0], 0), max(y1 + j[1], 0), min(x2 + j[2], W), min(y2 + j[3], H)])
            out.append((b, int(cls[i]) + 1, rng.uniform(0.3, 0.95), c))
            if duplicates and rng.random() < 0.25:   # near-duplicate, lower score: exercises the 2D NMS
                out.append((b + rng.uniform(-8, 8, size=4), int(cls[i]) + 1, rng.uniform(0.3, 0.6), c))
    if not out:
        return (np.zeros((0, 4), np.float32), np.zeros((0,), np.int64), np.zeros((0,), np.float32),
                np.zeros((0,), np.int64), np.zeros((0,), np.int64))
    bx = np.stack([o[0] for o in out]).astype(np.float32)
    return (bx, np.array([o[1] for o in out], np.int64), np.array([o[2] for o in out], np.float32),
            np.full((len(out),), batch_idx, np.int64), np.array([o[3] for o in out], np.int64))


# Scene variants of the Box Seeker parity set (tests/golden/make_boxseeker_golden.py): which edge case a seed carries.
#   aug          non-identity lidar_aug_matrix (world rotation / scaling / translation [/ flip] as the reference's
#                augmentor accumulates them; points are handed over in the augmented frame)
#   empty_cam    two cameras without any detection
#   lone_point   one extra lidar return above every beam + a large detection that contains only that point
#                (all three depth quantiles equal, clamp_bottom collapses the frustum: frustum_proposals_v1.py:616-629,817-826),
#                a sub-pixel detection around one ordinary return and one around two returns (interpolated quantile)
#   no_dets      the detector returns nothing (early return :694-700);  low_scores: everything under score_thr
#   img_aug      non-identity img_aug_matrix (image-plane resize / rotation / shift, :1456-1458,1525-1527)
SEEKER_VARIANTS = {3: ("aug",), 4: ("aug", "flip"), 5: ("aug", "empty_cam"), 6: ("empty_cam",), 7: ("lone_point",),
                   8: (), 9: (), 10: ("aug", "flip", "empty_cam", "lone_point"), 11: ("no_dets",), 12: ("low_scores",),
                   13: ("aug", "lone_point"), 14: ("img_aug",), 15: ("aug", "img_aug", "empty_cam")}
# Parity seeds that run with other than the shipped PARAMS / model_cfg (the reference's optional score terms):
# key -> (PARAMS overrides, model_cfg overrides)
SEEKER_PARAM_VARIANTS = {16: ({"dst_w": 0.5}, {"MULT": True}), 17: ({"ego_w": 0.4}, {}),
                         # the options no shipped config sets (frustum_proposals_v1.py:154-196): several boxes per frustum through
                         # the 3D NMS, a fixed search depth, the multi-camera IoU, the occlusion terms, random search positions,
                         # xywh detections, a single search position, and aln_w (the reference raises: see the mirror)
                         18: ({"topk": 3, "nms_normal": 0.3}, {}), 19: ({"search_depth": 3.0}, {}), 20: ({}, {"MULTICAM_IOU": True}),
                         21: ({"occl_w": 0.5}, {}), 22: ({}, {"OCCL_MULT": True}), 23: ({"rand_center": True}, {}),
                         24: ({}, {"BOX_FORMAT": "xywh"}), 25: ({"num_mags": 0}, {}), 26: ({"aln_w": 0.3}, {}),
                         27: ({"topk": 2, "search_depth": 4.0, "occl_w": 0.3, "dst_w": 0.2}, {"MULTICAM_IOU": True})}
LONE_POINT = np.array([15.0, 0.3, 4.5], np.float32)      # elevation 16.7 deg: above the top beam (10.67 deg)


def _project(L, xyz):
    p = xyz.astype(np.float64) @ L[:3, :3].T.astype(np.float64) + L[:3, 3].astype(np.float64)
    return p[:, 0] / p[:, 2], p[:, 1] / p[:, 2], p[:, 2]


def make_seeker_scene(seed, variant=None):
    """Everything FrustumProposerOG.get_proposals reads for one scene (batch size 1).
    variant: tuple of SEEKER_VARIANTS flags (default: the seed's entry, () for seeds 0-2)."""
    flags = tuple(SEEKER_VARIANTS.get(seed, ())) if variant is None else tuple(variant)
    pts, boxes, cls = make_scene(seed, return_boxes=True)
    rng = np.random.default_rng(10_000 + seed)
    cams = make_cameras(1)
    if "img_aug" in flags:
        cams["img_aug_matrix"] = make_img_aug(np.random.default_rng(20_000 + seed))[None]
    dets = make_detections(boxes, cls, cams, rng, img_aug=cams["img_aug_matrix"][0] if "img_aug" in flags else None)
    if "lone_point" in flags:
        lone = np.zeros((1, 5), np.float32)
        lone[0, :3], lone[0, 3] = LONE_POINT, 17.0
        pts = np.concatenate([pts[:1000], lone, pts[1000:]])
        H, W = IMAGE_SIZE
        extra = []
        u, v, _ = _project(cams["lidar2image"][0, 0], LONE_POINT[None])
        extra.append((np.array([u[0] - 115.0, max(v[0] - 58.0, 0.0), u[0] + 115.0, v[0] + 57.0]), 1, 0.93, 0))
        # sub-pixel boxes around one / two ordinary returns seen by camera 3 (the back camera)
        L3 = cams["lidar2image"][0, 3]
        u, v, d = _project(L3, pts[:, :3])
        vis = np.nonzero((d > 3.0) & (u > 50) & (u < W - 50) & (v > 50) & (v < H - 50))[0]
        i0 = vis[len(vis) // 3]
        extra.append((np.array([u[i0] - 0.4, v[i0] - 0.4, u[i0] + 0.4, v[i0] + 0.4]), 9, 0.91, 3))
        # two returns: the nearest neighbour of another return in the image plane
        i1 = vis[2 * len(vis) // 3]
        dist = (u[vis] - u[i1]) ** 2 + (v[vis] - v[i1]) ** 2
        dist[vis == i1] = np.inf
        i2 = vis[int(np.argmin(dist))]
        lo_u, hi_u, lo_v, hi_v = min(u[i1], u[i2]), max(u[i1], u[i2]), min(v[i1], v[i2]), max(v[i1], v[i2])
        extra.append((np.array([lo_u - 0.3, lo_v - 0.3, hi_u + 0.3, hi_v + 0.3]), 10, 0.9, 3))
        dets = (np.concatenate([dets[0], np.stack([e[0] for e in extra]).astype(np.float32)]),
                np.concatenate([dets[1], np.array([e[1] for e in extra], np.int64)]),
                np.concatenate([dets[2], np.array([e[2] for e in extra], np.float32)]),
                np.concatenate([dets[3], np.zeros(len(extra), np.int64)]),
                np.concatenate([dets[4], np.array([e[3] for e in extra], np.int64)]))
    if "empty_cam" in flags:
        keep = (dets[4] != 1) & (dets[4] != 5)
        dets = tuple(d[keep] for d in dets)
    if "no_dets" in flags:
        dets = tuple(d[:0] for d in dets)
    if "low_scores" in flags:
        dets = (dets[0], dets[1], (dets[2] * 0.4).astype(np.float32), dets[3], dets[4])
    if "aug" in flags:
        # DataAugmentor order: flip, rotation, scaling, translation, each left-multiplied into lidar_aug_matrix
        th, sc = rng.uniform(-0.785, 0.785), rng.uniform(0.9, 1.1)
        A = np.eye(4)
        if "flip" in flags:
            A = np.diag([1.0, -1.0, 1.0, 1.0]) @ A
        Rz = np.eye(4)
        Rz[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
        A = Rz @ A
        A = np.diag([sc, sc, sc, 1.0]) @ A
        T = np.eye(4)
        T[:3, 3] = rng.normal(0.0, 0.5, size=3)
        A = (T @ A).astype(np.float32)
        pts = pts.copy()
        pts[:, :3] = (pts[:, :3].astype(np.float64) @ A[:3, :3].T.astype(np.float64) + A[:3, 3]).astype(np.float32)
        cams["lidar_aug_matrix"] = A[None].copy()
    points = np.concatenate([np.zeros((pts.shape[0], 1), np.float32), pts], axis=1)   # collate_batch: [b, x, y, z, i, t]
    return {"points": points, "gt_boxes": boxes, "gt_cls": cls, "dets": dets, "variant": flags, **cams}


class SeekerScenes:
    """A dataset of synthetic Box Seeker scenes in the collated single-scene form tools/extract_pseudo_labels.py:115
    iterates (batch_size 1; 'points' (N,6) [b,x,y,z,i,t], camera matrices (1,6,4,4), 'lidar_aug_matrix' (1,4,4),
    'gt_boxes' (1,G,10) with the 1-based class last, 'frame_id'), plus 'dets' = the 5-tuple a PreprocessedGLIP-like
    detector returns (give the head image_detector=lambda bd: bd['dets']).  `distinct` scenes (seeds seed0 ..) resident
    on `device`, cycled to `n` frames with distinct frame ids."""

    def __init__(self, n, distinct, device, seed0=0, variants=None):
        import torch

        self.n, self.base, self.raw = n, [], []
        for s in range(distinct):
            sc = make_seeker_scene(seed0 + s, variant=None if variants is None else variants[s % len(variants)])
            self.raw.append(sc)
            d = {"points": torch.from_numpy(sc["points"]).to(device), "batch_size": 1,
                 "dets": tuple(torch.from_numpy(a) for a in sc["dets"])}
            for k in ("camera_intrinsics", "camera2lidar", "lidar2image", "lidar_aug_matrix"):
                d[k] = torch.from_numpy(sc[k]).to(device)
            g = np.zeros((1, sc["gt_boxes"].shape[0], 10), np.float32)
            g[0, :, :7] = sc["gt_boxes"]
            g[0, :, 9] = sc["gt_cls"] + 1
            d["gt_boxes"] = torch.from_numpy(g).to(device)
            self.base.append(d)

    def __len__(self):
        return self.n

    def frame_id(self, i):
        return f"synthetic-{i:06d}.pcd.bin"

    def __getitem__(self, i):
        d = dict(self.base[i % len(self.base)])
        d["frame_id"] = self.frame_id(i)
        return d
