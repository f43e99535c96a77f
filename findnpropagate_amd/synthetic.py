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
