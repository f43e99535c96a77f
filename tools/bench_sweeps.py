#!/usr/bin/env python3
"""Secondary measurement: the fused path on 10-sweep-sized inputs (the reference's transfusion_lidar.yaml aggregates
10 lidar sweeps: ~250-300 k points, up to 120-160 k voxels per scene).  A 10-sweep scene is emulated by ten
synthetic sweeps of the same static world from ego positions 0.5 m apart (time channel = sweep age)."""
import argparse, os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--sweeps", type=int, default=10)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0); B = args.batch
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()


def scene(seed):
    parts = []
    for j in range(args.sweeps):
        p = syn.make_scene(seed).copy()           # same world (seeded), ego moved along x: points shift
        p[:, 0] += 0.5 * j + 0.013 * j * j
        p[:, 1] += 0.07 * j
        p[:, 4] = 0.05 * j
        parts.append(p)
    p = np.concatenate(parts, 0)
    r = syn.POINT_CLOUD_RANGE
    return np.ascontiguousarray(p[(p[:, 0] >= r[0]) & (p[:, 0] <= r[3]) & (p[:, 1] >= r[1]) & (p[:, 1] <= r[4])])


scenes = [scene(s) for s in range(B)]
off = np.zeros(B + 1, np.int32); off[1:] = np.cumsum([s.shape[0] for s in scenes])
pts = torch.from_numpy(np.concatenate(scenes, 0)).to(dev); offd = torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
with torch.no_grad():
    for _ in range(3): r = net.forward_points(pts, offd, B, cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.reps): r = net.forward_points(pts, offd, B, cfg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.reps
print(json.dumps({"scenes_per_step": B, "sweeps": args.sweeps, "points_per_scene": int(pts.shape[0] // B),
                  "voxels_per_scene": r["counts"][0] // B, "site_counts": r["counts"], "ms_per_step": round(dt * 1e3, 3),
                  "scenes_per_s": round(B / dt, 1)}))
