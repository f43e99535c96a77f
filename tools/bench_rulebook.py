#!/usr/bin/env python3
"""Development: time the SubM rulebook of stages 2 and 3 (B synthetic scenes) with and without the tile rulebook."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0); B = args.batch
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B))); pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
with torch.no_grad():
    res = net.forward_points(pts, off, B, cfg)
for key, ch in (("x_conv2", 32), ("x_conv3", 64)):
    t = res[key]; idx = t.indices.contiguous(); n = idx.shape[0]
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    cap = n * 5 // 2
    idxc = torch.zeros((cap, 4), dtype=torch.int32, device=dev); idxc[:n] = idx
    g = S.build_grid(idxc, n_dev, B, t.spatial_shape)
    for tc in (None, ch):
        for _ in range(3): rb = S.rulebook_subm(idxc, n_dev, g, 3, tile_channels=tc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(args.reps): rb = S.rulebook_subm(idxc, n_dev, g, 3, tile_channels=tc)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"stage": key, "rows": n, "cap": cap, "tile_channels": tc, "ms": round(e0.elapsed_time(e1) / args.reps, 4)}))
