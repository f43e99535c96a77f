#!/usr/bin/env python3
"""Development: does keeping TWO 64-scene batches in flight (PointsPipeline, depth 2) beat one after the other?"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B))); pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
out = {}
with torch.no_grad():
    for _ in range(5): net.forward_points(pts, off, B, cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): net.forward_points(pts, off, B, cfg)
    torch.cuda.synchronize(); out["stream_scenes_per_s"] = 20 * B / (time.perf_counter() - t0)
    cap = (pts.shape[0] + 65535) // 65536 * 65536
    for depth in (1, 2, 3):
        pipe = net.points_pipeline(B, cfg, depth=depth, capacity=cap)
        frames = [(pts, off)] * 24
        for _ in pipe.map(frames[:6]): pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in pipe.map(frames): pass
        torch.cuda.synchronize(); out[f"pipeline_depth{depth}_scenes_per_s"] = len(frames) * B / (time.perf_counter() - t0)
        del pipe
print(json.dumps(out))
