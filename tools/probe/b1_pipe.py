"""Development (GPU box): one-scene frames through PointsPipeline at depth 1 / 2 / 3 (frames per second), for a given GPU_MAX_HW_QUEUES."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
dev = torch.device("cuda", 0)
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch([0]); pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
out = {"queues": os.environ.get("GPU_MAX_HW_QUEUES")}
order = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3]
with torch.no_grad():
    for depth in order:
        pipe = net.points_pipeline(1, cfg, depth=depth, capacity=65536)
        frames = [(pts, off)] * 300
        for r in pipe.map(frames[:20]): pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in pipe.map(frames): pass
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(frames)
        out.setdefault("frames_per_s", []).append((depth, round(1.0 / dt)))
        del pipe
print(json.dumps(out))
