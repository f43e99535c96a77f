"""Development (GPU box): 30 one-scene forwards for a kernel trace.  argv[1] = "graph": replayed from the hipGraph; "ops": print
torch's op table of one stream-launched forward instead (which host-side copies a step issues)."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
mode = sys.argv[1] if len(sys.argv) > 1 else "stream"
dev = torch.device("cuda", 0)
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch([0]); pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
fwd = net.forward_points_graphed if mode == "graph" else net.forward_points
with torch.no_grad():
    if mode == "ops":
        for _ in range(3): fwd(pts, off, 1, cfg)
        torch.cuda.synchronize()
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            fwd(pts, off, 1, cfg)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=80))
    else:
        for _ in range(30): fwd(pts, off, 1, cfg)
torch.cuda.synchronize()
