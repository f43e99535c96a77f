"""Development (GPU box): VoxelResBackBone8x.forward_points_iter (the batch path's default: two batches in flight, one convolution graph per
batch, counts stored to the host by their launch) at 128 scenes per batch, against the probe form bench.py times."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
dev = torch.device("cuda", 0); B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B))); pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
out = {}
def run_probe():
    for depth in (() if (len(sys.argv) > 2 and sys.argv[2] == "only2") else (2,)):
        pipe = net.points_pipeline(B, cfg, depth=depth, capacity=(pts.shape[0] + 65535) // 65536 * 65536, probe=True)
        for r in pipe.map([(pts, off)] * 8): pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in pipe.map([(pts, off)] * K): pass
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        out["probe_form_depth2" + ("_first" if first else "")] = {"ms_per_batch": round(dt * 1e3, 3), "scenes_per_s": round(B / dt)}
K = 40
first = len(sys.argv) > 2 and sys.argv[2] == "probe_first"
with torch.no_grad():
    if first:
        run_probe()
    for depth in ((2,) if (len(sys.argv) > 2 and sys.argv[2] == "only2") else (2, 3)):    # (forward_points_iter's pipeline, kept across the two passes: a fresh one captures its slots first)
        pipe = net.points_pipeline(B, cfg, depth=depth, capacity=(pts.shape[0] + 65535) // 65536 * 65536)
        for r in pipe.map([(pts, off)] * 8): pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in pipe.map([(pts, off)] * K): pass
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        out["iter_form_depth%d" % depth] = {"ms_per_batch": round(dt * 1e3, 3), "scenes_per_s": round(B / dt)}
        del pipe
    if not first:
        run_probe()
print(json.dumps(out))
