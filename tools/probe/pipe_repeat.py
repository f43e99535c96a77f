"""Development (GPU box): the same pipeline created several times in one process (depth and scenes per batch from argv): batches per second each time."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
dev = torch.device("cuda", 0); B = int(sys.argv[1]); depth = int(sys.argv[2]); reps = int(sys.argv[3])
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B))); pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
K = 30 if B > 8 else 200
res = []
with torch.no_grad():
    for r in range(reps):
        pipe = net.points_pipeline(B, cfg, depth=depth, capacity=(pts.shape[0] + 65535) // 65536 * 65536)
        for _ in pipe.map([(pts, off)] * 6): pass
        ts, tr = [], []
        sub, resu = pipe.submit, pipe.result
        def tsub(*a):
            t = time.perf_counter(); sub(*a); ts.append(time.perf_counter() - t)
        def tres():
            t = time.perf_counter(); r = resu(); tr.append(time.perf_counter() - t); return r
        pipe.submit, pipe.result = tsub, tres
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in pipe.map([(pts, off)] * K): pass
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        res.append((round(B / dt), "submit ms median %.2f max %.2f" % (1e3 * float(np.median(ts)), 1e3 * max(ts)), "result ms median %.2f" % (1e3 * float(np.median(tr)))))
        del pipe
print(json.dumps({"scenes_per_batch": B, "depth": depth, "scenes_per_s": res}))
