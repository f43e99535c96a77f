"""Development (GPU box): one-scene replayed forward — host time inside the two hipGraphLaunch calls against the step's elapsed time."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
from findnpropagate_amd.backbones_3d import spconv_backbone as SB
dev = torch.device("cuda", 0)
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch([0]); pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
acc = {"g": 0.0, "gb": 0.0, "n": 0}
orig = torch.cuda.CUDAGraph.replay
def timed(self):
    t = time.perf_counter(); orig(self); acc["t_last"] = time.perf_counter() - t
    acc.setdefault("calls", []).append(acc["t_last"])
torch.cuda.CUDAGraph.replay = timed
_counts = SB._PointsGraph.counts
def counts_timed(self):
    t = time.perf_counter(); r = _counts(self); acc.setdefault("wait", []).append(time.perf_counter() - t); return r
SB._PointsGraph.counts = counts_timed
with torch.no_grad():
    for _ in range(20): net.forward_points_graphed(pts, off, 1, cfg)
    torch.cuda.synchronize()
    acc["calls"] = []; acc["wait"] = []
    t0 = time.perf_counter()
    K = 200
    for _ in range(K): net.forward_points_graphed(pts, off, 1, cfg)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    # the GPU chain alone: the captured graph(s) replayed back to back, nothing read in between
    g = [v for v in net.engine()._graphs.values()][0]
    torch.cuda.synchronize(); acc2 = acc["calls"]; acc["calls"] = []
    t0 = time.perf_counter()
    for _ in range(K): g.replay()
    t_issue = (time.perf_counter() - t0) / K
    torch.cuda.synchronize()
    el_chain = (time.perf_counter() - t0) / K
    acc["calls"] = acc2
c = np.array(acc["calls"]).reshape(K, -1) * 1e6
print(json.dumps({"ms_per_step": round(el * 1e3, 4), "ms_per_replay_back_to_back": round(el_chain * 1e3, 4), "host_ms_per_replay_issue": round(t_issue * 1e3, 4), "replay_calls_per_step": c.shape[1], "host_us_per_replay_call_median": np.median(c, 0).round(1).tolist(),
                  "host_us_in_replays": round(float(np.median(c.sum(1))), 1),
                  "host_us_waiting_for_counts": round(float(np.median(acc["wait"][:K])) * 1e6, 1),
                  "host_us_elsewhere": round(el * 1e6 - float(np.median(c.sum(1))) - float(np.median(acc["wait"][:K])) * 1e6, 1)}))
