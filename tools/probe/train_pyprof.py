"""Development (GPU box): cProfile of training steps with the backward on the calling thread (autograd multithreading off), so that the Python
of the custom backward functions shows."""
import os, sys, cProfile, pstats, io, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
dev = torch.device("cuda", 0); B = 16
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": "bf16"}, 5, grid), 0).to(dev)
pts, off = syn.make_batch(list(range(B)))
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
vox = S.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), B, cfg)
n = int(vox["n"].item())
bd = lambda: {"voxel_features": vox["mean"][:n], "voxel_coords": vox["coords"][:n].float(), "batch_size": B}
opt = torch.optim.SGD(net.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    out = net(bd())
    loss = sum((t.features.float() ** 2).mean() for t in list(out["multi_scale_3d_features"].values()) + [out["encoded_spconv_tensor"]])
    loss.backward(); opt.step()
net.train()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("multithreaded backward: %.2f ms/step" % ((time.perf_counter() - t0) * 100))
torch.autograd.set_multithreading_enabled(False)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("backward on the calling thread: %.2f ms/step" % ((time.perf_counter() - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:11000])
