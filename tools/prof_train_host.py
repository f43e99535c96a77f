#!/usr/bin/env python3
"""Development: where the host time of the module-path training step goes (cProfile over 10 steps, top cumulative)."""
import cProfile, pstats, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
dev = torch.device("cuda", 0); B = 16
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": "bf16"}, 5, grid), 0).to(dev).train()
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
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 100)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
