import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
dev=torch.device('cuda',0)
B=16
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B)))
pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
eng = net.engine()
with torch.no_grad():
    for it in range(3):
        eng.rulebook_log = []
        res = net.forward_points(pts, off, B, cfg)
        torch.cuda.synchronize()
        out=[]
        for tag, rb, n_dev in eng.rulebook_log:
            n = int(n_dev.item())
            out.append((tag[:3], n, int((rb.nbr[:, :n] >= 0).sum().item()), rb.nbr.shape[1]))
        print(it, eng.cap_factor, out[:3], out[5:7], out[10:12])
