import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
dev=torch.device('cuda',0)
for B in (1,2,16):
    pts, off = syn.make_batch(list(range(B)))
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    r = S.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), B, cfg)
    n = int(r['n'].item())
    rb = S.rulebook_subm(r['coords'], r['n'], r['grid'], 3)
    nbr = rb.nbr[:, :n]
    print(B, n, 'pairs', int((nbr>=0).sum()), 'center ok', int((nbr[13]==torch.arange(n,device=dev)).sum()), 'perm -1', int((r['grid'].perm[:n]<0).sum()))
    # second call reusing same grid object after sparse clear
    S.clear_grid(r['grid'], r['coords'], r['n'])
    print('  bits nonzero after clear', int((r['grid'].bits!=0).sum()))
    r2 = S.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), B, cfg, grid=r['grid'])
    rb2 = S.rulebook_subm(r2['coords'], r2['n'], r2['grid'], 3)
    print('  rerun pairs', int((rb2.nbr[:, :n]>=0).sum()))
