#!/usr/bin/env python3
"""Development probe: graph replay / eager forward interleaved at one scene; after each step: are the persistent grids
zero, what the stage counts are, and whether buffers the graph captured were replaced."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
dev = torch.device("cuda", 0)
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
p, o = syn.make_batch([0]); p, o = torch.from_numpy(p).to(dev), torch.from_numpy(o).to(dev)
e = net.engine()

def ws_view():
    n, maxp = 65536, 10
    ws = e._vox_ws
    code = ws[0:8 * n].view(torch.int64)
    rank = ws[524288:524288 + 4 * n].view(torch.int32)
    flag = ws[786432:786432 + 4 * n].view(torch.int32)
    top = ws[1048576:1048576 + 4 * n * maxp].view(torch.int32).view(n, maxp)
    misc = ws[3670016:3670016 + 1024].view(torch.int32)
    return code, rank, flag, top, misc


def state(tag, r):
    torch.cuda.synchronize()
    if "graph" in tag:
        code, rank, flag, top, misc = ws_view()
        nv = int((code >= 0).sum()); nr = int((rank >= 0).sum()); rmax = int(rank.max())
        first = top[:, 0]
        print("   ws: code>=0", nv, "rank>=0", nr, "rank max", rmax, "flag(scan) last", int(flag[-1]), "flag[30786]", int(flag[30786]) if flag.numel() > 30786 else None,
              "top[:,0] != sentinel", int((first != 0x7f7f7f7f).sum()), "top[:,0] < 65536", int(((first >= 0) & (first < 65536)).sum()),
              "scene", misc[:8].tolist(), "n_sorted", int(misc[64]), "n_first", int(misc[128]), flush=True)
    gs = e._get_grids(1, dev)
    dirty = [int((g.bits != 0).sum().item()) for g in gs]
    summ = [int((g.summary != 0).sum().item()) for g in gs]
    ptrs = (gs[0].perm.data_ptr() if gs[0].perm is not None else None, gs[0].perm.numel() if gs[0].perm is not None else None,
            e._vox_ws.data_ptr(), e._vox_ws.numel())
    print(tag, "counts", r["counts"], "dirty words", dirty, "dirty summary", summ, "perm/ws", ptrs, flush=True)

with torch.no_grad():
    for i in range(2):
        state(f"graph {i}", net.forward_points_graphed(p, o, 1, cfg))
    for i in range(1):
        state(f"eager {i}", net.forward_points(p, o, 1, cfg))
    for i in range(2):
        state(f"graph again {i}", net.forward_points_graphed(p, o, 1, cfg))
    g = list(e._graphs.values())[0]
    print("graph static pts valid rows:", int((g.pts[:, 0] < 1e8).sum().item()), "off", g.off.tolist(), "n_prev", g.n_prev)
    print("vox n:", g.vox["n"].tolist(), "n_cells", g.vox["n_cells"].tolist())
