#!/usr/bin/env python3
"""Development probe: per layer, the share of output sites with no neighbour in the dz = -1 plane (offsets 0..8), in the
dz = +1 plane (18..26), in both — and what an offset-plane skip per 16-row block would save if the rows of a tile were
grouped by that 2-bit class."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--tile", type=int, default=384)
args = ap.parse_args()
dev = torch.device("cuda", 0); B = args.batch
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B)))
pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
eng = net.engine()
with torch.no_grad():
    net.forward_points(pts, off, B, cfg)
    eng.rulebook_log = []
    net.forward_points(pts, off, B, cfg)
log, eng.rulebook_log = eng.rulebook_log, None
seen = set()
for tag, rb, n_dev in log:
    cin, cout, K, has_res, ranked = tag
    if K != 27 or (cin, cout) in seen or cin != cout: continue
    seen.add((cin, cout))
    n = int(n_dev.item())
    v = (rb.nbr[:, :n] >= 0)
    lo, hi = v[0:9].any(0), v[18:27].any(0)
    cls = lo.int() * 2 + hi.int()                      # 0: flat, 1: only above, 2: only below, 3: both
    share = [float((cls == c).float().mean()) for c in range(4)]
    # grouping by class inside tiles of `tile` rows, blocks of 16: planes executed per block
    T = args.tile; nt = n // T
    c = cls[: nt * T].reshape(nt, T)
    c_sorted, _ = torch.sort(c, dim=1)
    blk = c_sorted.reshape(nt, T // 16, 16)
    need_lo = ((blk & 2) != 0).any(2); need_hi = ((blk & 1) != 0).any(2)
    planes = 1.0 + need_lo.float().mean().item() + need_hi.float().mean().item()     # of 3
    # without regrouping
    blk0 = c.reshape(nt, T // 16, 16)
    planes0 = 1.0 + ((blk0 & 2) != 0).any(2).float().mean().item() + ((blk0 & 1) != 0).any(2).float().mean().item()
    print(json.dumps({"layer": f"{cin}x{cout}", "n": n, "flat": round(share[0], 3), "only_above": round(share[1], 3),
                      "only_below": round(share[2], 3), "both": round(share[3], 3),
                      "planes_of_3_as_is": round(planes0, 3), "planes_of_3_grouped": round(planes, 3)}))
