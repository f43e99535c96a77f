#!/usr/bin/env python3
"""Development probe: fraction of (row block, kernel offset) pairs of each layer's rulebook that hold no
neighbour at all (matrix work an output-stationary kernel could skip), for several block heights."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8)
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
    if (cin, cout, K) in seen: continue
    seen.add((cin, cout, K))
    n = int(n_dev.item())
    v = (rb.nbr[:, :n] >= 0)
    out = {"layer": f"{cin}x{cout}k{K}", "n": n, "density": round(float(v.float().mean()), 3)}
    for h in (16, 32, 48, 64):
        nb = n // h
        blk = v[:, : nb * h].reshape(K, nb, h).any(dim=2)
        out[f"nonempty_{h}"] = round(float(blk.float().mean()), 3)
    # spread of neighbour rows relative to the output row (SubM only)
    print(json.dumps(out))
