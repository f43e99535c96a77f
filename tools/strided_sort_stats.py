#!/usr/bin/env python3
"""Development probe (CPU only): the strided 3x3x3 layers of VoxelResBackBone8x — per output row the 27-bit mask of the input cells
that exist; how many (tile, offset) / (16-row block, offset) pairs stay live if the rows a workgroup owns are swept in an order
sorted by mask class (which of the three input z planes hold anything, then the full mask)."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from findnpropagate_amd import synthetic as syn
from oracle import oracle as O
from tools.classsort_stats import rank_key, evaluate

def strided_masks(in_idx, in_shape, out_idx, k, s, p):
    D, H, W = in_shape
    B = int(in_idx[:, 0].max()) + 1
    occ = np.zeros((B, D + 4, H + 4, W + 4), bool)
    occ[in_idx[:, 0], in_idx[:, 1] + 2, in_idx[:, 2] + 2, in_idx[:, 3] + 2] = True
    b, z, y, x = [out_idx[:, i].astype(np.int64) for i in range(4)]
    m = np.zeros((out_idx.shape[0], 27), bool)
    j = 0
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                iz, iy, ix = z * s[0] - p[0] + kz, y * s[1] - p[1] + ky, x * s[2] - p[2] + kx
                ok = (iz >= -2) & (iy >= -2) & (ix >= -2) & (iz < D + 2) & (iy < H + 2) & (ix < W + 2)
                m[:, j] = ok & occ[b, np.clip(iz, -2, D + 1) + 2, np.clip(iy, -2, H + 1) + 2, np.clip(ix, -2, W + 1) + 2]
                j += 1
    return m

def key_planes(mr):
    w = (1 << np.arange(27)).astype(np.int64)
    pl = mr[:, 0:9].any(1).astype(np.int64) + 2 * mr[:, 9:18].any(1) + 4 * mr[:, 18:27].any(1)
    return np.argsort(pl * (1 << 28) + (mr.astype(np.int64) * w).sum(1), kind="stable")
def key_planes_only(mr):
    pl = mr[:, 0:9].any(1).astype(np.int64) + 2 * mr[:, 9:18].any(1) + 4 * mr[:, 18:27].any(1)
    return np.argsort(pl, kind="stable")

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4); args = ap.parse_args()
shape = [41, 1440, 1440]
idx_all = []
for b in range(args.batch):
    pts = syn.make_scene(b)
    _, c, _ = O.voxelize(pts, syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 10, 160000)
    idx_all.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
idx = np.concatenate(idx_all, 0)
for name, k, s, p, T, WR in (("16->32", 3, (2, 2, 2), (1, 1, 1), 128, 32), ("32->64", 3, (2, 2, 2), (1, 1, 1), 256, 64), ("64->128", 3, (2, 2, 2), (0, 1, 1), 384, 48)):
    out_idx, out_shape, *_ = O.rulebook_strided(idx, shape, k, s, p)
    out_idx = out_idx[np.argsort(rank_key(out_idx, out_shape), kind="stable")]
    m = strided_masks(idx, shape, out_idx, k, s, p)
    N = out_idx.shape[0]
    print(json.dumps({"layer": name, "out_rows": N, "inputs_per_row": round(float(m.sum(1).mean()), 2),
                      "planes_present_hist": np.bincount(m[:, 0:9].any(1).astype(int) + m[:, 9:18].any(1) + m[:, 18:27].any(1), minlength=4).tolist()}))
    R = int(N / args.batch * 128 / 512) // T * T
    for kn, fn in (("as is", None), ("planes", key_planes_only), ("planes+lex", key_planes)):
        r = evaluate(m, max(R, T), T, WR, fn)
        print("   ", f"T={T} R={max(R, T)}", kn, {a: round(v, 3) for a, v in r.items()})
    idx, shape = out_idx, out_shape
