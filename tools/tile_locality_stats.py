#!/usr/bin/env python3
"""Development probe (CPU, numpy + the oracle's coordinate functions): for stages 2-4 of the synthetic workload, rows in
rank-grid order cut into tiles of T rows — share of the neighbour references that fall into the window [tile - H, tile + T + H)
and the number of DISTINCT far rows per tile (what an overflow table has to hold)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from findnpropagate_amd import synthetic as syn
from oracle import oracle as O
from tools.classsort_stats import rank_key


def stage_indices(seeds, sweeps=1):
    out = {}
    idxs = []
    for b, s in enumerate(seeds):
        pts = syn.make_scene(s) if sweeps == 1 else syn.make_sweeps_scene(s, sweeps)
        v, c, n = O.voxelize(pts, syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 10, 160000)
        idxs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    idx = np.concatenate(idxs)
    shape = [41, 1440, 1440]
    for st, (k, s, p) in enumerate((((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))), start=2):
        idx, shape, *_ = O.rulebook_strided(idx, shape, k, s, p)
        out[st] = (idx.copy(), list(shape))
    return out


def nbr_rows(idx, shape):
    """(N, 27) int: row of the neighbour at each offset or -1, rows in rank order"""
    order = np.argsort(rank_key(idx, shape), kind="stable")
    idx = idx[order]
    D, H, W = shape
    b, z, y, x = [idx[:, i].astype(np.int64) for i in range(4)]
    B = int(b.max()) + 1
    grid = np.full((B, D + 2, H + 2, W + 2), -1, np.int32)
    grid[b, z + 1, y + 1, x + 1] = np.arange(idx.shape[0], dtype=np.int32)
    out = np.empty((idx.shape[0], 27), np.int32)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                out[:, k] = grid[b, z + 1 + dz, y + 1 + dy, x + 1 + dx]
                k += 1
    return out


if __name__ == "__main__":
    seeds = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3]
    st = stage_indices(seeds)
    for stage in (2, 3, 4):
        idx, shape = st[stage]
        nb = nbr_rows(idx, shape)
        N = nb.shape[0]
        print(f"stage {stage}: {N} rows, {(nb >= 0).sum() / N:.2f} neighbours per row")
        for T in (128, 256, 512):
            for Hh in (32, 64, 128):
                nt = N // T
                far_counts, inside, total = [], 0, 0
                for t in range(nt):
                    e = nb[t * T:(t + 1) * T].ravel()
                    e = e[e >= 0]
                    lo, hi = t * T - Hh, t * T + T + Hh
                    m = (e >= lo) & (e < hi)
                    inside += int(m.sum()); total += e.size
                    far_counts.append(np.unique(e[~m]).size)
                fc = np.array(far_counts)
                print(f"  T {T:4d} halo {Hh:4d}: in-window {inside / total:.3f}; distinct far rows per tile mean {fc.mean():6.1f} p90 {np.percentile(fc, 90):5.0f} p99 {np.percentile(fc, 99):5.0f} max {fc.max():4d}")
