#!/usr/bin/env python3
"""Development probe (CPU only, numpy + the oracle's coordinate functions): how much matrix work an
output-stationary SubM sweep could skip if the rows a workgroup owns were processed in an order sorted by
neighbourhood class.  For stages 2-4 of the synthetic workload: rows in rank-grid order, cut into ranges
(the rows of one persistent workgroup), ranges cut into tiles, tiles into 16-row blocks; reports the share
of (tile, offset), (wave, offset) and (block, offset) pairs that hold at least one neighbour, as is and
after sorting the rows of a range by several keys."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from findnpropagate_amd import synthetic as syn
from oracle import oracle as O


def morton3(v):
    return (v & 1) | ((v & 2) << 1) | ((v & 4) << 2)


def rank_key(idx, shape):
    """rank-grid order of rankgrid.h: scene, patch (row-major), Z-order column, block bottom-to-top, bit."""
    b, z, y, x = [idx[:, i].astype(np.int64) for i in range(4)]
    D, H, W = shape
    bd, bh, bw = (D + 3) >> 2, (H + 3) >> 2, (W + 3) >> 2
    th, tw = (bh + 7) >> 3, (bw + 7) >> 3
    by, bx = y >> 2, x >> 2
    col = (morton3(by & 7) << 1) | morton3(bx & 7)
    blk = (((b * th + (by >> 3)) * tw + (bx >> 3)) * 64 + col) * bd + (z >> 2)
    bit = ((z & 3) << 4) | ((y & 3) << 2) | (x & 3)
    return blk * 64 + bit


def subm_masks(idx, shape):
    """(N, 27) bool: neighbour present at offset k = (dz+1)*9 + (dy+1)*3 + (dx+1)."""
    D, H, W = shape
    b, z, y, x = [idx[:, i].astype(np.int64) for i in range(4)]
    B = int(b.max()) + 1
    occ = np.zeros((B, D + 2, H + 2, W + 2), bool)
    occ[b, z + 1, y + 1, x + 1] = True
    m = np.zeros((idx.shape[0], 27), bool)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                m[:, k] = occ[b, z + 1 + dz, y + 1 + dy, x + 1 + dx]
                k += 1
    return m


def evaluate(m, R, T, WR, order_fn):
    """m (N,27) in processing order before the per-range sort; returns shares of non-empty pairs."""
    N = m.shape[0]
    nr = N // R
    mm = m[: nr * R].reshape(nr, R, 27)
    if order_fn is not None:
        out = np.empty_like(mm)
        for r in range(nr):
            out[r] = mm[r][order_fn(mm[r])]
        mm = out
    nt = R // T
    t = mm[:, : nt * T].reshape(nr, nt, T, 27)
    tile = t.any(2)                                   # (nr, nt, 27)
    wave = t.reshape(nr, nt, T // WR, WR, 27).any(3)
    blk = t.reshape(nr, nt, T // 16, 16, 27).any(3)
    # time model of a barrier-per-offset workgroup: an offset costs the maximum over the SIMDs (waves w, w + NW/2 share one)
    nw = T // WR
    half = nw // 2 if nw >= 2 else 1
    per_simd = wave[:, :, :half].astype(int) + (wave[:, :, half:2 * half].astype(int) if nw >= 2 else 0)
    lock = per_simd.max(2) / (2.0 if nw >= 2 else 1.0)   # (nr, nt, 27) in units of "both waves of a SIMD busy"
    return {"tile": float(tile.mean()), "wave": float(wave.mean()), "block": float(blk.mean()),
            "lockstep": float(lock.mean()), "density": float(mm.mean())}


def key_lex(planes_first=True):
    def f(mr):
        w = (1 << np.arange(27)).astype(np.int64)
        lo, hi = mr[:, 0:9].any(1), mr[:, 18:27].any(1)
        cls = lo.astype(np.int64) * 2 + hi.astype(np.int64)          # 0 flat, 1 above only, 2 below only, 3 both
        cls = np.array([0, 1, 3, 2])[cls]                            # order: flat, above, both, below
        key = cls * (1 << 28) + (mr.astype(np.int64) * w).sum(1)
        return np.argsort(key, kind="stable")
    return f


def key_zonly(mr):
    lo, hi = mr[:, 0:9].any(1), mr[:, 18:27].any(1)
    cls = np.array([0, 1, 3, 2])[lo.astype(np.int64) * 2 + hi.astype(np.int64)]
    return np.argsort(cls, kind="stable")


def key_gray(mr):
    """sort by (z class, in-plane class of the own plane, then full mask)"""
    w = (1 << np.arange(27)).astype(np.int64)
    lo, hi = mr[:, 0:9].any(1), mr[:, 18:27].any(1)
    cls = np.array([0, 1, 3, 2])[lo.astype(np.int64) * 2 + hi.astype(np.int64)]
    pc_lo, pc_hi = mr[:, 0:9].sum(1), mr[:, 18:27].sum(1)
    key = (cls * 16 + np.where(cls == 1, pc_hi, np.where(cls == 3, pc_lo, pc_lo))) * (1 << 28) + (mr.astype(np.int64) * w).sum(1)
    return np.argsort(key, kind="stable")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--grid", type=int, default=256, help="workgroups of the persistent grid at the FULL batch of 64")
    args = ap.parse_args()
    shape = [41, 1440, 1440]
    idx_all = []
    for b in range(args.batch):
        pts = syn.make_scene(b)
        _, c, _ = O.voxelize(pts, syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 10, 160000)
        idx_all.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    idx = np.concatenate(idx_all, 0)
    stages = [("stage2 32ch", 3, 1, (2, 2, 2), (1, 1, 1)), ("stage3 64ch", 3, 1, (2, 2, 2), (1, 1, 1)), ("stage4 128ch", 3, 1, (2, 2, 2), (0, 1, 1))]
    for name, k, _, s, p in stages:
        idx, shape, *_ = O.rulebook_strided(idx, shape, k, s, p)
        idx = idx[np.argsort(rank_key(idx, shape), kind="stable")]
        m = subm_masks(idx, shape)
        N = idx.shape[0]
        print(json.dumps({"stage": name, "rows": N, "rows_per_scene": N // args.batch, "shape": shape,
                          "pairs_per_row": round(float(m.sum(1).mean()), 2)}))
        per_scene = N / args.batch
        for label, T, WR, grid in (("128ch: 8 waves x 48", 384, 48, 256), ("64ch tile: 4 waves x 32", 128, 32, 512),
                                   ("8 waves x 32", 256, 32, 512)):
            R = int(per_scene * 64 / grid) // T * T
            if R < T:
                continue
            for kn, fn in (("as is", None), ("z class", key_zonly), ("lex", key_lex()), ("zc+pc+lex", key_gray)):
                r = evaluate(m, R, T, WR, fn)
                print("   ", label, f"R={R}", kn, {a: round(v, 3) for a, v in r.items()})


if __name__ == "__main__":
    main()
