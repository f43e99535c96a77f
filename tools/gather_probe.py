#!/usr/bin/env python3
"""Development: a GATHER-ONLY probe on the real stage-4 rulebook of the 128-scene batch — what the gather
pattern of the four 128 -> 128 layers can give before anything is built on it.  tools/probe/gather_probe.hip walks 384-position
tiles like the product kernel (8 waves x 48 positions, 4 lanes x 16 B x 4 steps per 256-byte row), offsets that are dead for a tile
skipped, with the rows in
  natural      rank order (every offset is live for nearly every tile)
  product      the shipped class sort (classsort_place_kernel: XCD rounds of 32 tiles, classes [none | above | both | below])
  rangeN       the same four classes sorted (stable) inside ranges of N consecutive tiles (N = 2, 4, 8, 32)
  random       a random permutation (no locality at all: the floor)
  identity     natural order, every entry replaced by the row itself (perfect locality: the ceiling)
and prints time per launch, live (tile, offset) share, gathered TB/s (valid pairs x 256 B / time).
--pmc: one launch per order only (for `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace`: hits / misses per order by dispatch order)."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--pmc", action="store_true")
    args = ap.parse_args()
    from findnpropagate_amd import lib, sparse as S, synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    lib.load()
    P = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libgather_probe.so"))
    dev = torch.device("cuda", 0)
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
    pts, off = syn.make_batch(list(range(args.batch)))
    pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
    eng = net.engine()
    with torch.no_grad():
        net.forward_points(pts, off, args.batch, cfg)
        eng.rulebook_log = []
        net.forward_points(pts, off, args.batch, cfg)
        log, eng.rulebook_log = eng.rulebook_log, None
    tag, rb, n_dev = [e for e in log if e[0][:3] == (128, 128, 27)][0]
    n = int(n_dev.item())
    nbr, stride = rb.nbr, rb.nbr.shape[1]
    assert getattr(rb, "_sorted", None) is not None, "the class sort did not run at this size"
    rowmask = rb._rowmask[:n].to(torch.int64) & 0x7ffffff
    above, below = (rowmask >> 18) != 0, (rowmask & 0x1ff) != 0
    zclass = torch.where(above & below, 2, torch.where(above, 1, torch.where(below, 3, 0)))   # [none | above only | both | below only]
    x = torch.randn((stride, 128), device=dev).to(torch.bfloat16)
    ntiles = (n + 383) // 384
    ar = torch.arange(n, device=dev)

    def range_sort(ntile):
        key = (ar // (384 * ntile)) * 4 + zclass
        return torch.sort(key, stable=True)[1].to(torch.int32)

    orders = [("natural", None, 0), ("product", rb._sorted[0][:n].contiguous(), 0)]
    for r in (2, 4, 8, 32):
        orders.append((f"range{r}", range_sort(r), 0))
    orders += [("random", torch.randperm(n, device=dev).to(torch.int32), 0), ("identity", None, 1)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    sink = torch.zeros((4096,), dtype=torch.int32, device=dev)
    out = {"batch": args.batch, "rows": n, "tiles": ntiles, "neighbours_per_row": float((nbr[:, :n] >= 0).sum().item()) / n, "orders": {}}
    for name, perm, ident in orders:
        tm = torch.zeros((ntiles,), dtype=torch.int32, device=dev)
        pairs = torch.zeros((1,), dtype=torch.int64, device=dev)
        assert P.probe_tile_mask(p(nbr), stride, p(perm), n, p(tm), p(pairs), st) == 0
        torch.cuda.synchronize()
        live = float(sum(bin(int(v) & 0x7ffffff).count("1") for v in tm.cpu().tolist())) / (27.0 * ntiles)
        npairs = int(pairs.item())
        def launch(mode=1):
            rc = P.probe_gather(p(x), ctypes.c_longlong(x.numel() * 2), p(nbr), stride, p(perm), p(tm), n, ident, mode, 256, p(sink), st)
            assert rc == 0, rc
        if args.pmc:
            launch()
            torch.cuda.synchronize()
            out["orders"][name] = {"live_tile_offsets": live, "valid_pairs": npairs}
            continue
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        out["orders"][name] = {"ms_per_launch": ms, "live_tile_offsets": live, "valid_pairs": npairs,
                               "gathered_TBps": npairs * 256 / (ms * 1e-3) / 1e12, "GBps_per_CU": npairs * 256 / (ms * 1e-3) / 1e9 / 256}
        # the same rows fetched ROW-WISE (four whole 256-byte rows per wave instruction instead of sixteen 64-byte quarters)
        for _ in range(3):
            launch(3)
        e0.record()
        for _ in range(args.reps):
            launch(3)
        e1.record()
        torch.cuda.synchronize()
        ms_r = e0.elapsed_time(e1) / args.reps
        for _ in range(3):
            launch(4)
        e0.record()
        for _ in range(args.reps):
            launch(4)
        e1.record()
        torch.cuda.synchronize()
        ms_l = e0.elapsed_time(e1) / args.reps
        out["orders"][name]["rowwise_lds_transpose_ms_per_launch"] = ms_l
        print(f"{name:9s} whole rows through an LDS transpose {ms_l:7.3f} ms", file=sys.stderr)
        eff = (27 * n if ident else npairs)
        out["orders"][name].update({"rowwise_ms_per_launch": ms_r, "rowwise_TBps": eff * 256 / (ms_r * 1e-3) / 1e12, "loads_counted": eff,
                                    "gathered_TBps": eff * 256 / (ms * 1e-3) / 1e12, "GBps_per_CU": eff * 256 / (ms * 1e-3) / 1e9 / 256})
        print(f"{name:9s} quarter-rows {ms:7.3f} ms {eff * 256 / (ms * 1e-3) / 1e12:5.2f} TB/s ({eff * 256 / (ms * 1e-3) / 1e9 / 256:.1f} GB/s per CU)   "
              f"whole rows {ms_r:7.3f} ms {eff * 256 / (ms_r * 1e-3) / 1e12:5.2f} TB/s   live {live:.3f}  pairs {npairs / 1e6:6.2f} M", file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
