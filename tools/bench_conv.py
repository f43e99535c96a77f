#!/usr/bin/env python3
"""Time each sparse-conv layer class of VoxelResBackBone8x in isolation on real rulebooks
(B synthetic scenes), 20 launches each, HIP events.  Development tool for kernel iteration."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16"]); ap.add_argument("--valu", action="store_true"); ap.add_argument("--identity", action="store_true", help="replace every valid rulebook entry by the output row itself (perfect gather locality, same instruction stream)"); ap.add_argument("--fracs", type=str, default="1.0", help="comma list: time each layer on the first frac*n output rows too (tile staircase)")
ap.add_argument("--tile", default="auto", choices=["auto", "on", "off"], help="LDS-tile kernel for the ranked 32 -> 32 layers: by size / forced / forbidden")
ap.add_argument("--no-residual", action="store_true"); ap.add_argument("--only", default="", help="e.g. 32x32: time this layer class only")
ap.add_argument("--sort", default="auto", choices=["auto", "on", "off", "identity", "fake18", "fakemix"], help="class-sorted sweep of the 128 -> 128 layers (identity: the sorted kernel on the natural order with every offset live = the mechanism's overhead)")
ap.add_argument("--tile-stats", action="store_true", help="far-neighbour statistics of the ranked 32 -> 32 layer per 256-row tile")
args = ap.parse_args()
TILE = {"auto": None, "on": True, "off": False}[args.tile]
dev = torch.device("cuda", 0)
B = args.batch
TD = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[args.dtype]
EB = 4 if args.dtype == "fp32" else 2
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B)))
pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
eng = net.engine()
with torch.no_grad():
    net.forward_points(pts, off, B, cfg)
    eng.rulebook_log = []
    res = net.forward_points(pts, off, B, cfg)
log, eng.rulebook_log = eng.rulebook_log, None
P = eng.prepare()
seen = {}
for tag, rb, n_dev in log:
    cin, cout, K, has_res, ranked = tag
    if cin == 5 or (cin, cout, K) in seen: continue
    if args.only and args.only != f"{cin}x{cout}": continue
    seen[(cin, cout, K)] = 1
    n = int(n_dev.item()); pairs = int((rb.nbr[:, :n] >= 0).sum().item())
    n_in = int(rb.nbr[:, :n].max().item()) + 1
    x = torch.randn((n_in, cin), device=dev).to(TD)
    w = (torch.randn((K, cout, cin), device=dev) * 0.05).to(TD)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    resid = None if args.no_residual else torch.randn((rb.cap_out, cout), device=dev).to(TD)
    n_full = n
    if (cin, cout, K) == (128, 128, 27) and args.sort != "auto":
        rb.__dict__.pop("_sorted", None)
        if args.sort == "identity":
            rb._sorted = (torch.arange(rb.cap_out, dtype=torch.int32, device=dev), torch.full((rb.cap_out // 16 + 1,), (1 << 27) - 1, dtype=torch.int32, device=dev))
        if args.sort == "fake18":   # timing probe (wrong results): natural order, every tile sweeps offsets 9..26 only
            rb._sorted = (torch.arange(rb.cap_out, dtype=torch.int32, device=dev), torch.full((rb.cap_out // 16 + 1,), ((1 << 27) - 1) & ~0x1ff, dtype=torch.int32, device=dev))
        if args.sort == "fakemix":  # timing probe (wrong results): natural order, blocks alternate per 384 rows between 18 and 27 live offsets
            bmk = torch.full((rb.cap_out // 16 + 1,), (1 << 27) - 1, dtype=torch.int32, device=dev)
            blk = torch.arange(rb.cap_out // 16 + 1, device=dev)
            bmk[(blk // 24) % 3 != 1] = ((1 << 27) - 1) & ~0x1ff
            rb._sorted = (torch.arange(rb.cap_out, dtype=torch.int32, device=dev), bmk)
        if args.sort == "on":
            for _ in range(3): S.classsort(rb, n_dev, 128)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(args.reps): S.classsort(rb, n_dev, 128)
            e1.record(); torch.cuda.synchronize()
            bm = rb._sorted[1][: (n + 15) // 16].cpu().numpy().view(np.uint32)
            # tile masks: the 8 XCD groups of the 256-workgroup grid, tiles of 24 blocks from each group's first row
            Gr = 256; nb16 = (n + 15) // 16; live = []; rows = []
            for xg in range(8):
                b0, b1 = nb16 * (xg * 32) // Gr, nb16 * (xg * 32 + 32) // Gr
                for t0 in range(b0, b1, 24):
                    t1 = min(b1, t0 + 24)
                    live.append(bin(int(np.bitwise_or.reduce(bm[t0:t1]))).count("1")); rows.append(t1 - t0)
            print(json.dumps({"tile_offsets_live(weighted by rows; the partial last round counted as full tiles)": round(float(np.average(live, weights=rows)) / 27, 3)}))
            nt24 = len(bm) // 24
            tm = np.bitwise_or.reduce(bm[: nt24 * 24].reshape(nt24, 24), axis=1)
            print(json.dumps({"classsort_ms": round(e0.elapsed_time(e1) / args.reps, 4), "seg": os.environ.get("FNP_SORT_SEG", "3"),
                              "block_offsets_live": round(float(np.mean([bin(int(v)).count("1") for v in bm])) / 27, 3),
                              "tile24_offsets_live(approx)": round(float(np.mean([bin(int(v)).count("1") for v in tm])) / 27, 3)}))
    if args.tile_stats and ranked and K == 27:
        # far rows of spconv_tile.hip's tiles: per producer wave (32 rows x 27 offsets) the unique far row ids
        T, HALO, WR = 256, 64, 32
        nt = (n + T - 1) // T
        nb = torch.full((K, nt * T), -1, dtype=torch.int64, device=dev); nb[:, :n] = rb.nbr[:, :n].long()
        nb = nb.view(K, nt, T // WR, WR).permute(1, 2, 0, 3).reshape(nt, T // WR, K * WR)     # (tile, wave, entries)
        base = (torch.arange(nt, device=dev) * T)[:, None, None]
        lo = (base - HALO).clamp(min=0)
        far = (nb >= 0) & ((nb < lo) | (nb >= lo + T + 2 * HALO))
        key = torch.where(far, nb, torch.full_like(nb, -1))
        srt = key.sort(dim=2).values
        first = torch.ones_like(srt, dtype=torch.bool); first[..., 1:] = srt[..., 1:] != srt[..., :-1]
        uniq = (first & (srt >= 0)).sum(2).float()                    # unique far rows per (tile, wave)
        refs = far.sum(2).float()
        # unique far rows per tile
        kt = key.reshape(nt, -1).sort(dim=1).values
        ft = torch.ones_like(kt, dtype=torch.bool); ft[:, 1:] = kt[:, 1:] != kt[:, :-1]
        uniq_t = (ft & (kt >= 0)).sum(1).float()
        q = lambda x, p: round(x.flatten().quantile(p).item(), 1) if x.numel() < 16_000_000 else None
        print(json.dumps({"tile_stats": f"{cin}x{cout}", "tiles": nt, "far_refs_per_wave_mean": round(refs.mean().item(), 1),
                          "uniq_far_per_wave": {"mean": round(uniq.mean().item(), 2), "p50": q(uniq, .5), "p90": q(uniq, .9), "p99": q(uniq, .99), "max": uniq.max().item()},
                          "uniq_far_per_tile": {"mean": round(uniq_t.mean().item(), 1), "p90": q(uniq_t, .9), "p99": q(uniq_t, .99), "max": uniq_t.max().item()},
                          "waves_over": {c: round((uniq > c).float().mean().item(), 4) for c in (8, 16, 24, 32, 48)},
                          "tiles_over": {c: round((uniq_t > c).float().mean().item(), 4) for c in (64, 128, 192, 256)}}))
        for halo in (32, 96, 128):
            lo2 = (base - halo).clamp(min=0)
            far2 = (nb >= 0) & ((nb < lo2) | (nb >= lo2 + T + 2 * halo))
            k2 = torch.where(far2, nb, torch.full_like(nb, -1)).reshape(nt, -1).sort(dim=1).values
            f2 = torch.ones_like(k2, dtype=torch.bool); f2[:, 1:] = k2[:, 1:] != k2[:, :-1]
            u2 = (f2 & (k2 >= 0)).sum(1).float()
            print(json.dumps({"halo": halo, "far_refs_per_tile": round(far2.sum().item() / nt, 1), "uniq_far_per_tile_mean": round(u2.mean().item(), 1), "p99": q(u2, .99)}))
    for frac in [float(f) for f in args.fracs.split(",")]:
        n = int(n_full * frac); n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        pairs = int((rb.nbr[:, :n] >= 0).sum().item())
        for _ in range(3): S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=ranked, valu=args.valu, tile=TILE)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(args.reps): S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=ranked, valu=args.valu, tile=TILE)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        if os.environ.get("FNP_LIB_PATH", "").find("mstamp") >= 0 and (cin, cout, K) == (128, 128, 27):
            import ctypes
            raw = ctypes.CDLL(os.environ["FNP_LIB_PATH"]); buf = (ctypes.c_ulonglong * 8)()
            raw.fnp_debug_mfma_stamps(buf)
            S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=ranked, valu=args.valu, tile=TILE)
            torch.cuda.synchronize(); raw.fnp_debug_mfma_stamps(buf)
            tot = sum(buf[i] for i in range(4))
            print(json.dumps({"mfma128_wave_cycle_share[offset_body,barrier,tile_prologue,epilogue]": [round(buf[i] / tot, 3) for i in range(4)],
                              "cycles_per_wave": round(tot / (256 * 8))}))
        if TILE and (cin, cout, K) in ((32, 32, 27), (64, 64, 27)):   # the one-off restatement of the rulebook
            L = S._l.load(); tb = torch.empty_like(rb._tile_rb[cin]); REC, TR, OV = (14864, 256, 256) if cin == 32 else (7440, 128, 128)
            torch.cuda.synchronize(); e0.record()
            for _ in range(args.reps): L.fnp_tile_rulebook_build(S._l.ptr(rb.nbr), rb.nbr.shape[1], rb.K, S._l.ptr(n_dev), rb.cap_out, cin, S._l.ptr(tb), S._l.stream())
            e1.record(); torch.cuda.synchronize()
            nt_ = (n + TR - 1) // TR; esc = tb.view(-1, REC)[:nt_, REC - 16:REC - 16 + TR // 32]
            print(json.dumps({"tile_rulebook_build_ms": round(e0.elapsed_time(e1) / args.reps, 4), "wave_tiles_with_escape": round(esc.float().mean().item(), 5),
                              "far_rows_per_tile": round((tb.view(-1, REC)[:nt_, 27 * TR * 2:27 * TR * 2 + OV * 4].contiguous().view(torch.int32) >= 0).float().sum(1).mean().item(), 1)}))
        if os.environ.get("FNP_LIB_PATH", "").find("stamp") >= 0 and (cin, cout) in ((32, 32), (64, 64)):
            import ctypes
            raw = ctypes.CDLL(os.environ["FNP_LIB_PATH"])
            buf = (ctypes.c_ulonglong * 32)()
            raw.fnp_debug_tile_stamps(buf)
            S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=ranked, valu=args.valu, tile=TILE)
            torch.cuda.synchronize()
            raw.fnp_debug_tile_stamps(buf)
            tiles = (n + 255) // 256 if cin == 32 else (n + 127) // 128
            if cin == 64:
                print(json.dumps({"stamp64_cycles_per_tile[slab_req,barrier,put,req_next,sweep,epilogue]": [round(buf[i] / (tiles * 4)) for i in range(6)], "slowest": [round(buf[16 + i] / (tiles / 512)) for i in range(6)]}))
            # s_memtime ticks (shader cycles) summed over the waves of a role: per tile and wave
            print(json.dumps({"stamp_cycles_per_tile": {"consumer[sweep,epilogue,barrier]": [round(buf[i] / (tiles * 8)) for i in range(3)],
                                                     "producer[codes,requests,translate,rows,overflow_req,barrier]": [round(buf[8 + i] / (tiles * 8)) for i in range(6)]},
                              "slowest_wave_cycles_per_tile": {"consumer": [round(buf[16 + i] / (tiles / 256)) for i in range(3)], "producer": [round(buf[24 + i] / (tiles / 256)) for i in range(6)]},
                              "producer_wave_tiles_with_escape": round(buf[8 + 6] / (tiles * 8), 4), "far_list_lanes_per_wave_tile(lane0 only)": round(buf[8 + 7] / (tiles * 8), 4)}))
        dense_flop = 2.0 * n * K * cin * cout; alg_flop = 2.0 * pairs * cin * cout
        byts = pairs * (cin * EB + 8) + 2 * n * cout * EB + K * cin * cout * EB
        print(json.dumps({"layer": f"{cin}x{cout}k{K}", "n_out": n, "pairs": pairs, "density": round(pairs / (n * K), 3), "ms": round(ms, 4),
                          "dense_TF": round(dense_flop / ms / 1e9, 1), "alg_TF": round(alg_flop / ms / 1e9, 1), "alg_GBs": round(byts / ms / 1e6, 1)}))
