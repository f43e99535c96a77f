#!/usr/bin/env python3
"""Isolated A/B of the 64 -> 64 and 128 -> 128 SubM layers on real rulebooks of B synthetic scenes: the kernel the engine runs
today (128-row tile kernel / class-sorted gather sweep) against the wide-tile kernel (spconv_wtile.hip), interleaved rounds in
one process, outputs compared bit for bit.  Development tool.  One JSON line per layer class."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--only", default=""); ap.add_argument("--sweeps", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda", 0)
B = args.batch
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
    if (cin, cout, K) not in ((64, 64, 27), (128, 128, 27)) or (cin, cout) in seen:
        continue
    if args.only and args.only != f"{cin}x{cout}":
        continue
    seen.add((cin, cout))
    n = int(n_dev.item())
    x = torch.randn((rb.cap_out, cin), device=dev).bfloat16()
    w = (torch.randn((K, cout, cin), device=dev) * 0.05).bfloat16()
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    resid = torch.randn((rb.cap_out, cout), device=dev).bfloat16()
    if cin == 128:
        rb.__dict__.pop("_sorted", None)
        S.classsort(rb, n_dev, 128)
    variants = {"current": dict(wide=False), "wide": dict(wide=True)}
    outs, times = {}, {k: [] for k in variants}
    L = S._l.load()
    tb = S.tile_rulebook(rb, n_dev, cin, wide=True)
    REC, TR, OV = (28688, 512, 256) if cin == 64 else (14480, 256, 160)
    nt_ = (n + TR - 1) // TR
    rec = tb.view(-1, REC)[:nt_]
    esc = rec[:, REC - 16:REC - 16 + TR // 32]
    far = (rec[:, 27 * TR * 2:27 * TR * 2 + OV * 4].contiguous().view(torch.int32) >= 0).float().sum(1)
    for name, kw in variants.items():
        for _ in range(3):
            outs[name] = S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=True, **kw)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for name, kw in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=True, **kw)
            e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / args.reps)
    if os.environ.get("FNP_LIB_PATH", "").find("wstamp") >= 0:
        import ctypes
        raw = ctypes.CDLL(os.environ["FNP_LIB_PATH"]); buf = (ctypes.c_ulonglong * 8)()
        raw.fnp_debug_wtile_stamps(buf)
        S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=True, wide=True)
        torch.cuda.synchronize(); raw.fnp_debug_wtile_stamps(buf)
        print(json.dumps({"wtile_cycles_per_tile_and_wave[sweep,wait,put,epilogue+barrier]": [round(buf[i] / (nt_ * 4)) for i in range(4)]}))
    dense = 2.0 * n * K * cin * cout
    row = {"layer": f"{cin}x{cout}", "rows": n, "tiles": nt_, "equal": bool(torch.equal(outs["current"][:n], outs["wide"][:n])),
           "groups_with_escape": round(esc.float().mean().item(), 5), "far_rows_per_tile": round(far.mean().item(), 1), "far_rows_max": int(far.max().item())}
    for name in variants:
        ms = float(np.median(times[name]))
        row[name] = {"ms": round(ms, 4), "min_ms": round(min(times[name]), 4), "dense_TF": round(dense / ms / 1e9, 1), "frac_mfma_dense": round(dense / ms / 1e9 / 2500.0, 3)}
    print(json.dumps(row), flush=True)
