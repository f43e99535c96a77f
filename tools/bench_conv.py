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
args = ap.parse_args()
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
    seen[(cin, cout, K)] = 1
    n = int(n_dev.item()); pairs = int((rb.nbr[:, :n] >= 0).sum().item())
    n_in = int(rb.nbr[:, :n].max().item()) + 1
    x = torch.randn((n_in, cin), device=dev).to(TD)
    w = (torch.randn((K, cout, cin), device=dev) * 0.05).to(TD)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    resid = torch.randn((rb.cap_out, cout), device=dev).to(TD)
    n_full = n
    if args.identity and n_in >= n:
        import copy
        rb = copy.copy(rb); ar = torch.arange(rb.nbr.shape[1], device=dev, dtype=torch.int32)[None].expand_as(rb.nbr)
        rb.nbr = torch.where(rb.nbr >= 0, ar, rb.nbr).contiguous()
    for frac in [float(f) for f in args.fracs.split(",")]:
        n = int(n_full * frac); n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        pairs = int((rb.nbr[:, :n] >= 0).sum().item())
        for _ in range(3): S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=ranked, valu=args.valu)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(args.reps): S.conv_forward(x, w, rb, n_dev, scale=sc, shift=sh, residual=resid, relu=True, ranked=ranked, valu=args.valu)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        dense_flop = 2.0 * n * K * cin * cout; alg_flop = 2.0 * pairs * cin * cout
        byts = pairs * (cin * EB + 8) + 2 * n * cout * EB + K * cin * cout * EB
        print(json.dumps({"layer": f"{cin}x{cout}k{K}", "n_out": n, "pairs": pairs, "density": round(pairs / (n * K), 3), "ms": round(ms, 4),
                          "dense_TF": round(dense_flop / ms / 1e9, 1), "alg_TF": round(alg_flop / ms / 1e9, 1), "alg_GBs": round(byts / ms / 1e6, 1)}))
