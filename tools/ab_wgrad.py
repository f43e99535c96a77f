#!/usr/bin/env python3
"""Development (GPU box): same-process A/B of the weight gradient of the ranked 32 -> 32 / 64 -> 64 SubM stages of a B-scene forward
across builds of the library: fnp_spconv_wgrad on the table (`table`) and fnp_spconv_wgrad_pairs on the pair lists (`pairs`), same
buffers, interleaved rounds, results compared.  usage: tools/ab_wgrad.py --batch 16 --variants a,b [--channels 32,64]"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import lib as _l, sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--variants", default=""); ap.add_argument("--channels", default="32,64")
ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--reps", type=int, default=10)
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
libs = {"main": _l.load()}
for v in [v for v in args.variants.split(",") if v]:
    path = v if v.endswith(".so") else os.path.join(ROOT, "findnpropagate_amd", "csrc", "ab", f"libfnp_{v}.so")
    libs[v if not v.endswith(".so") else "ext:" + path.split("/")[-3]] = ctypes.CDLL(path)
P, I = ctypes.c_void_p, ctypes.c_int
for L in libs.values():
    L.fnp_spconv_wgrad.restype = ctypes.c_int; L.fnp_spconv_wgrad_pairs.restype = ctypes.c_int; L.fnp_spconv_wgrad_workspace_bytes.restype = ctypes.c_longlong
want = [int(c) for c in args.channels.split(",")]
seen = set()
stream = P(torch.cuda.current_stream(dev).cuda_stream)
for tag, rb, n_dev in log:
    cin, cout, K, has_res, ranked = tag
    if not (ranked and K == 27 and cin == cout and cin in want) or cin in seen:
        continue
    n = int(n_dev.item())
    if rb.nbr is None or getattr(rb, "_lean", False):
        # the engine's lean table: rebuild the full one from the stage's coordinates is not available here -> transpose-free fallback
        print(json.dumps({"channels": cin, "skipped": "lean table"})); continue
    seen.add(cin)
    x = torch.randn((n, cin), device=dev).to(torch.bfloat16)
    dy = torch.randn((n, cout), device=dev).to(torch.bfloat16)
    po, pi, cnt = S.rulebook_pairs(rb, n_dev, rows=n)
    outs, wss = {}, {}
    for name, L in libs.items():
        wss[name] = torch.empty((int(L.fnp_spconv_wgrad_workspace_bytes(I(K), I(cin), I(cout))),), dtype=torch.uint8, device=dev)
        outs[name] = {"table": torch.zeros((K, cout, cin), device=dev), "pairs": torch.zeros((K, cout, cin), device=dev)}

    def run(name, mode):
        L, ws, dw = libs[name], wss[name], outs[name][mode]
        if mode == "table":
            rc = L.fnp_spconv_wgrad(P(x.data_ptr()), I(_l.dtype_code(x)), P(dy.data_ptr()), I(_l.dtype_code(dy)), P(rb.nbr.data_ptr()), I(rb.nbr.shape[1]), I(K),
                                    P(n_dev.data_ptr()), I(n), P(dw.data_ptr()), I(0), I(cin), I(cout), P(ws.data_ptr()), ctypes.c_longlong(ws.numel()), stream)
        else:
            rc = L.fnp_spconv_wgrad_pairs(P(x.data_ptr()), I(_l.dtype_code(x)), P(dy.data_ptr()), I(_l.dtype_code(dy)), P(po.data_ptr()), P(pi.data_ptr()), P(cnt.data_ptr()),
                                          I(po.shape[1]), I(K), P(n_dev.data_ptr()), I(n), P(dw.data_ptr()), I(0), I(cin), I(cout), P(ws.data_ptr()),
                                          ctypes.c_longlong(ws.numel()), stream)
        assert rc == 0, (name, mode, rc)

    keys = [(name, mode) for name in libs for mode in ("table", "pairs")]
    for k in keys:
        for _ in range(2):
            run(*k)
    torch.cuda.synchronize()
    ref = outs["main"]["pairs"]
    err = {f"{a}:{m}": float((outs[a][m] - ref).abs().max() / ref.abs().max()) for a, m in keys}
    times = {k: [] for k in keys}
    for _ in range(args.rounds):
        for k in keys:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run(*k)
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / args.reps * 1e3)
    print(json.dumps({"channels": cin, "rows": n, "scenes": B, "us_median": {f"{a}:{m}": round(float(np.median(v)), 1) for (a, m), v in times.items()},
                      "rel_err_vs_main_pairs": err}), flush=True)
