#!/usr/bin/env python3
"""Development (GPU box): same-process, interleaved A/B of fnp_spconv_forward_tiled across several builds of the library.

The rulebooks, features and weights of the ranked 32 -> 32 and 64 -> 64 SubM stages of a B-scene forward are made once with the
shipped library; every variant library (tools/build_variant.sh: findnpropagate_amd/csrc/ab/libfnp_<name>.so) is loaded with
ctypes and launched on the SAME device buffers in interleaved rounds (box-to-box and run-to-run drift cancels), its output
compared bit for bit with the shipped kernel's.  usage: tools/ab_tiled.py --batch 64 --variants s2,s3 [--channels 64,32]"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import lib as _l, sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--variants", default=""); ap.add_argument("--channels", default="64,32")
ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--plain128", action="store_true", help="channels 128: fnp_spconv_forward (row order) instead of the class-sorted sweep")
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
    path = os.path.join(ROOT, "findnpropagate_amd", "csrc", "ab", f"libfnp_{v}.so")
    libs[v] = ctypes.CDLL(path)
for L in libs.values():
    L.fnp_spconv_forward_tiled.restype = ctypes.c_int
    L.fnp_spconv_forward_sorted.restype = ctypes.c_int
P = ctypes.c_void_p
want = [int(c) for c in args.channels.split(",")]
seen = set()
for tag, rb, n_dev in log:
    cin, cout, K, has_res, ranked = tag
    if not (ranked and K == 27 and cin == cout and cin in want) or cin in seen:
        continue
    seen.add(cin)
    n = int(n_dev.item())
    if cin == 128:    # the class-sorted sweep of stage 4 (fnp_spconv_forward_sorted): same harness, the shipped library's class order
        x = torch.randn((rb.cap_out, cin), device=dev).to(torch.bfloat16)
        w = (torch.randn((K, cout, cin), device=dev) * 0.05).to(torch.bfloat16)
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        resid = torch.randn((rb.cap_out, cout), device=dev).to(torch.bfloat16)
        rb.__dict__.pop("_sorted", None)
        S.classsort(rb, n_dev, 128)
        perm, bmask = rb._sorted[:2]
        outs = {k: torch.zeros((rb.cap_out, cout), dtype=torch.bfloat16, device=dev) for k in libs}
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def launch(name):
            rc = libs[name].fnp_spconv_forward_sorted(P(x.data_ptr()), ctypes.c_int(_l.dtype_code(x)), ctypes.c_int(x.shape[0]), P(w.data_ptr()),
                                                      P(rb.nbr.data_ptr()), ctypes.c_int(rb.nbr.shape[1]), P(perm.data_ptr()), P(bmask.data_ptr()),
                                                      P(n_dev.data_ptr()), ctypes.c_int(rb.cap_out), P(outs[name].data_ptr()), P(sc.data_ptr()),
                                                      P(sh.data_ptr()), P(resid.data_ptr()), ctypes.c_int(1), ctypes.c_int(cin), ctypes.c_int(cout), stream)
            assert rc == 0, (name, rc)

        if args.plain128:     # the row-order sweep of the same table (fnp_spconv_forward) instead of the class-sorted one
            for L in libs.values():
                L.fnp_spconv_forward.restype = ctypes.c_int

            def launch(name):
                rc = libs[name].fnp_spconv_forward(P(x.data_ptr()), ctypes.c_int(_l.dtype_code(x)), ctypes.c_int(x.shape[0]), P(w.data_ptr()), P(rb.nbr.data_ptr()),
                                                   ctypes.c_int(rb.nbr.shape[1]), ctypes.c_int(27), P(n_dev.data_ptr()), ctypes.c_int(rb.cap_out), P(outs[name].data_ptr()),
                                                   ctypes.c_int(_l.dtype_code(x)), P(sc.data_ptr()), P(sh.data_ptr()), P(resid.data_ptr()), ctypes.c_int(1), ctypes.c_int(0),
                                                   ctypes.c_int(cin), ctypes.c_int(cout), stream)
                assert rc == 0, (name, rc)
        for name in libs:
            for _ in range(3):
                launch(name)
        torch.cuda.synchronize()
        equal = {name: bool(torch.equal(outs[name][:n], outs["main"][:n])) for name in libs}
        times = {name: [] for name in libs}
        for _ in range(args.rounds):
            for name in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    launch(name)
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / args.reps)
        print(json.dumps({"channels": cin, "kernel": "plain" if args.plain128 else "sorted", "rows": n, "scenes": B,
                          "ms_per_launch_median": {k: round(float(np.median(v)), 4) for k, v in times.items()},
                          "ms_all_rounds": {k: [round(t, 4) for t in v] for k, v in times.items()}, "bit_identical_to_main": equal}), flush=True)
        continue
    x = torch.randn((rb.cap_out, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((K, cout, cin), device=dev) * 0.05).to(torch.bfloat16)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    resid = torch.randn((rb.cap_out, cout), device=dev).to(torch.bfloat16)
    # every library restates the table into ITS tile rulebook (a build may change the tile geometry: FNP_TILE32_MB)
    trbs = {}
    for name, L in libs.items():
        L.fnp_tile_rulebook_bytes.restype = ctypes.c_longlong
        nb = L.fnp_tile_rulebook_bytes(ctypes.c_int(rb.cap_out), ctypes.c_int(cin))
        trbs[name] = torch.empty((nb,), dtype=torch.uint8, device=dev)
        rc = L.fnp_tile_rulebook_build(P(rb.nbr.data_ptr()), ctypes.c_int(rb.nbr.shape[1]), ctypes.c_int(27), P(n_dev.data_ptr()), ctypes.c_int(rb.cap_out),
                                       ctypes.c_int(cin), P(trbs[name].data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        assert rc == 0, (name, rc)
    outs = {k: torch.zeros((rb.cap_out, cout), dtype=torch.bfloat16, device=dev) for k in libs}
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch(name):
        rc = libs[name].fnp_spconv_forward_tiled(P(x.data_ptr()), ctypes.c_int(_l.dtype_code(x)), ctypes.c_int(x.shape[0]), P(w.data_ptr()),
                                                 P(trbs[name].data_ptr()), P(rb.nbr.data_ptr()), ctypes.c_int(rb.nbr.shape[1]), P(n_dev.data_ptr()),
                                                 ctypes.c_int(rb.cap_out), P(outs[name].data_ptr()), P(sc.data_ptr()), P(sh.data_ptr()),
                                                 P(resid.data_ptr()), ctypes.c_int(1), ctypes.c_int(cin), ctypes.c_int(cout), stream)
        assert rc == 0, (name, rc)

    for name in libs:
        for _ in range(3):
            launch(name)
    torch.cuda.synchronize()
    equal = {name: bool(torch.equal(outs[name][:n], outs["main"][:n])) for name in libs}
    times = {name: [] for name in libs}
    for _ in range(args.rounds):
        for name in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                launch(name)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / args.reps)
    print(json.dumps({"channels": cin, "rows": n, "scenes": B,
                      "ms_per_launch_median": {k: round(float(np.median(v)), 4) for k, v in times.items()},
                      "ms_all_rounds": {k: [round(t, 4) for t in v] for k, v in times.items()}, "bit_identical_to_main": equal}), flush=True)
