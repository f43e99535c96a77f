#!/usr/bin/env python3
"""Development (GPU box): same-process A/B of fnp_bn_train_forward / fnp_bn_train_backward across builds of the library
(the shipped one, tools/build_variant.sh variants by name, or any .so by path), on the row counts and channel widths of the
backbone's stages.  usage: tools/ab_bnorm.py --variants name_or_path,... [--dtype bf16|fp16] [--scale 1.0]"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import lib as _l

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default=""); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
td = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
libs = {"main": _l.load()}
for v in [v for v in args.variants.split(",") if v]:
    path = v if v.endswith(".so") else os.path.join(ROOT, "findnpropagate_amd", "csrc", "ab", f"libfnp_{v}.so")
    libs[os.path.basename(os.path.dirname(path)) + "/" + os.path.basename(path) if v.endswith(".so") else v] = ctypes.CDLL(path)
for L in libs.values():
    L.fnp_bn_train_forward.restype = ctypes.c_int; L.fnp_bn_train_backward.restype = ctypes.c_int
    L.fnp_bn_workspace_bytes.restype = ctypes.c_longlong
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
stream = P(torch.cuda.current_stream(dev).cuda_stream)
# (rows, channels) of the four stages at 16 single-sweep scenes
for rows, C in [(int(301444 * args.scale), 16), (int(560000 * args.scale), 32), (int(330000 * args.scale), 64), (int(136000 * args.scale), 128)]:
    g = torch.Generator(device=dev).manual_seed(rows + C)
    x = torch.randn((rows, C), device=dev, generator=g).to(td)
    res = torch.randn((rows, C), device=dev, generator=g).to(td)
    dy = torch.randn((rows, C), device=dev, generator=g).to(td)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    n_dev = torch.full((1,), rows, dtype=torch.int32, device=dev)
    code = _l.dtype_code(x)
    out = {}
    for name, L in libs.items():
        ws = torch.empty((int(L.fnp_bn_workspace_bytes(I(C))),), dtype=torch.uint8, device=dev)
        o = dict(y=torch.empty_like(x), mean=torch.empty(C, device=dev), invstd=torch.empty(C, device=dev), dx=torch.empty_like(x), dres=torch.empty_like(x),
                 dgamma=torch.empty(C, device=dev), dbeta=torch.empty(C, device=dev), rm=torch.zeros(C, device=dev), rv=torch.ones(C, device=dev), ws=ws)
        out[name] = o

    def fwd(name):
        L, o = libs[name], out[name]
        rc = L.fnp_bn_train_forward(P(x.data_ptr()), I(code), P(n_dev.data_ptr()), I(rows), I(C), P(gamma.data_ptr()), P(beta.data_ptr()), P(o["rm"].data_ptr()),
                                    P(o["rv"].data_ptr()), F(0.01), F(1e-3), P(res.data_ptr()), I(1), P(o["y"].data_ptr()), P(o["mean"].data_ptr()),
                                    P(o["invstd"].data_ptr()), P(None), P(o["ws"].data_ptr()), ctypes.c_longlong(o["ws"].numel()), stream)
        assert rc == 0, (name, rc)

    def bwd(name):
        L, o = libs[name], out[name]
        rc = L.fnp_bn_train_backward(P(dy.data_ptr()), P(x.data_ptr()), P(o["y"].data_ptr()), I(code), P(n_dev.data_ptr()), I(rows), I(C), P(gamma.data_ptr()),
                                     P(o["mean"].data_ptr()), P(o["invstd"].data_ptr()), I(1), P(o["dx"].data_ptr()), P(o["dres"].data_ptr()), P(o["dgamma"].data_ptr()),
                                     P(o["dbeta"].data_ptr()), P(o["ws"].data_ptr()), ctypes.c_longlong(o["ws"].numel()), stream)
        assert rc == 0, (name, rc)

    for name in libs:
        for _ in range(3):
            fwd(name); bwd(name)
    torch.cuda.synchronize()
    same = {name: {k: bool(torch.equal(out[name][k], out["main"][k])) for k in ("y", "mean", "invstd", "dx", "dgamma", "dbeta")} for name in libs}
    maxdiff = {name: {k: float((out[name][k].float() - out["main"][k].float()).abs().max()) for k in ("mean", "invstd", "dgamma", "dbeta")} for name in libs}
    times = {(name, d): [] for name in libs for d in ("fwd", "bwd")}
    for _ in range(args.rounds):
        for name in libs:
            for d, fn in (("fwd", fwd), ("bwd", bwd)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn(name)
                e1.record(); torch.cuda.synchronize()
                times[(name, d)].append(e0.elapsed_time(e1) / args.reps * 1e3)
    print(json.dumps({"rows": rows, "C": C, "dtype": args.dtype, "us_median": {f"{n}:{d}": round(float(np.median(v)), 1) for (n, d), v in times.items()},
                      "equal_to_main": same, "maxdiff": maxdiff}), flush=True)
