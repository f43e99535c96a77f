#!/usr/bin/env python3
"""Time HeightCompression's densification on the backbone's real output:
tiled single-pass writer vs torch.zeros + row scatter.  Development tool."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn, lib as _l
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0); B = args.batch
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B)))
pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
with torch.no_grad():
    t = net.forward_points(pts, off, B, cfg)["out"]
f, idx, n_dev = t.features.contiguous(), t.indices, t.n_dev()
shape = list(t.spatial_shape); C = f.shape[1]
L = _l.load()
ws = torch.empty((int(L.fnp_sparse_to_dense_workspace_bytes(B, *shape)),), dtype=torch.uint8, device=dev)
out = torch.empty((B, C, *shape), dtype=f.dtype, device=dev)

def timed(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(args.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps

def scatter():
    o = torch.zeros((B, C, *shape), dtype=f.dtype, device=dev)
    rc = L.fnp_sparse_to_dense(_l.ptr(f), _l.dtype_code(f), _l.ptr(idx), _l.ptr(n_dev), idx.shape[0], C, B, *shape, _l.ptr(o), None, 0, _l.stream())
    assert rc == 0
    return o

ms_t = timed(lambda: S.to_dense(f, idx, n_dev, B, shape, workspace=ws, out=out))
ms_s = timed(scatter)
assert torch.equal(S.to_dense(f, idx, n_dev, B, shape, workspace=ws, out=out), scatter())
byts = out.numel() * out.element_size()
print(json.dumps({"batch": B, "sites": int(f.shape[0]), "dense_MB": round(byts / 1e6, 1), "tiled_ms": round(ms_t, 4), "tiled_GBs": round(byts / ms_t / 1e6, 1),
                  "memset_scatter_ms": round(ms_s, 4), "speedup": round(ms_s / ms_t, 2)}))
