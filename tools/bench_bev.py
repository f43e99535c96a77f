#!/usr/bin/env python3
"""Secondary measurement ((f)1): the hand-off from the sparse backbone to the dense BEV backbone on B synthetic scenes —
the reference's way (SparseConvTensor.dense() -> view -> ZeroPad2d + Conv2d(256 -> 128, 3x3) + BatchNorm2d + ReLU as torch /
MIOpen modules) against the first block evaluated on the sparse rows (fnp_rulebook_strided + fnp_spconv_forward with the
BatchNorm / ReLU epilogue + fnp_sparse_to_dense_fill).  One JSON line."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_2d import BaseBEVBackbone, HeightCompression
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0); B = args.batch
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net3d = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
bev = BaseBEVBackbone({"LAYER_NUMS": [5, 5], "LAYER_STRIDES": [1, 2], "NUM_FILTERS": [128, 256], "UPSAMPLE_STRIDES": [1, 2],
                       "NUM_UPSAMPLE_FILTERS": [256, 256], "USE_CONV_FOR_NO_STRIDE": True}, 256).to(dev).eval()
hc = HeightCompression({"NUM_BEV_FEATURES": 256, "REUSE_OUTPUT": True})
pts, off = syn.make_batch(list(range(B)))
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
with torch.no_grad():
    r = net3d.forward_points(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), B, cfg)
t = r["out"]
t = t.replace_feature(t.features.float())      # the reference boundary: f32 rows


def timed(fn):
    with torch.no_grad():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(args.reps): out = fn()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps, out


def dense_way():
    d = hc({"encoded_spconv_tensor": t, "encoded_spconv_tensor_stride": 8})["spatial_features"]
    return bev.blocks[0][:4](d)


ms_dense_only, _ = timed(lambda: hc({"encoded_spconv_tensor": t, "encoded_spconv_tensor_stride": 8})["spatial_features"])
ms_dense, want = timed(dense_way)
ms_sparse, got = timed(lambda: bev.first_block_from_sparse(t))
t16 = t.replace_feature(t.features.bfloat16())
ms_sparse16, got16 = timed(lambda: bev.first_block_from_sparse(t16))
err = float((got - want).abs().max() / want.abs().max().clamp(min=1.0))
print(json.dumps({"workload": "first BaseBEVBackbone block behind the sparse backbone: (B,128,2,180,180) rows -> (B,128,180,180) f32",
                  "scenes": B, "encoded_rows": int(t.features.shape[0]), "reference_way_ms": {"dense_map_write": round(ms_dense_only, 4), "dense_map + torch conv/bn/relu (f32)": round(ms_dense, 4)},
                  "from_sparse_rows_ms": {"f32_rows": round(ms_sparse, 4), "bf16_rows": round(ms_sparse16, 4)},
                  "speedup_f32": round(ms_dense / ms_sparse, 2), "max_rel_err_f32_vs_torch": err}))
