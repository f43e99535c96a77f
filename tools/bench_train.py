#!/usr/bin/env python3
"""Forward + backward of VoxelResBackBone8x in train mode (BASELINE.json configs[4]: the backbone part of the
self-training step) on B synthetic scenes; HIP events.  Development tool."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4); ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--sweeps", type=int, default=1, help="10: the density transfusion_lidar.yaml trains on (nuscenes_dataset.yaml:5 MAX_SWEEPS 10): ~300 k points, ~150 k voxels per scene")
ap.add_argument("--amp", action="store_true", help="the reference's AMP recipe (tools/train_utils/train_utils.py:135-176): autocast(fp16) around the forward, "
                "GradScaler, unscale_, clip_grad_norm_(10), scaler.step / update")
args = ap.parse_args()
dev = torch.device("cuda", 0); B = args.batch
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": args.dtype}, 5, grid), 0).to(dev)
pts, off = syn.make_sweeps_batch(list(range(B)), args.sweeps) if args.sweeps > 1 else syn.make_batch(list(range(B)))
# (transfusion_lidar.yaml:54-59: MAX_NUMBER_OF_VOXELS 120 k train / 160 k test; the 10-sweep scenes hold ~150 k cells: the training cap fires)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 120000 if args.sweeps > 1 else 160000)
vox = S.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), B, cfg)
n = int(vox["n"].item())
bd = lambda: {"voxel_features": vox["mean"][:n], "voxel_coords": vox["coords"][:n].float(), "batch_size": B}
opt = torch.optim.SGD(net.parameters(), lr=1e-4)

def fwd(all_outputs=True):
    # the stand-in loss: mean of squares.  all_outputs: over the four multi-scale tensors and the encoded tensor (`train_step_ms`, every
    # round's definition); False: over `encoded_spconv_tensor` alone — the only backbone output transfusion_lidar.yaml's graph consumes
    # (HeightCompression, height_compression.py:20-21; `multi_scale_3d_features` feed PV-RCNN / Voxel R-CNN heads only)
    out = net(bd())
    ts = (list(out["multi_scale_3d_features"].values()) if all_outputs else []) + [out["encoded_spconv_tensor"]]
    return sum((t.features.float() ** 2).mean() for t in ts)

def timed(fn, reps):
    # MEDIAN of the steps' own durations (events around every step, steps issued back to back): one step in a few dozen stalls for
    # 50-100 ms on this pool (allocator growth, a host hiccup) and would otherwise decide the mean of five
    for _ in range(2): fn()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda.synchronize(); evs[0].record()
    for i in range(reps):
        fn(); evs[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]))

scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 12) if args.amp else None

def step(all_outputs=True):
    opt.zero_grad(set_to_none=True)
    if scaler is None:
        loss = fwd(all_outputs); loss.backward(); opt.step()
        return
    with torch.autocast("cuda", dtype=torch.float16):
        loss = fwd(all_outputs)
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)      # optim_cfg.GRAD_NORM_CLIP of the nuScenes configs
    scaler.step(opt)
    scaler.update()

net.train()
ms_step = timed(step, args.reps)
ms_step_enc = timed(lambda: step(False), args.reps)
# what the stand-in loss itself costs (forward + backward of the five mean-of-squares on detached copies of the outputs)
with torch.no_grad():
    o = net(bd())
leaves = [t.features.detach().clone().requires_grad_(True) for t in list(o["multi_scale_3d_features"].values()) + [o["encoded_spconv_tensor"]]]
def loss_only():
    for l in leaves: l.grad = None
    sum((l.float() ** 2).mean() for l in leaves).backward()
ms_loss = timed(loss_only, args.reps)
del o, leaves
with torch.no_grad():
    ms_fwd_train = timed(fwd, args.reps)          # module path, BN batch statistics, no graph
net.eval()
with torch.no_grad():
    ms_fwd_eval = timed(lambda: net(bd()), args.reps)   # fused inference path
print(json.dumps({"batch": B, "sweeps": args.sweeps, "amp": bool(args.amp), "points": int(pts.shape[0]), "voxels": n, "dtype": "fp16 autocast" if args.amp else args.dtype,
                  "train_step_ms": round(ms_step, 2),
                  "train_step_ms_loss_on_encoded_tensor_only": round(ms_step_enc, 2), "stand_in_loss_alone_ms": round(ms_loss, 2),
                  "module_forward_ms": round(ms_fwd_train, 2),
                  "fused_eval_forward_ms": round(ms_fwd_eval, 2), "scenes_per_s_train": round(B / ms_step * 1e3, 1)}))
