#!/usr/bin/env python3
"""Secondary measurement (BASELINE.json configs[3]): pseudo-label extraction end to end — Greedy Box Seeker per
scene (batch size 1, like tools/extract_pseudo_labels.py), fixed-shape all-gather of the boxes (when launched
under torch.distributed.run), running recall, `.pth` files in the reference's format.  One JSON line from rank 0.

    python tools/bench_extract.py --scenes 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/bench_extract.py --scenes 512
"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import extract as E, synthetic as syn
from findnpropagate_amd.dense_heads import FrustumProposerOG

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=64); ap.add_argument("--distinct", type=int, default=8); ap.add_argument("--per-step", type=int, default=1); ap.add_argument("--sync", action="store_true", help="plain synchronous loop instead of the pipeline")
ap.add_argument("--gpus", type=int, default=0, help="GPUs of the job (0 = take WORLD_SIZE); for N > 1 launch under torch.distributed.run --nproc-per-node N")
ap.add_argument("--force-collective", action="store_true", help="one rank: issue the RCCL all-gather all the same (init a one-rank nccl group)")
args = ap.parse_args()
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
if args.gpus and args.gpus != world:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch under torch.distributed.run --nproc-per-node {args.gpus}")
torch.cuda.set_device(local); dev = torch.device("cuda", local)
dist = None
if world > 1 or args.force_collective:
    import torch.distributed as dist
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=dev)
PARAMS = {'lq': 0.0, 'uq': 0.25, 'cq': 1.0, 'iou_w': 1.0, 'nms_normal': 1.0, 'dst_w': 0.0, 'dns_w': 1.0,
          'min_cam_iou': 0.3, 'score_thr': 0.45, 'nms_2d': 0.4, 'nms_3d': 0.0, 'clamp_bottom': 1, 'num_sizes': 1}


Scenes = lambda n, distinct: syn.SeekerScenes(n, distinct, dev)


data = Scenes(args.scenes, args.distinct)
cur = {}
head = FrustumProposerOG(model_cfg={"PARAMS": PARAMS, "PREDS_PATH": "PreprocessedGLIP", "BOX_FORMAT": "xyxy"},
                         image_detector=lambda bd: bd["dets"]).eval()
with tempfile.TemporaryDirectory() as warm:
    E.extract_pseudo_labels(Scenes(2 * world * args.per_step, args.distinct), head, warm, dev, dist=dist, write="own", scenes_per_step=args.per_step, pipeline=False if args.sync else None, force_collective=args.force_collective)
out_dir = tempfile.mkdtemp(prefix="fnp_extract_")
rec = {}
torch.cuda.synchronize(); t0 = time.perf_counter()
written = E.extract_pseudo_labels(data, head, out_dir, dev, dist=dist, write="own", recall=rec, scenes_per_step=args.per_step, pipeline=False if args.sync else None, force_collective=args.force_collective)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
if dist is not None:
    t = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
if rank == 0:
    print(json.dumps({"workload": "pseudo-label extraction: Box Seeker per scene + all-gather + recall + .pth", "n_gpus": world,
                      "scenes": args.scenes, "scenes_per_step": args.per_step, "pipeline": not args.sync, "collective": "rccl all_gather_into_tensor per step" if (world > 1 or args.force_collective) else "none (one rank)", "seconds": round(dt, 3), "scenes_per_s": round(args.scenes / dt, 1),
                      "ms_per_scene_per_gpu": round(1e3 * dt * world / args.scenes, 3),
                      "recall": {k: round(v, 3) for k, v in rec.items() if k.startswith("recall_")}, "gt": rec.get("gt")}))
import shutil; shutil.rmtree(out_dir, ignore_errors=True)
if dist is not None:
    dist.destroy_process_group()
