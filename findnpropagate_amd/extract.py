"""Sharded pseudo-label extraction (BASELINE.json configs[3]): scenes 1-per-GPU across the node,
one fixed-shape all-gather of the pseudo-label boxes per step.

The reference's tools/extract_pseudo_labels.py is single process, batch size 1 (asserted at
:36,54) and writes `<frame_id with . -> _>.pth` = torch.save([pred_dict]) per frame (:134-137);
its multi-GPU helpers exchange results through pickled byte tensors or files
(pcdet/utils/commu_utils.py:50-111, common_utils.py:229-250).  Here:

  * rank r takes scenes r, r+W, r+2W, ... exactly like pcdet's DistributedSampler
    (pcdet/datasets/__init__.py:31-51: wrap-around padding so every rank runs the same number
    of steps, no shuffle);
  * every step each rank runs the Greedy Box Seeker on its scene(s) and contributes ONE
    fixed-shape record  (K_MAX, 9) f32 = [x, y, z, dx, dy, dz, yaw, score, label]  plus
    (count, dataset index) to a single `all_gather_into_tensor` (RCCL over xGMI with backend
    "nccl"; "gloo" in the CPU tests) — no pickling, no per-rank size exchange, no files;
  * the gathered boxes are written in the reference's on-disk format (by rank 0, or by every
    rank for its own frames with write="own"), so pseudo_loader.py:561-679 reads them unchanged;
    frames whose file already exists are skipped (the reference refuses to run at all when the
    folder exists, :80-84).
"""
import os
from pathlib import Path

import torch

K_MAX = 256          # boxes per scene in the exchange record (a scene yields tens)
RECORD_WIDTH = 9


def shard_indices(n_items, rank, world_size):
    """pcdet.datasets.DistributedSampler.__iter__ with shuffle=False (datasets/__init__.py:43-51)."""
    if n_items == 0:
        return []
    num_samples = (n_items + world_size - 1) // world_size
    total = num_samples * world_size
    indices = list(range(n_items))
    while len(indices) < total:                     # wrap-around padding (also when world_size > n_items)
        indices += indices[:total - len(indices)]
    return indices[rank:total:world_size]


def pack_record(pred_dict, index, device):
    """pred_dict (pred_boxes (K,7), pred_scores (K,), pred_labels (K,)) -> (K_MAX, 9) f32, (2,) i64."""
    rec = torch.zeros((K_MAX, RECORD_WIDTH), dtype=torch.float32, device=device)
    k = int(pred_dict["pred_boxes"].shape[0])
    if k > K_MAX:
        raise ValueError(f"{k} boxes in one scene exceed the exchange record ({K_MAX}); raise extract.K_MAX")
    if k:
        rec[:k, :7] = pred_dict["pred_boxes"].to(device=device, dtype=torch.float32)
        rec[:k, 7] = pred_dict["pred_scores"].to(device=device, dtype=torch.float32)
        rec[:k, 8] = pred_dict["pred_labels"].to(device=device, dtype=torch.float32)
    meta = torch.tensor([k, index], dtype=torch.int64, device=device)
    return rec, meta


def unpack_record(rec, meta):
    """Inverse of pack_record -> (pred_dict on CPU in the reference's dtypes, dataset index)."""
    k, index = int(meta[0]), int(meta[1])
    rec = rec[:k].cpu()
    return {"pred_boxes": rec[:, :7].contiguous(), "pred_scores": rec[:, 7].contiguous(),
            "pred_labels": rec[:, 8].to(torch.int32)}, index


def all_gather_records(rec, meta, dist=None):
    """One collective for the boxes, one tiny one for (count, index).  Returns (W,K_MAX,9), (W,2)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rec[None], meta[None]
    W = dist.get_world_size()
    # concatenated (W*K_MAX, 9) output: the form both RCCL and gloo implement
    out = torch.empty((W * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    metas = torch.empty((W * 2,), dtype=meta.dtype, device=meta.device)
    dist.all_gather_into_tensor(out, rec)
    dist.all_gather_into_tensor(metas, meta)
    return out.view(W, *rec.shape), metas.view(W, 2)


def frame_path(out_dir, frame_id):
    """extract_pseudo_labels.py:134."""
    return Path(out_dir) / f"{str(frame_id).replace('.', '_')}.pth"


def save_frame(out_dir, frame_id, pred_dict):
    """torch.save([pred_dict], <frame>.pth) — the list-of-one-dict the reference writes (:137)."""
    path = frame_path(out_dir, frame_id)
    tmp = path.with_suffix(".pth.tmp%d" % os.getpid())
    torch.save([pred_dict], tmp)
    os.replace(tmp, path)
    return path


RECALL_THRESH = (0.3, 0.5, 0.7)   # extract_pseudo_labels.py:108


def extract_pseudo_labels(dataset, head, out_dir, device, dist=None, write="rank0", resume=True, progress=None,
                          recall=None, recall_fn=None):
    """Run `head` (a FrustumProposerOG-like module: forward(batch_dict) -> batch_dict with
    'final_box_dicts') over `dataset` sharded across the process group.

    dataset: len(), __getitem__(i) -> batch_dict for ONE scene (batch_size 1; tensors on `device` or
             CPU) with 'frame_id' (str) — the collated form extract_pseudo_labels.py:115 iterates.
    recall:  optional dict; when given and the scenes carry 'gt_boxes', the running recall of
             extract_pseudo_labels.py:108-131 is kept (Detector3DTemplate.generate_recall_record per
             frame, thresholds 0.3/0.5/0.7) and, at the end, summed over the ranks with one all-reduce
             of the counter vector (wrap-around duplicates are counted once); `recall` then holds the
             totals plus 'recall_<thr>' = rcnn_<thr> / gt.  recall_fn: the record function
             (default Detector3DTemplate.generate_recall_record; the CPU tests inject a counter).
    Returns the number of frames this rank wrote.
    """
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    os.makedirs(out_dir, exist_ok=True)
    n = len(dataset)
    mine = shard_indices(n, rank, world)
    written = 0
    rec_local = {}
    head.eval()
    with torch.no_grad():
        for step, index in enumerate(mine):
            data = dataset[index]
            frame_id = data["frame_id"]
            skip = resume and frame_path(out_dir, frame_id).exists()
            if skip:
                pred = {"pred_boxes": torch.zeros((0, 7)), "pred_scores": torch.zeros((0,)), "pred_labels": torch.zeros((0,), dtype=torch.int32)}
                index_tag = -1                       # tells the writers to leave the existing file alone
            else:
                pred = head.forward(data)["final_box_dicts"][0]
                index_tag = index
                duplicate = step * world + rank >= n      # wrap-around padding of the sampler
                if recall is not None and "gt_boxes" in data and not duplicate:
                    if recall_fn is None:
                        from .detectors import Detector3DTemplate
                        recall_fn = Detector3DTemplate.generate_recall_record
                    rec_local = recall_fn(pred["pred_boxes"], rec_local, 0, data, thresh_list=list(RECALL_THRESH))
            rec, meta = pack_record(pred, index_tag, device)
            recs, metas = all_gather_records(rec, meta, dist)
            for r in range(recs.shape[0]):
                if write == "own" and r != rank:
                    continue
                if write == "rank0" and rank != 0:
                    continue
                pd, idx = unpack_record(recs[r], metas[r])
                if idx < 0:
                    continue
                fid = dataset.frame_id(idx) if hasattr(dataset, "frame_id") else dataset[idx]["frame_id"]
                if resume and frame_path(out_dir, fid).exists():
                    continue                          # wrap-around duplicates, earlier runs
                save_frame(out_dir, fid, pd)
                written += 1
            if progress is not None:
                progress(step, len(mine))
    if recall is not None:
        keys = ["gt", "num_3known", "num_6known", "num_4unknown", "num_7unknown"]
        for t in RECALL_THRESH:
            keys += [stem % str(t) for stem in ("roi_%s", "rcnn_%s", "rcnn_3known_%s", "rcnn_6known_%s", "rcnn_4unknown_%s",
                                                "rcnn_7unknown_%s")]
        vec = torch.tensor([float(rec_local.get(k, 0)) for k in keys], dtype=torch.float64, device=device)
        if dist is not None and dist.is_initialized() and world > 1:
            dist.all_reduce(vec)
        recall.clear()
        recall.update({k: int(v) for k, v in zip(keys, vec.cpu().tolist())})
        for t in RECALL_THRESH:
            recall["recall_%s" % str(t)] = recall["rcnn_%s" % str(t)] / max(recall["gt"], 1)
    if dist is not None and dist.is_initialized():
        dist.barrier()
    return written
