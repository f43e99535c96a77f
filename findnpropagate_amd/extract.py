"""Sharded pseudo-label extraction (BASELINE.json configs[3]): scenes sharded across the GPUs of the node,
ONE fixed-shape all-gather of the pseudo-label boxes per step.

The reference's tools/extract_pseudo_labels.py is single process, batch size 1 (asserted at
:36,54) and writes `<frame_id with . -> _>.pth` = torch.save([pred_dict]) per frame (:134-137);
its multi-GPU helpers exchange results through pickled byte tensors or files
(pcdet/utils/commu_utils.py:50-111, common_utils.py:229-250).  Here:

  * rank r takes scenes r, r+W, r+2W, ... exactly like pcdet's DistributedSampler
    (pcdet/datasets/__init__.py:31-51: wrap-around padding so every rank runs the same number
    of steps, no shuffle); a step covers `scenes_per_step` of them in one Box Seeker launch;
  * every step each rank contributes ONE fixed-shape record per scene,
        (K_MAX + 1, 9) f32:  row 0 = [count, dataset index (-1 = nothing to write), 0 ...]
                             rows 1.. = [x, y, z, dx, dy, dz, yaw, score, label]
    to a single `all_gather_into_tensor` (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests) —
    no pickling, no per-rank size exchange, no second collective for the counts, no files;
  * with the fused head (FrustumProposerOG.launch) on a GPU the loop is a pipeline without host
    synchronisation inside a step: the record is packed on the device, the collective and the
    device->host copy of the gathered records run on a side stream, and step k's files are written
    by a writer thread while step k+1 computes; the host waits only for the copy of the previous step;
  * the gathered boxes are written in the reference's on-disk format (by rank 0, or by every
    rank for its own frames with write="own"), so pseudo_loader.py:561-679 reads them unchanged;
    frames whose file already exists are skipped (the reference refuses to run at all when the
    folder exists, :80-84).
"""
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import torch

K_MAX = 256          # boxes per scene in the exchange record (a scene yields tens)
RECORD_WIDTH = 9
RECORD_ROWS = K_MAX + 1


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
    """pred_dict (pred_boxes (K,7), pred_scores (K,), pred_labels (K,)) + dataset index -> (K_MAX + 1, 9) f32."""
    rec = torch.zeros((RECORD_ROWS, RECORD_WIDTH), dtype=torch.float32, device=device)
    k = int(pred_dict["pred_boxes"].shape[0])
    if k > K_MAX:
        raise ValueError(f"{k} boxes in one scene exceed the exchange record ({K_MAX}); raise extract.K_MAX")
    if abs(int(index)) >= 1 << 24:
        raise ValueError("dataset index does not fit the f32 record header")
    if k:
        rec[1:k + 1, :7] = pred_dict["pred_boxes"].to(device=device, dtype=torch.float32)
        rec[1:k + 1, 7] = pred_dict["pred_scores"].to(device=device, dtype=torch.float32)
        rec[1:k + 1, 8] = pred_dict["pred_labels"].to(device=device, dtype=torch.float32)
    rec[0, 0], rec[0, 1] = float(k), float(index)
    return rec


def unpack_record(rec):
    """Inverse of pack_record (rec on any device) -> (pred_dict on CPU in the reference's dtypes, dataset index)."""
    rec = rec.cpu()
    k, index = int(rec[0, 0]), int(rec[0, 1])
    body = rec[1:k + 1]
    return {"pred_boxes": body[:, :7].contiguous(), "pred_scores": body[:, 7].contiguous(),
            "pred_labels": body[:, 8].to(torch.int32)}, index


def all_gather_records(rec, dist=None, async_op=False, force_collective=False):
    """The step's one collective.  rec (S, K_MAX + 1, 9) -> (W, S, K_MAX + 1, 9) [, work handle].
    A process group of one rank has nothing to exchange and skips it — unless force_collective: the collective is then
    issued all the same (RCCL with backend "nccl"), so that the path a multi-GPU run takes — communicator set-up, the
    collective on the side stream, the consumer one step later — executes on a single GPU too."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        out = rec[None]
        return (out, None) if async_op else out
    W = dist.get_world_size()
    flat = rec.reshape(-1, RECORD_WIDTH)
    # concatenated (W*rows, 9) output: the form both RCCL and gloo implement
    out = torch.empty((W * flat.shape[0], RECORD_WIDTH), dtype=rec.dtype, device=rec.device)
    work = dist.all_gather_into_tensor(out, flat, async_op=async_op)
    out = out.view(W, *rec.shape)
    return (out, work) if async_op else out


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
_SIDE_STREAMS = {}                # (device, caller's stream) -> the pipeline's side stream


def recall_keys():
    keys = ["gt", "num_3known", "num_6known", "num_4unknown", "num_7unknown"]
    for t in RECALL_THRESH:
        keys += [stem % str(t) for stem in ("roi_%s", "rcnn_%s", "rcnn_3known_%s", "rcnn_6known_%s", "rcnn_4unknown_%s",
                                            "rcnn_7unknown_%s")]
    return keys


def collate_scenes(scenes):
    """Default collate of single-scene batch_dicts into one batch (dataset.py:221-344 for the keys the Box
    Seeker reads): points get their scene number in column 0, (1, ...) tensors are concatenated, 'dets'
    (the 5-tuple a PreprocessedGLIP-like detector returns) are concatenated with the scene number as batch
    index, 'frame_id' / 'gt_boxes' become lists, 'points_per_scene' (host ints) spares the head a sync."""
    if len(scenes) == 1:
        d = dict(scenes[0])
        d["frame_id"] = [d["frame_id"]]
        if "gt_boxes" in d:
            d["gt_boxes_list"] = [d["gt_boxes"][0]]
        return d
    out = {"batch_size": len(scenes), "frame_id": [s["frame_id"] for s in scenes]}
    pts = []
    for b, s in enumerate(scenes):
        p = s["points"].clone()
        p[:, 0] = b
        pts.append(p)
    out["points"] = torch.cat(pts)
    out["points_per_scene"] = [int(p.shape[0]) for p in pts]
    for k in ("camera_intrinsics", "camera2lidar", "lidar2image", "lidar_aug_matrix", "img_aug_matrix"):
        if k in scenes[0]:
            out[k] = torch.cat([s[k] for s in scenes])
    if "dets" in scenes[0]:
        parts = [s["dets"] for s in scenes]
        out["dets"] = tuple(torch.cat([torch.full_like(p[i], b) if i == 3 else p[i] for b, p in enumerate(parts)]) for i in range(5))
    if "gt_boxes" in scenes[0]:
        out["gt_boxes_list"] = [s["gt_boxes"][0] for s in scenes]
    for k in ("metadata", "image_paths"):
        if k in scenes[0]:
            out[k] = [v for s in scenes for v in s[k]]
    return out


def _records_from_launch(r, index_tags, device):
    """Pack the fused head's fixed-shape output into (S, K_MAX + 1, 9) records ON THE DEVICE with one launch
    (fnp_seeker_pack_records; no host sync): a box's row = number of valid frustums of its scene before it."""
    import ctypes

    from . import lib as _l

    S = len(index_tags)
    rec = torch.empty((S, RECORD_ROWS, RECORD_WIDTH), dtype=torch.float32, device=device)
    tags = (ctypes.c_float * S)(*[float(t) for t in index_tags])
    if r is not None:
        fr = r["frustums"]                                   # host (F, 8)
        most = int(fr.shape[0]) if S == 1 else int(torch.bincount(fr[:, 0].long(), minlength=S).max())   # (one scene per step: the extraction script's)
        if most > K_MAX:
            raise ValueError(f"{most} frustums in one scene exceed the exchange record ({K_MAX}); raise extract.K_MAX")
        args = (_l.ptr(r["d_frustums"]), _l.ptr(r["out_valid"]), _l.ptr(r["out_box"]), int(fr.shape[0]))
    else:
        args = (None, None, None, 0)
    rc = _l.load().fnp_seeker_pack_records(*args, ctypes.cast(tags, ctypes.c_void_p), S, RECORD_ROWS, _l.ptr(rec), _l.stream())
    _l.check(rc, "fnp_seeker_pack_records")
    return rec


class _Writer:
    """File writes of the gathered records on one background thread (torch.save + rename release the GIL in IO)."""

    def __init__(self, out_dir, dataset, resume, enabled=True):
        self.out_dir, self.dataset, self.resume = out_dir, dataset, resume
        self.pool = ThreadPoolExecutor(max_workers=1) if enabled else None
        self.futures = []
        self.written = 0
        self.seen = set()

    def _fid(self, idx):
        return self.dataset.frame_id(idx) if hasattr(self.dataset, "frame_id") else self.dataset[idx]["frame_id"]

    def _write(self, recs):
        n = 0
        for rec in recs:
            pd, idx = unpack_record(rec)
            if idx < 0 or idx in self.seen:
                continue                              # skipped frames, wrap-around duplicates
            self.seen.add(idx)
            fid = self._fid(idx)
            if self.resume and frame_path(self.out_dir, fid).exists():
                continue                              # earlier runs
            save_frame(self.out_dir, fid, pd)
            n += 1
        return n

    def submit(self, recs_cpu):
        """recs_cpu: (n, K_MAX + 1, 9) host tensor whose contents are final."""
        if self.pool is None:
            self.written += self._write(recs_cpu)
        else:
            self.futures.append(self.pool.submit(self._write, recs_cpu))

    def close(self):
        for f in self.futures:
            self.written += f.result()
        self.futures = []
        if self.pool is not None:
            self.pool.shutdown()
        return self.written


def extract_pseudo_labels(dataset, head, out_dir, device, dist=None, write="rank0", resume=True, progress=None,
                          recall=None, recall_fn=None, scenes_per_step=1, pipeline=None, collate=None, force_collective=False,
                          trace=None):
    """Run `head` (a FrustumProposerOG-like module: forward(batch_dict) -> batch_dict with
    'final_box_dicts') over `dataset` sharded across the process group.

    dataset: len(), __getitem__(i) -> batch_dict for ONE scene (batch_size 1; tensors on `device` or
             CPU) with 'frame_id' (str) — the collated form extract_pseudo_labels.py:115 iterates.
    scenes_per_step: scenes a rank runs per launch (the reference's script is fixed at 1); they are merged by
             `collate` (default collate_scenes).
    pipeline: None = automatic (on when `head` has the fused sync-free `launch` and `device` is a GPU): record packing
             on the device, collective + device->host copy on a side stream, files written one step later by a thread.
    recall:  optional dict; when given and the scenes carry 'gt_boxes', the running recall of
             extract_pseudo_labels.py:108-131 is kept (Detector3DTemplate.generate_recall_record per
             frame, thresholds 0.3/0.5/0.7; in pipeline mode as a device-resident counter vector that is read once at
             the end) and summed over the ranks with one all-reduce of the counter vector (wrap-around
             duplicates are counted once); `recall` then holds the totals plus 'recall_<thr>' = rcnn_<thr> / gt.
             recall_fn: the record function (default Detector3DTemplate.generate_recall_record; the CPU tests
             inject a counter); a custom one runs on host pred_dicts, i.e. outside the pipeline.
    force_collective: issue the step's all-gather even in a process group of one rank (see all_gather_records).
    trace:   optional list; the pipeline appends ("launch", step), ("collective", step) and ("consume", step) as it goes — the
             order a test reads the one-step-late consumption from.
    Returns the number of frames this rank wrote.
    """
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    os.makedirs(out_dir, exist_ok=True)
    out_dir_s = os.fspath(out_dir)
    n = len(dataset)
    S = max(1, int(scenes_per_step))
    collate = collate or collate_scenes
    device = torch.device(device)
    if pipeline is None:
        pipeline = device.type == "cuda" and hasattr(head, "launch") and recall_fn is None
    # every rank walks its shard S scenes at a time; global position of (step, slot, rank) decides duplicates
    mine = shard_indices(n, rank, world)
    steps = [mine[i:i + S] for i in range(0, len(mine), S)]
    keys = recall_keys()
    rec_local = {}
    rec_vec = torch.zeros((len(keys),), dtype=torch.int64, device=device)
    writer = _Writer(out_dir, dataset, resume, enabled=pipeline)
    # (a stream that was SEEN to run beside the caller's: HIP shares four hardware queues out in the order streams are first used, and
    #  in a fresh process torch's first pool streams sit in the default stream's queue — the collective and the copy to the host would
    #  then run behind the next scene's launch instead of beside it; sparse.concurrent_streams)
    side = None
    if pipeline:
        from .sparse import concurrent_streams
        cur0 = torch.cuda.current_stream(device)
        skey = (str(device), cur0.cuda_stream)
        side = _SIDE_STREAMS.get(skey)      # (tested once per process and caller stream: the test costs a few milliseconds)
        if side is None:
            side = _SIDE_STREAMS[skey] = concurrent_streams(device, 1, beside=[cur0])[0]
    pending = None       # (host records, event) of the previous step
    head.eval()

    def writes_for(recs_w):
        """Rows of the gathered (W, S, R, 9) records this rank is responsible for, as one (n, R, 9) tensor."""
        if write == "own":
            return recs_w[rank]
        if write == "rank0":
            return recs_w.reshape(-1, RECORD_ROWS, RECORD_WIDTH) if rank == 0 else recs_w[:0].reshape(0, RECORD_ROWS, RECORD_WIDTH)
        raise ValueError(write)

    def drain(p):
        host, ev, st = p
        ev.synchronize()                              # the only host wait of the pipeline: previous step's copy
        if trace is not None:
            trace.append(("consume", st))
        writer.submit(host)

    with torch.no_grad():
        for step, idxs in enumerate(steps):
            scenes, tags, dup = [], [], []
            for slot, index in enumerate(idxs):
                data = dataset[index]
                skip = resume and os.path.exists(os.path.join(out_dir_s, str(data["frame_id"]).replace('.', '_') + ".pth"))   # (frame_path, as a string)
                if skip:
                    continue
                scenes.append(data)
                tags.append(index)
                dup.append((step * S + slot) * world + rank >= n)      # wrap-around padding of the sampler
            # records always have S slots so that every rank contributes the same shape
            slot_tags = tags + [-1] * (S - len(tags))
            if pipeline:
                if scenes:
                    batch = collate(scenes)
                    r = head.launch(batch)
                    if trace is not None:
                        trace.append(("launch", step))
                    rec = _records_from_launch(r, slot_tags, device)
                    if recall is not None and "gt_boxes_list" in batch:
                        from .detectors import Detector3DTemplate
                        for b, g in enumerate(batch["gt_boxes_list"]):
                            if dup[b]:
                                continue
                            Detector3DTemplate.recall_counter_vector(rec[b, 1:], g.to(device), list(RECALL_THRESH),
                                                                     pred_count=rec[b, 0, 0:1], out=rec_vec)
                else:
                    rec = _records_from_launch(None, slot_tags, device)
                cur = torch.cuda.current_stream(device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    recs_w, work = all_gather_records(rec, dist, async_op=True, force_collective=force_collective)
                    if trace is not None:
                        trace.append(("collective", step))
                    if work is not None:
                        work.wait()                   # stream-level wait (side stream), not a host wait
                    mine_w = writes_for(recs_w)
                    host = torch.empty(mine_w.shape, dtype=mine_w.dtype, pin_memory=True)
                    host.copy_(mine_w, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                rec.record_stream(side)
                recs_w.record_stream(side)
                if pending is not None:
                    drain(pending)
                pending = (host, ev, step)
            else:
                recs = []
                if scenes:
                    batch = collate(scenes)
                    preds = head.forward(batch)["final_box_dicts"]
                    for b, pred in enumerate(preds):
                        if recall is not None and "gt_boxes_list" in batch and not dup[b]:
                            fn = recall_fn
                            if fn is None:
                                from .detectors import Detector3DTemplate
                                fn = Detector3DTemplate.generate_recall_record
                            rec_local = fn(pred["pred_boxes"], rec_local, 0, {"gt_boxes": batch["gt_boxes_list"][b][None]},
                                           thresh_list=list(RECALL_THRESH))
                        recs.append(pack_record(pred, tags[b], device))
                empty = {"pred_boxes": torch.zeros((0, 7)), "pred_scores": torch.zeros((0,)), "pred_labels": torch.zeros((0,), dtype=torch.int32)}
                while len(recs) < S:
                    recs.append(pack_record(empty, -1, device))   # tells the writers to leave an existing file alone
                recs_w = all_gather_records(torch.stack(recs), dist, force_collective=force_collective)
                writer.submit(writes_for(recs_w).cpu())
            if progress is not None:
                progress(step, len(steps))
    if pending is not None:
        drain(pending)
    written = writer.close()
    if recall is not None:
        if not pipeline:
            rec_vec = torch.tensor([int(rec_local.get(k, 0)) for k in keys], dtype=torch.int64, device=device)
        if dist is not None and dist.is_initialized() and world > 1:
            dist.all_reduce(rec_vec)
        recall.clear()
        recall.update({k: int(v) for k, v in zip(keys, rec_vec.cpu().tolist())})
        for t in RECALL_THRESH:
            recall["recall_%s" % str(t)] = recall["rcnn_%s" % str(t)] / max(recall["gt"], 1)
    if dist is not None and dist.is_initialized():
        dist.barrier()
    return written
