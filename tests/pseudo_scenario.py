"""Seeded inputs of the pseudo-label mixing parity tests (shared by tests/golden/make_pseudo_golden.py, which runs the
REFERENCE classes on them, and tests/test_pseudo_mixing.py, which runs this package's classes): frames with lidar
points, known-class ground truth, Box-Seeker pseudo-label files and last round's self-training files."""
import os

import numpy as np
import torch

from findnpropagate_amd import synthetic as syn

KNOWN = ['car', 'construction_vehicle', 'trailer', 'barrier', 'bicycle', 'pedestrian']      # the 6-known split
ALL = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle', 'bicycle', 'pedestrian', 'traffic_cone']
N_FRAMES = 5


def frame_id(i):
    return f"n015-2018-{i:04d}.pcd.bin"


def frame_path(folder, i):
    return os.path.join(folder, frame_id(i).replace('.', '_') + ".pth")


def make_frames(folder_frustum, folder_st, write=True):
    """-> list of per-frame dicts (frame_id, points (N,5) f32, gt_boxes (G,8) f32 [box, KNOWN-list label]); writes the
    frustum files (list of one dict: the extraction's format) and self-training files (plain dict with 'epoch')."""
    frames = []
    rng = np.random.default_rng(99)
    for i in range(N_FRAMES):
        pts, boxes, cls = syn.make_scene(200 + i, n_azimuth=400, n_boxes=24, return_boxes=True)
        names = [ALL[c] for c in cls]
        known = np.array([n in KNOWN for n in names])
        gt = np.zeros((int(known.sum()), 8), np.float32)
        gt[:, :7] = boxes[known]
        gt[:, 7] = [KNOWN.index(n) + 1 for n, k in zip(names, known) if k]
        frames.append({"frame_id": frame_id(i), "points": pts.astype(np.float32), "gt_boxes": gt})
        if not write:
            continue
        # Box Seeker output: the unknown-class objects (jittered), two known-class boxes, one box on top of a GT box,
        # one empty box
        unk = np.nonzero(~known)[0]
        pb = boxes[unk] + rng.normal(0, 0.05, size=(len(unk), 7)).astype(np.float32)
        pl = (cls[unk] + 1).astype(np.int32)
        extra = boxes[known][:3] + np.float32(0.02)
        el = np.array([ALL.index(n) + 1 for n, k in zip(names, known) if k][:3], np.int32)
        el[2] = 4                                                     # an unknown label sitting on a GT box: removed
        empty = np.array([[5.0, 5.0, 0.0, 0.0, 1.0, 1.0, 0.0]], np.float32)
        fb = np.concatenate([pb, extra, empty]).astype(np.float32)
        fl = np.concatenate([pl, el, np.array([2], np.int32)])
        fs = rng.uniform(0.3, 0.95, size=len(fb)).astype(np.float32)
        if i != 3:                                                    # frame 3 has no frustum file
            torch.save([{"pred_boxes": torch.from_numpy(fb), "pred_scores": torch.from_numpy(fs), "pred_labels": torch.from_numpy(fl)}],
                       frame_path(folder_frustum, i))
        # last round's self-training predictions: the unknown objects again (some shifted: duplicates for the NMS), a few
        # low scores, 9-column boxes (with velocity)
        sb = np.concatenate([boxes[unk] + rng.normal(0, 0.3, size=(len(unk), 7)).astype(np.float32),
                             syn.random_boxes(rng, 6, centre_range=30.0)])
        sl = np.concatenate([pl, rng.integers(1, 11, size=6).astype(np.int32)])
        ss = rng.uniform(0.02, 0.9, size=len(sb)).astype(np.float32)
        sb9 = np.concatenate([sb, rng.normal(size=(len(sb), 2)).astype(np.float32)], 1)
        if i != 1:                                                    # frame 1 has no self-training file yet
            torch.save({"pred_boxes": torch.from_numpy(sb9), "pred_scores": torch.from_numpy(ss), "pred_labels": torch.from_numpy(sl), "epoch": 3},
                       frame_path(folder_st, i))
    return frames


LOADER_CONFIGS = {
    "default": dict(ctor=dict(min_score=0.1, max_selftrain_per_class=3, mom=0.9), sampler=dict(max_queue_size_per_class=2)),
    "conf_fixcp_stonly": dict(ctor=dict(min_score=0.05, max_selftrain_per_class=None, mom=0.9997, fix_cp=3, copy_st_only=True,
                                        pseudo_nms_thresh=0.1),
                              sampler=dict(max_queue_size_per_class=3, queue_metric="conf", trans_noise=1.0, rot_noise=0.3)),
}


def run_loader(PseudoLoader, name, folder_frustum, folder_st, frames, epochs=2):
    """Drives load_frustum_pseudos -> load_selftrain_pseudos -> copy_and_paste over the frames `epochs` times with a
    seeded np.random and returns a flat dict of everything observable."""
    cfg = LOADER_CONFIGS[name]
    loader = PseudoLoader(known_class_names=KNOWN, pseudo_path=folder_frustum, self_train_path=folder_st, **cfg["ctor"])
    for k, v in cfg["sampler"].items():
        setattr(loader.sampler, k, v)
    np.random.seed(2024)
    out = {}
    step = 0
    for _ in range(epochs):
        for fr in frames:
            bd = {"frame_id": fr["frame_id"], "points": fr["points"].copy(), "gt_boxes": fr["gt_boxes"].copy()}
            bd = loader.load_frustum_pseudos(bd)
            out[f"{name}_s{step}_frustum_boxes"] = np.asarray(bd["pseudo_boxes"], np.float64)
            bd = loader.load_selftrain_pseudos(bd)
            out[f"{name}_s{step}_st_boxes"] = np.asarray(bd["pseudo_boxes"], np.float64)
            out[f"{name}_s{step}_st_scores"] = np.asarray(bd["pseudo_scores"], np.float64)
            out[f"{name}_s{step}_types"] = np.asarray(loader.pseudo_types, np.int64)
            bd = loader.copy_and_paste(bd)
            out[f"{name}_s{step}_boxes"] = np.asarray(bd["pseudo_boxes"], np.float64)
            out[f"{name}_s{step}_mask"] = np.asarray(bd["pseudo_samples_mask"], bool)
            out[f"{name}_s{step}_points_shape"] = np.array(bd["points"].shape)
            out[f"{name}_s{step}_points_sum"] = np.array([np.asarray(bd["points"], np.float64).sum(), np.abs(np.asarray(bd["points"], np.float64)).sum()])
            assert "pseudo_scores" not in bd
            out[f"{name}_s{step}_ema"] = np.array([loader.unknown_score_ema[l] for l in loader.unknown_class_labels], np.float64)
            out[f"{name}_s{step}_prop"] = np.array([loader.sampler.prop_per_unk[l] for l in loader.unknown_class_labels], np.float64)
            out[f"{name}_s{step}_queue"] = np.array([len(loader.sampler.unknown_queue[l]) for l in loader.unknown_class_labels])
            out[f"{name}_s{step}_queue_pts"] = np.array(sorted(o.num_points for q in loader.sampler.unknown_queue.values() for o in q))
            step += 1
    out[f"{name}_missing"] = np.array(sorted(os.path.basename(p) for p in loader.pseudos_missing))
    return out


def processor_inputs():
    """batch of 2 for PseudoProcessor.__call__ / save_predictions"""
    rng = np.random.default_rng(5)
    G, P = 9, 7
    gt = np.zeros((2, G, 10), np.float32)
    ps = np.zeros((2, P, 8), np.float32)
    for b in range(2):
        ng, npb = (7, 5) if b == 0 else (4, 7)
        gt[b, :ng, :7] = syn.random_boxes(rng, ng)
        gt[b, :ng, 7:9] = rng.normal(size=(ng, 2))
        gt[b, :ng, 9] = rng.integers(1, len(KNOWN) + 1, size=ng)
        ps[b, :npb, :7] = syn.random_boxes(rng, npb)
        ps[b, :npb, 7] = rng.choice([2, 4, 7, 10], size=npb)
    ps[0, 1, 3] = 0.0                                  # an empty pseudo box inside the valid range
    mask = np.zeros((2, P), np.float32)
    mask[0, 2] = mask[1, 0] = mask[1, 3] = 1.0
    preds = []
    for b in range(2):
        n = 8
        pb = syn.random_boxes(rng, n, centre_range=15.0)
        pb[0, :7] = ps[b, 2 if b == 0 else 0, :7] + 0.01   # sits on a pasted sample: dropped
        pb9 = np.concatenate([pb, rng.normal(size=(n, 2)).astype(np.float32)], 1)
        preds.append({"pred_boxes": pb9, "pred_scores": rng.uniform(0.1, 0.9, n).astype(np.float32),
                      "pred_labels": rng.integers(1, 11, n).astype(np.int64)})
    aug = {"flip_x": np.array([True, False]), "flip_y": np.array([False, True]), "noise_rot": np.array([0.3, -0.2], np.float32),
           "noise_scale": np.array([1.05, 0.95], np.float32), "noise_translate": rng.normal(0, 0.5, size=(2, 3)).astype(np.float32)}
    return gt, ps, mask, preds, aug


def run_processor(PseudoProcessor, st_folder):
    gt, ps, mask, preds, aug = processor_inputs()
    proc = PseudoProcessor(known_class_names=KNOWN, self_training_folder=st_folder)
    out = {}
    bd = {"gt_boxes": torch.from_numpy(gt.copy()), "pseudo_boxes": torch.from_numpy(ps.copy()), "pseudo_samples_mask": torch.from_numpy(mask.copy())}
    bd = proc(bd)
    out["proc_gt_boxes"] = bd["gt_boxes"].numpy()
    out["proc_stats_keys"] = np.array(sorted(proc.forward_pseudo_stats))
    out["proc_stats_vals"] = np.array([float(proc.forward_pseudo_stats[k]) for k in sorted(proc.forward_pseudo_stats)])
    for epoch in (0, 1):
        bd = {"frame_id": ["fr.a.bin", "fr.b.bin"], "batch_size": 2, "pseudo_boxes": torch.from_numpy(ps.copy()),
              "pseudo_samples_mask": torch.from_numpy(mask.copy())}
        bd.update({k: torch.from_numpy(v.copy()) for k, v in aug.items()})
        pd = [{k: torch.from_numpy(v.copy() + (np.float32(0.05 * epoch) if k == "pred_boxes" else 0)).to(torch.from_numpy(v).dtype) for k, v in p.items()}
              for p in preds]
        proc.save_predictions(bd, pd, epoch=epoch)
        for j, fid in enumerate(bd["frame_id"]):
            d = torch.load(os.path.join(st_folder, fid.replace('.', '_') + ".pth"), map_location="cpu", weights_only=False)
            assert sorted(d) == ["epoch", "pred_boxes", "pred_labels", "pred_scores"] and d["epoch"] == epoch
            for k in ("pred_boxes", "pred_scores", "pred_labels"):
                out[f"proc_e{epoch}_f{j}_{k}"] = d[k].numpy()
        keys = sorted(k for k in proc.forward_pseudo_stats if k.startswith("mean_consistent"))
        out[f"proc_e{epoch}_cons"] = np.array([float(proc.forward_pseudo_stats[k]) for k in keys])
    # a processor over all ten classes is a no-op; relabel on its own
    full = PseudoProcessor(known_class_names=ALL)
    same = {"gt_boxes": torch.from_numpy(gt.copy())}
    assert full(same) is same and np.array_equal(same["gt_boxes"].numpy(), gt)
    out["proc_relabel"] = proc.relabel_gt_boxes(torch.from_numpy(gt.copy())).numpy()
    return out
