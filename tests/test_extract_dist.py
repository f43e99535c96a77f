"""N > 1 path of the sharded pseudo-label extraction (findnpropagate_amd/extract.py) on CPU:
world_size 2 over gloo, a deterministic stand-in head (the Box Seeker itself needs a GPU and is
covered by tests/test_gpu_boxseeker.py).  Checks sharding = pcdet's DistributedSampler, the
fixed-shape all-gather record, the reference on-disk format, resume, and 1-rank == 2-rank output."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from findnpropagate_amd import extract as E


class FakeScenes:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def frame_id(self, i):
        return f"n015-2018-{i:04d}.pcd.bin"

    def __getitem__(self, i):
        # (i + 2) ground-truth rows: the recall counters below depend on the scene index
        return {"frame_id": self.frame_id(i), "index": i, "batch_size": 1, "gt_boxes": torch.ones((1, i + 2, 10))}


class FakeHead(torch.nn.Module):
    """K = index % 5 boxes whose values encode the index; a batch is a list of scene indices."""

    def forward(self, bd):
        out = []
        for i in (bd["index"] if isinstance(bd["index"], list) else [bd["index"]]):
            k = i % 5
            g = torch.Generator().manual_seed(i)
            out.append(dict(pred_boxes=torch.rand((k, 7), generator=g), pred_scores=torch.rand((k,), generator=g),
                            pred_labels=torch.randint(1, 11, (k,), generator=g, dtype=torch.int32)))
        bd["final_box_dicts"] = out
        return bd


def fake_collate(scenes):
    return {"index": [s["index"] for s in scenes], "frame_id": [s["frame_id"] for s in scenes], "batch_size": len(scenes),
            "gt_boxes_list": [s["gt_boxes"][0] for s in scenes]}


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler

    for n, w in ((10, 2), (7, 2), (13, 4), (3, 8), (1, 2)):
        got = [E.shard_indices(n, r, w) for r in range(w)]
        for r in range(w):
            want = list(DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=False))
            assert got[r] == want, (n, w, r)
        assert len({len(g) for g in got}) == 1, "every rank runs the same number of steps"
    assert E.shard_indices(0, 0, 2) == []


def test_record_roundtrip_and_limits():
    pd = dict(pred_boxes=torch.rand(7, 7), pred_scores=torch.rand(7), pred_labels=torch.randint(1, 11, (7,), dtype=torch.int32))
    rec = E.pack_record(pd, 42, torch.device("cpu"))
    assert rec.shape == (E.K_MAX + 1, 9) and rec[0, :2].tolist() == [7.0, 42.0], "count and index ride in the record's header row"
    back, idx = E.unpack_record(rec)
    assert idx == 42 and torch.equal(back["pred_boxes"], pd["pred_boxes"]) and torch.equal(back["pred_labels"], pd["pred_labels"])
    assert back["pred_labels"].dtype == torch.int32
    empty, i3 = E.unpack_record(E.pack_record(dict(pred_boxes=torch.zeros(0, 7), pred_scores=torch.zeros(0), pred_labels=torch.zeros(0, dtype=torch.int32)), 3, torch.device("cpu")))
    assert empty["pred_boxes"].shape == (0, 7) and i3 == 3
    assert E.unpack_record(E.pack_record(empty, -1, torch.device("cpu")))[1] == -1
    with pytest.raises(ValueError):
        E.pack_record(dict(pred_boxes=torch.zeros(E.K_MAX + 1, 7), pred_scores=torch.zeros(E.K_MAX + 1), pred_labels=torch.zeros(E.K_MAX + 1)), 0, torch.device("cpu"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _count_recall(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
    """stand-in for generate_recall_record (which needs the GPU IoU): gt rows and predictions per frame"""
    if not recall_dict:
        recall_dict = {"gt": 0}
        for t in thresh_list:
            recall_dict["rcnn_%s" % str(t)] = 0
    recall_dict["gt"] += int(data_dict["gt_boxes"][batch_index].shape[0])
    for j, t in enumerate(thresh_list):
        recall_dict["rcnn_%s" % str(t)] += int(box_preds.shape[0]) + j
    return recall_dict


def _expected_recall(n):
    gt = sum(i + 2 for i in range(n))
    out = {"gt": gt}
    for j, t in enumerate(E.RECALL_THRESH):
        out["rcnn_%s" % str(t)] = sum(i % 5 + j for i in range(n))
        out["recall_%s" % str(t)] = out["rcnn_%s" % str(t)] / gt
    return out


def _worker(rank, world, port, out_dir, n, write, per_step):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec = {}
        calls = []
        orig = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1]
        E.extract_pseudo_labels(FakeScenes(n), FakeHead(), out_dir, torch.device("cpu"), dist=dist, write=write,
                                recall=rec, recall_fn=_count_recall, scenes_per_step=per_step, collate=fake_collate)
        dist.all_gather_into_tensor = orig
        steps = -(-(-(-n // world)) // per_step)
        assert len(calls) == steps, "ONE collective per step (count and index ride in the record)"
        assert all(tuple(c) == (world * per_step * (E.K_MAX + 1), 9) for c in calls)
        # every rank ends with the totals over ALL scenes, wrap-around duplicates counted once
        want = _expected_recall(n)
        assert {k: rec[k] for k in want} == want, (rank, rec, want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("write,per_step", [("rank0", 1), ("own", 1), ("own", 3), ("rank0", 2)])
def test_two_ranks_equal_one_rank(write, per_step):
    n = 7   # odd: exercises the wrap-around padding of the last step (and a short last step when per_step > 1)
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        rec1 = {}
        assert E.extract_pseudo_labels(FakeScenes(n), FakeHead(), d1, torch.device("cpu"), recall=rec1, recall_fn=_count_recall,
                                       collate=fake_collate) == n
        want = _expected_recall(n)
        assert {k: rec1[k] for k in want} == want
        mp.spawn(_worker, args=(2, _free_port(), d2, n, write, per_step), nprocs=2, join=True)
        files1, files2 = sorted(os.listdir(d1)), sorted(os.listdir(d2))
        assert files1 == files2 == sorted(f"n015-2018-{i:04d}_pcd_bin.pth" for i in range(n))
        for f in files1:
            a, b = torch.load(os.path.join(d1, f)), torch.load(os.path.join(d2, f))
            assert isinstance(a, list) and len(a) == 1 and set(a[0]) == {"pred_boxes", "pred_scores", "pred_labels"}
            for k in a[0]:
                assert torch.equal(a[0][k], b[0][k]), (f, k)
        # resume: a second run writes nothing and leaves the files untouched
        before = {f: os.path.getmtime(os.path.join(d1, f)) for f in files1}
        assert E.extract_pseudo_labels(FakeScenes(n), FakeHead(), d1, torch.device("cpu"), collate=fake_collate) == 0
        assert before == {f: os.path.getmtime(os.path.join(d1, f)) for f in files1}
