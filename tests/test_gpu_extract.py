"""BASELINE.json configs[3] on the GPU: findnpropagate_amd.extract.extract_pseudo_labels with the REAL FrustumProposerOG
on cuda:0 — the sync-free pipeline (device-side record packing, one collective per step on a side stream, files written
one step later), through a world-size-1 process group of backend "nccl" (= RCCL), at 1 and 3 scenes per launch —
against (a) the plain synchronous single-process path, (b) oracle/boxseeker.py scene by scene, (c) the reference's
on-disk format and its recall bookkeeping."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch

from findnpropagate_amd import extract as E, synthetic as syn

pytestmark = pytest.mark.gpu
PARAMS = {'lq': 0.0, 'uq': 0.25, 'cq': 1.0, 'iou_w': 1.0, 'nms_normal': 1.0, 'dst_w': 0.0, 'dns_w': 1.0,
          'min_cam_iou': 0.3, 'score_thr': 0.45, 'nms_2d': 0.4, 'nms_3d': 0.0, 'clamp_bottom': 1, 'num_sizes': 1}
VARIANTS = [(), ("aug",), ("empty_cam", "lone_point"), ("no_dets",), ("aug", "flip")]
N_FRAMES, DISTINCT = 7, 5


def _head():
    from findnpropagate_amd.dense_heads import FrustumProposerOG

    return FrustumProposerOG(model_cfg={"PARAMS": dict(PARAMS), "PREDS_PATH": "PreprocessedGLIP", "BOX_FORMAT": "xyxy"},
                             image_detector=lambda bd: bd["dets"]).eval()


@pytest.fixture(scope="module")
def nccl_world1(cuda):
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=cuda)
    yield dist
    dist.destroy_process_group()


def _load(d):
    out = {}
    for f in sorted(os.listdir(d)):
        v = torch.load(os.path.join(d, f))
        assert isinstance(v, list) and len(v) == 1 and set(v[0]) == {"pred_boxes", "pred_scores", "pred_labels"}   # extract_pseudo_labels.py:137
        assert v[0]["pred_boxes"].dtype == torch.float32 and v[0]["pred_labels"].dtype == torch.int32
        out[f] = v[0]
    return out


def test_extraction_nccl_pipeline_equals_sync_path_and_oracle(cuda, nccl_world1):
    from oracle import boxseeker as OB
    from seeker_parity import BOX_ATOL

    data = syn.SeekerScenes(N_FRAMES, DISTINCT, cuda, seed0=30, variants=VARIANTS)
    head = _head()
    with tempfile.TemporaryDirectory() as d_sync, tempfile.TemporaryDirectory() as d_p1, tempfile.TemporaryDirectory() as d_p3:
        r_sync, r_p1, r_p3 = {}, {}, {}
        assert E.extract_pseudo_labels(data, head, d_sync, cuda, recall=r_sync, pipeline=False) == N_FRAMES
        assert E.extract_pseudo_labels(data, head, d_p1, cuda, dist=nccl_world1, write="rank0", recall=r_p1) == N_FRAMES
        assert E.extract_pseudo_labels(data, head, d_p3, cuda, dist=nccl_world1, write="own", recall=r_p3, scenes_per_step=3) == N_FRAMES
        a, b, c = _load(d_sync), _load(d_p1), _load(d_p3)
        assert sorted(a) == sorted(b) == sorted(c) == sorted(f"synthetic-{i:06d}_pcd_bin.pth" for i in range(N_FRAMES))
        for f in a:
            for k in a[f]:
                assert torch.equal(a[f][k], b[f][k]) and torch.equal(a[f][k], c[f][k]), (f, k)   # same kernel, same bits
        assert r_sync == r_p1 == r_p3 and r_sync["gt"] == sum(sc["gt_boxes"].shape[0] for sc in (data.raw[i % DISTINCT] for i in range(N_FRAMES)))
        assert 0 < r_sync["rcnn_0.3"] <= r_sync["gt"] and r_sync["rcnn_0.7"] <= r_sync["rcnn_0.5"] <= r_sync["rcnn_0.3"]
        # scene by scene against the numpy oracle: labels / scores exact; each box equals an oracle candidate whose oracle
        # score is within float noise of the oracle's best (exact ties are the reference's own unstable sort)
        for i in range(N_FRAMES):
            sc = data.raw[i % DISTINCT]
            trace = []
            ob, ol, os_ = OB.get_proposals(sc, trace=trace)
            got = a[f"synthetic-{i:06d}_pcd_bin.pth"]
            assert got["pred_labels"].tolist() == ol.tolist()
            np.testing.assert_allclose(got["pred_scores"].numpy(), os_, rtol=0, atol=1e-7)
            scored = [t for t in trace if "scores" in t]
            for k, t in enumerate(scored):
                cand = t["cand_boxes"][t["idx_final"]]
                match = np.abs(cand - got["pred_boxes"][k].numpy()[None]).max(1) <= BOX_ATOL
                assert match.any(), (i, k)
                tol = 1e-4 + 2.0 * 2 / max(float(t["counts"].max()), 1.0)     # <= 2 face-grazing points (test_gpu_boxseeker pins that)
                assert t["scores"][match].max() >= t["scores"].max() - tol, (i, k)
        # resume: nothing is rewritten
        before = {f: os.path.getmtime(os.path.join(d_p1, f)) for f in b}
        assert E.extract_pseudo_labels(data, head, d_p1, cuda, dist=nccl_world1) == 0
        assert before == {f: os.path.getmtime(os.path.join(d_p1, f)) for f in b}


def test_pipeline_issues_one_collective_per_step_and_no_sync(cuda, nccl_world1):
    """8 frames, 2 per step -> 4 all_gather_into_tensor calls of the packed record; the loop itself never calls
    torch.cuda.synchronize / .item() on the compute stream (checked through the head: launch() only)."""
    data = syn.SeekerScenes(8, 2, cuda, seed0=40)
    head = _head()
    calls = []
    orig = nccl_world1.all_gather_into_tensor
    nccl_world1.all_gather_into_tensor = lambda *a, **k: (calls.append(tuple(a[0].shape)), orig(*a, **k))[1]
    fwd = []
    head.forward = lambda *a, **k: fwd.append(1)          # the pipeline must not take the synchronous entry
    try:
        with tempfile.TemporaryDirectory() as d:
            assert E.extract_pseudo_labels(data, head, d, cuda, dist=nccl_world1, scenes_per_step=2) == 8
    finally:
        nccl_world1.all_gather_into_tensor = orig
    assert not fwd
    # world size 1 short-circuits the collective (nothing to exchange): the record shape is checked by the gloo test
    assert calls == [] or calls == [(2 * (E.K_MAX + 1), 9)] * 4


def test_rccl_collective_runs_on_one_gpu_and_is_consumed_a_step_late(cuda, nccl_world1):
    """force_collective: the nccl (= RCCL) all_gather_into_tensor of the packed record is really issued every step in the
    world-size-1 group — on the side stream, shape (S (K_MAX + 1), 9) in and out — its result is consumed one step later
    (step k's records reach the writer after step k + 1 has been launched), and the files equal those of the path that
    skips the collective."""
    data = syn.SeekerScenes(8, 4, cuda, seed0=60, variants=[(), ("aug",), ("no_dets",), ("lone_point",)])
    head = _head()
    calls = []
    orig = nccl_world1.all_gather_into_tensor

    def spy(out, inp, **kw):
        calls.append((tuple(out.shape), tuple(inp.shape), torch.cuda.current_stream(cuda) != torch.cuda.default_stream(cuda), kw.get("async_op", False)))
        return orig(out, inp, **kw)
    nccl_world1.all_gather_into_tensor = spy
    trace = []
    try:
        with tempfile.TemporaryDirectory() as d_f, tempfile.TemporaryDirectory() as d_s:
            assert E.extract_pseudo_labels(data, head, d_f, cuda, dist=nccl_world1, scenes_per_step=2, force_collective=True, trace=trace) == 8
            n_calls = len(calls)
            assert E.extract_pseudo_labels(data, head, d_s, cuda, dist=nccl_world1, scenes_per_step=2) == 8
            assert len(calls) == n_calls, "without the switch a one-rank group skips the collective"
            a, b = _load(d_f), _load(d_s)
            assert sorted(a) == sorted(b) and len(a) == 8
            for f in a:
                for k in a[f]:
                    assert torch.equal(a[f][k], b[f][k]), (f, k)
    finally:
        nccl_world1.all_gather_into_tensor = orig
    rows = 2 * (E.K_MAX + 1)
    assert calls[:4] == [((rows, 9), (rows, 9), True, True)] * 4 and n_calls == 4
    # order: launch k, collective k, then (for k >= 1) consume k - 1; the last step is consumed after the loop
    want = []
    for k in range(4):
        want += [("launch", k), ("collective", k)] + ([("consume", k - 1)] if k else [])
    want.append(("consume", 3))
    assert trace == want, trace


def test_collate_scenes_batches_like_single_scenes(cuda):
    data = syn.SeekerScenes(3, 3, cuda, seed0=50, variants=[(), ("aug",), ("lone_point",)])
    head = _head()
    batch = E.collate_scenes([data[i] for i in range(3)])
    assert batch["batch_size"] == 3 and batch["points_per_scene"] == [data[i]["points"].shape[0] for i in range(3)]
    with torch.no_grad():
        out = head.forward(dict(batch))["final_box_dicts"]
        for i in range(3):
            one = head.forward(dict(data[i]))["final_box_dicts"][0]
            for k in one:
                assert torch.equal(one[k].cpu(), out[i][k].cpu()), (i, k)
