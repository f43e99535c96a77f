"""Pins oracle/boxseeker.py (numpy restatement of FrustumProposerOG.get_proposals) against golden
vectors produced by running the reference's own get_proposals on the same synthetic scenes
(tests/golden/make_boxseeker_golden.py; fixtures tests/golden/boxseeker_seed*.npz).  CPU only."""
import os

import numpy as np
import pytest

from findnpropagate_amd import synthetic as syn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ragged(d, key):
    off = d[key + "_off"]
    return [d[key][off[i]:off[i + 1]] for i in range(len(off) - 1)]


@pytest.fixture(scope="module")
def bs():
    from oracle import boxseeker

    return boxseeker


def test_constructor_tables_match_reference(bs):
    d = np.load(os.path.join(GOLD, "boxseeker_seed0.npz"))
    bb, bc = bs.base_proposals()
    np.testing.assert_allclose(bb, d["base_boxes"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(bc, d["base_corners"], rtol=0, atol=2e-6)
    assert bb.shape == (10, 10, 7) and bb[0, 0, 3] == pytest.approx(4.63 * 0.957, rel=1e-6)
    assert bs.linspace_f32(0, 1, 6).tolist() == pytest.approx([0, 0.2, 0.4, 0.6, 0.8, 1.0], abs=1e-7)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 7, 10, 14, 15])
def test_stage_values_match_reference(bs, seed):
    d = np.load(os.path.join(GOLD, f"boxseeker_seed{seed}.npz"))
    sc = syn.make_seeker_scene(seed)
    # 2D batched NMS per camera, in image_order
    ci = 0
    for c in bs.IMAGE_ORDER:
        m = sc["dets"][4] == c
        if not m.any():          # the reference skips batched_nms for a camera without detections (:586)
            continue
        boxes, scores, labels = _ragged(d, "nms2d_in_boxes")[ci], _ragged(d, "nms2d_in_scores")[ci][:, 0], _ragged(d, "nms2d_in_labels")[ci][:, 0]
        assert np.array_equal(boxes, sc["dets"][0][m])
        keep = bs.batched_nms_2d(boxes, scores, labels, 0.4)
        assert keep.tolist() == _ragged(d, "nms2d_keep")[ci][:, 0].tolist()
        ci += 1
    assert ci == len(d["nms2d_keep_off"]) - 1
    # back-projection (get_geometry_at_image_coords) on every recorded call
    gin, gout, gcam = _ragged(d, "geom_in"), _ragged(d, "geom_out"), d["geom_cam"]
    for a, b, c in zip(gin, gout, gcam):
        got = bs.geometry_at_image_coords(a, sc["camera2lidar"][0, c], sc["camera_intrinsics"][0, c], sc["lidar_aug_matrix"][0],
                                          sc["img_aug_matrix"][0, c] if "img_aug_matrix" in sc else None)
        np.testing.assert_allclose(got, b, rtol=1e-5, atol=2e-4)
    # corner projection + 2D IoU (calc_iou) on every recorded call
    cams_of_iou = [c for c, n in zip(d["proj_cam"], d["proj_n"]) if n <= 600]   # (the corner projections of calc_iou)
    for corners, box, want, c in zip(_ragged(d, "iou_corners"), _ragged(d, "iou_box"), _ragged(d, "iou_out"), cams_of_iou):
        got, _ = bs.calc_iou(corners.reshape(-1, 8, 3), box[:, 0], sc["lidar_aug_matrix"][0], sc["lidar2image"][0, c],
                             sc["img_aug_matrix"][0, c] if "img_aug_matrix" in sc else None)
        np.testing.assert_allclose(got, want[:, 0], rtol=1e-4, atol=1e-5)


SEEDS = list(range(18))    # 0-2 plain, 3-15 the edge cases of synthetic.SEEKER_VARIANTS, 16-17 other PARAMS (MULT, ego_w)


@pytest.mark.parametrize("seed", SEEDS)
def test_get_proposals_matches_reference(bs, seed):
    from seeker_parity import check_choices

    d = np.load(os.path.join(GOLD, f"boxseeker_seed{seed}.npz"))
    sc = syn.make_seeker_scene(seed)
    assert ",".join(sc["variant"]) == str(d["variant"])
    trace = []
    pv = syn.SEEKER_PARAM_VARIANTS.get(seed, ({}, {}))
    boxes, labels, scores = bs.get_proposals(sc, params={**pv[0], **pv[1]}, trace=trace)
    assert boxes.shape == d["out_boxes"].shape
    assert labels.tolist() == d["out_labels"].tolist()
    np.testing.assert_allclose(scores, d["out_scores"], rtol=0, atol=1e-7)
    if boxes.shape[0] == 0:      # no_dets / low_scores: the early return of :694-700
        assert "pib_count" not in d.files
        return
    # candidate point counts (the reference's per-candidate points_in_boxes_gpu calls), in call order
    mine = np.concatenate([t["counts"] for t in trace if "counts" in t]).astype(np.int64)
    want = d["pib_count"]
    assert mine.shape == want.shape
    assert (mine != want).mean() < 0.01 and np.abs(mine - want).max() <= 2      # face-grazing points only
    cand = np.concatenate([t["cand_boxes"][t["idx_final"]] for t in trace if "idx_final" in t])
    np.testing.assert_allclose(cand, d["pib_box"], rtol=0, atol=1e-4)
    dst_tol = 2e-3 * float(pv[0].get("dst_w", 0.0))     # torch.cdist's matmul formulation in the reference (seeker_parity.py)
    k0 = 0
    for t, ws in zip([t for t in trace if "scores" in t], _ragged(d, "nms3d_scores")):
        ref = want[k0:k0 + len(ws)]
        k0 += len(ws)
        dcount = np.abs(t["counts"].astype(np.int64) - ref).max()
        np.testing.assert_allclose(t["scores"], ws[:, 0], rtol=0, atol=1e-4 + dst_tol + 2.0 * dcount / max(ref.max(), 1))
    chosen = [(t["idx_final"], t["best"], t["counts"]) for t in trace if "scores" in t]
    n_unique, n_tied_ref, n_loose = check_choices(d, boxes, chosen, extra_tol=dst_tol)
    if "lone_point" in sc["variant"]:
        assert any(t["n_points"] == 1 and "scores" in t for t in trace), "the single-return frustum reaches the scoring stage"
    # the bound comes from the fixture: every frustum whose reference scores are NOT tied (beyond 1e-4) was compared with the
    # reference's output box, except the few decided only through a face-grazing point count (<= 2 points, < 1 % of candidates)
    assert n_unique == boxes.shape[0] - n_tied_ref - n_loose and n_loose <= max(1, boxes.shape[0] // 10)


OPTION_SEEDS = [18, 19, 20, 21, 22, 23, 24, 25, 27]   # synthetic.SEEKER_PARAM_VARIANTS: the options no shipped config sets


def option_scene(seed):
    """scene, params and rand_center draws of an option seed, as tests/golden/make_boxseeker_golden.py ran it"""
    d = np.load(os.path.join(GOLD, f"boxseeker_seed{seed}.npz"))
    sc = syn.make_seeker_scene(seed)
    pv = syn.SEEKER_PARAM_VARIANTS[seed]
    prm = {**pv[0], **pv[1]}
    if prm.get("BOX_FORMAT", "xyxy") != "xyxy":      # the detector hands out [x, y, w, h]
        b = sc["dets"][0].copy()
        b[:, 2:] -= b[:, :2]
        sc["dets"] = (b,) + tuple(sc["dets"][1:])
    noise = [n.reshape(-1, 3) for n in _ragged(d, "randn")] if "randn" in d.files else None
    return d, sc, prm, noise


def boxes_equal_mod_half_turn(a, b, atol=1e-4):
    """rows equal in all 7 components, the yaw modulo pi (a footprint turned by half a turn is the same candidate to every
    score term: the reference's own unstable sort decides such ties)"""
    dlt = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    dlt[:, 6] = np.minimum(dlt[:, 6], np.abs(dlt[:, 6] - np.pi))
    return (dlt <= atol).all(1)


@pytest.mark.parametrize("seed", OPTION_SEEDS)
def test_option_variants_match_reference(bs, seed):
    """topk 3 through the 3D NMS (threshold 0.3), search_depth, MULTICAM_IOU, occl_w, OCCL_MULT, rand_center (with the
    reference's recorded draws), xywh detections, num_mags 0 and a combination: every per-candidate second-stage score the
    reference handed to nms_normal_gpu, the candidates it counted points in, and the boxes it returned."""
    d, sc, prm, noise = option_scene(seed)
    trace = []
    boxes, labels, scores = bs.get_proposals(sc, params=prm, trace=trace, noise=noise)
    assert boxes.shape == d["out_boxes"].shape and labels.tolist() == d["out_labels"].tolist()
    np.testing.assert_allclose(scores, d["out_scores"], rtol=0, atol=1e-7)
    scored = [t for t in trace if "scores" in t]
    ws, sel = _ragged(d, "nms3d_scores"), _ragged(d, "nms3d_selected")
    assert len(scored) == len(ws)
    occl = bool(prm.get("occl_w", 0) or prm.get("OCCL_MULT", False))
    # the reference's points_in_boxes_gpu calls, in order: per frustum one per scored candidate for the density term, then the
    # same candidates again inside calc_occl_scores for each occlusion term that is on
    calls_per_cand = 1 + int(prm.get("occl_w", 0) > 0) + int(bool(prm.get("OCCL_MULT", False)))
    pos = 0
    for t in scored:
        cand = t["cand_boxes"][t["idx_final"]]
        for _ in range(calls_per_cand):
            np.testing.assert_allclose(cand, d["pib_box"][pos: pos + len(cand)], rtol=0, atol=1e-4)
            pos += len(cand)
    assert pos == d["pib_box"].shape[0]
    for t, w in zip(scored, ws):
        assert len(t["scores"]) == len(w)
        # a face-grazing point (<= 2 per candidate, < 1 % of them) moves a density term by 2 / max count and an occlusion
        # product by that share of itself
        np.testing.assert_allclose(t["scores"], w[:, 0], rtol=2e-2 if occl else 0, atol=1e-4 + 2.0 / max(float(t["counts"].max()), 1.0))
    same = boxes_equal_mod_half_turn(boxes, d["out_boxes"])
    assert same.mean() >= 0.9, f"{(~same).sum()} of {len(same)} boxes differ from the reference's beyond a half-turn tie"
    if prm.get("topk", 1) > 1:
        per = [len(t["selected"]) for t in scored]
        assert max(per) > 1 and sum(per) == boxes.shape[0] and [min(len(s_), prm["topk"]) for s_ in sel] == per


def test_aln_w_is_dead_code_in_the_reference():
    """PARAMS aln_w > 0: the reference's own get_proposals raises (a (1, N) mask on an (N, 3) tensor, :988) — recorded by the
    fixture generator; the mirror refuses the option with that explanation."""
    d = np.load(os.path.join(GOLD, "boxseeker_seed26.npz"))
    assert str(d["raised"]) == "IndexError" and "mask" in str(d["message"])
    from findnpropagate_amd.dense_heads import FrustumProposerOG
    with pytest.raises(NotImplementedError, match="IndexError"):
        FrustumProposerOG(model_cfg={"PARAMS": {"aln_w": 0.3, "nms_3d": 0.0}, "PREDS_PATH": "PreprocessedGLIP"}, image_detector=lambda bd: None)


def test_quantile_matches_torch(bs):
    import torch

    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 17, 1000):
        x = rng.uniform(2, 50, n).astype(np.float32)
        for q in (0.0, 0.25, 0.5, 1.0):
            assert bs.quantile(x, q) == pytest.approx(float(torch.quantile(torch.from_numpy(x), q)), rel=1e-6)
