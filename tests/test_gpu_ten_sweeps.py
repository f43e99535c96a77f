"""The workload transfusion_lidar.yaml actually feeds the backbone: 10 aggregated sweeps (tools/cfgs/dataset_configs/
nuscenes_dataset.yaml:5 MAX_SWEEPS 10; nuscenes_models/transfusion_lidar.yaml:54-59 caps of 120 k / 160 k voxels) — ~300 k
points and ~150 k voxels per scene, nine times the single-sweep scenes of the other full-grid tests — on the full
41 x 1440 x 1440 grid: voxel rows bit-exact against the oracle's sequential voxeliser (with the `max_voxels` rule firing in
the middle of a scene), site sets of every stage, the f32 engine equal to the oracle bit for bit, bf16 features against the
bf16-emulating oracle.  The engine's heuristics (class-sorted sweep of stage 4, tile windows of stages 2-3) were sized on
single-sweep geometry: the statistics they rest on are printed by bench.py's secondary.ten_sweep; here only results count."""
import numpy as np
import pytest
import torch

from findnpropagate_amd import lib as _l
from findnpropagate_amd import sparse as S
from findnpropagate_amd import synthetic as syn
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GRID = [41, 1440, 1440]


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _key(i, s):
    return ((i[:, 0].astype(np.int64) * s[0] + i[:, 1]) * s[1] + i[:, 2]) * s[2] + i[:, 3]


def _net(cuda, dtype):
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    return syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": dtype}, 5, grid), seed=0).to(cuda).eval()


def _oracle_voxels(seeds, max_voxels):
    coords, feats, kept = [], [], []
    for b, s in enumerate(seeds):
        v, c, n = O.voxelize(syn.make_sweeps_scene(s), syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, syn.MAX_POINTS_PER_VOXEL, max_voxels)
        coords.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
        feats.append(O.mean_vfe(v, n))
        kept.append(n)
    return np.concatenate(coords), np.concatenate(feats), np.concatenate(kept)


def _check_sites(res, coords, batch):
    shape, idx = GRID, coords
    for name, (k, s, p) in (("x_conv2", (3, 2, 1)), ("x_conv3", (3, 2, 1)), ("x_conv4", (3, 2, (0, 1, 1))), ("out", ((3, 1, 1), (2, 1, 1), 0))):
        idx, shape, _, _, _ = O.rulebook_strided(idx, shape, k, s, p)
        got = res[name]
        assert got.spatial_shape == shape and got.indices.shape[0] == idx.shape[0], name
        assert np.array_equal(np.sort(_key(got.indices.cpu().numpy(), shape)), np.sort(_key(idx, shape))), name


def _sorted_feats(t):
    i = t.indices.cpu().numpy() if hasattr(t.indices, "cpu") else t.indices
    f = t.features.float().cpu().numpy() if hasattr(t.features, "cpu") else t.features
    o = np.argsort(_key(i, t.spatial_shape))
    return f[o]


def test_ten_sweep_scene_with_the_voxel_cap_firing_f32_is_the_oracle(cuda):
    """One 10-sweep scene with max_voxels = 60,000 (the scene has ~150 k occupied cells: the `continue` of the sequential
    voxeliser fires from point ~60 k on, later points still join voxels that exist): coordinates, first-come order, point
    counts and mean features bit-exact; the f32 engine's five outputs array_equal to the oracle's."""
    seeds, cap = (31,), 60000
    pts, off = syn.make_sweeps_batch(seeds)
    assert pts.shape[0] > 250000
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, cap)
    net = _net(cuda, "fp32")
    aborts0 = _l.load().fnp_spconv_tiled_aborts()
    with torch.no_grad():
        res = net.forward_points(torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda), 1, cfg)
    coords, feats, kept = _oracle_voxels(seeds, cap)
    assert coords.shape[0] == cap, "the cap must fire for this test to mean anything"
    assert np.array_equal(res["voxel_coords"].cpu().numpy(), coords)
    assert np.array_equal(res["voxel_num_points"].cpu().numpy(), kept)
    assert np.array_equal(res["voxel_features"].cpu().numpy(), feats)
    _check_sites(res, coords, 1)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want = O.backbone_forward(sd, feats, coords, 1, GRID)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        assert np.array_equal(_sorted_feats(res[name]), _sorted_feats(want[name])), name
    assert _l.load().fnp_spconv_tiled_aborts() == aborts0


def test_two_ten_sweep_scenes_bf16_engine(cuda):
    """Two 10-sweep scenes (the eval cap of 160,000 voxels does not fire on them: ~150 k each) through the default bf16 engine —
    every heuristic of the fused path at 10-sweep density (300 k rows in stage 1, 450 k in stage 2): voxel rows bit-exact, site
    sets equal, features against the bf16-emulating oracle with the bounds of the single-sweep full-grid test, and the rerun
    bit-identical (grids left clean)."""
    seeds = (32, 33)
    pts, off = syn.make_sweeps_batch(seeds)
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
    net = _net(cuda, "bf16")
    d_pts, d_off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
    with torch.no_grad():
        res = net.forward_points(d_pts, d_off, 2, cfg)
        res2 = net.forward_points(d_pts, d_off, 2, cfg)
    coords, feats, kept = _oracle_voxels(seeds, syn.MAX_VOXELS_TEST)
    assert coords.shape[0] > 250000
    assert np.array_equal(res["voxel_coords"].cpu().numpy(), coords)
    assert np.array_equal(res["voxel_features"].cpu().numpy(), feats)
    _check_sites(res, coords, 2)
    assert torch.equal(res["out"].features, res2["out"].features)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want = O.backbone_forward(sd, feats, coords, 2, GRID, bf16=True)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        g, w = _sorted_feats(res[name]), _sorted_feats(want[name])
        bad = np.abs(g - w) > 3e-2 + 3e-2 * np.abs(w)
        assert bad.mean() <= 0.01, (name, bad.mean())
        ulp = 2.0 ** -8 * max(1.0, np.abs(w).max())      # one bf16 ulp at the tensor's scale
        assert np.abs(g - w).max() <= 4 * ulp, (name, np.abs(g - w).max(), ulp)


def test_tile_gate_switches_on_the_measured_escape_share(cuda):
    """The tile-rulebook kernels of stages 2-3 are taken by a measured statistic, not by a stage number: the rulebook kernel
    counts the 32-row groups with an escape entry, the engine reads the count with its per-forward counts and runs a stage on the
    gather kernels while the share exceeds TILE_ESC_MAX.  Single-sweep scenes stay on tiles (share ~1e-5); a 10-sweep scene
    switches at least stage 3 off for the next forwards; results are the same bits either way."""
    if S.TILE_MODE is not None:
        pytest.skip("FNP_TILE forces the kernel")
    net = _net(cuda, "bf16")
    eng = net.engine()
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
    p1, o1 = syn.make_batch((5, 6))
    with torch.no_grad():
        net.forward_points(torch.from_numpy(p1).to(cuda), torch.from_numpy(o1).to(cuda), 2, cfg)
    assert eng._heur_key() == [] and all(v < eng.TILE_ESC_MAX for v in eng.tile_escape_share.values()) and len(eng.tile_escape_share) == 2
    p10, o10 = syn.make_sweeps_batch((34,))
    d_p, d_o = torch.from_numpy(p10).to(cuda), torch.from_numpy(o10).to(cuda)
    with torch.no_grad():
        a = net.forward_points(d_p, d_o, 1, cfg)        # tiles (and the measurement)
        assert 1 in eng._heur_key(), eng.tile_escape_share
        b = net.forward_points(d_p, d_o, 1, cfg)        # stage 3 (at least) on the gather kernels
    for name in ("x_conv2", "x_conv3", "x_conv4", "out"):
        assert torch.equal(a[name].features, b[name].features), name
    assert eng.tile_off[1] == eng.TILE_REPROBE - 1


def test_tile_gate_reopens_for_graph_replays_and_backs_off_on_dense_streams(cuda):
    """ADVICE r04 (medium).  A captured-graph user meets one dense frame: the gated stage's period is counted down by REPLAYED
    frames too, the stage returns to tiles (a recapture) when the period ends and — the frames being single sweeps again — stays
    there; on a stream that stays dense every failed re-probe doubles the period.  Same bits throughout."""
    if S.TILE_MODE is not None:
        pytest.skip("FNP_TILE forces the kernel")
    net = _net(cuda, "bf16")
    eng = net.engine()
    eng.TILE_REPROBE = 3                      # (instance override: the class default is 64 frames)
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
    p1, o1 = syn.make_batch((5,))
    p10, o10 = syn.make_sweeps_batch((34,))
    s_p, s_o = torch.from_numpy(p1).to(cuda), torch.from_numpy(o1).to(cuda)
    d_p, d_o = torch.from_numpy(p10).to(cuda), torch.from_numpy(o10).to(cuda)
    cap = 393216
    with torch.no_grad():
        want = {k: v.features.clone() for k, v in net.forward_points(s_p, s_o, 1, cfg).items() if k.startswith("x_conv") or k == "out"}
        run = lambda p, o: net.forward_points_graphed(p, o, 1, cfg, capacity=cap)
        run(s_p, s_o)
        assert eng._heur_key() == []
        run(d_p, d_o)                                        # the dense frame: measured on tiles, stage 3 gated
        assert 1 in eng._heur_key() and eng.tile_off[1] == 3 and eng.tile_period[1] == 3
        g_before = eng._graphs[(1, cap, 5, str(cuda), False)]
        for left in (2, 1):
            got = run(s_p, s_o)                              # replays on the gather kernels count the period down
            assert eng.tile_off[1] == left
        assert eng._graphs[(1, cap, 5, str(cuda), False)] is not g_before        # (one recapture, when the gate closed)
        got = run(s_p, s_o)                                  # period over: tiles again, measured, fine
        assert eng._heur_key() == [] and 1 not in eng.tile_period and eng.tile_escape_share[1] < eng.TILE_ESC_MAX
        for k, w in want.items():
            assert torch.equal(got[k].features, w), k
        # a stream that stays dense: 3 frames gated, re-probe fails -> 6, fails again -> 12
        periods = []
        for _ in range(3 + 1 + 6 + 1):
            run(d_p, d_o)
            periods.append(eng.tile_period.get(1))
        assert periods[0] == 3 and periods[2] == 3 and periods[3] == 6 and periods[8] == 6 and periods[9] == 12, periods


def test_ten_sweep_scene_bf16x3_engine_with_and_without_tiles(cuda):
    """FNP_DTYPE: bf16x3 at 10-sweep density: the first forward runs stages 2-3 on the tile rulebook with a lean int32 table
    (2 % / 11 % of the 32-row groups fetch through it: the escape path of all three launches of a layer, the split epilogue
    included), the second — the tile gate has switched — on the gather kernels with the full table.  Both against the f32
    engine (= the oracle bit for bit): same sites, every output within 3e-5 of its feature scale; and the two agree to the same
    bound (the cross terms are rounded to bf16 by either path)."""
    if S.TILE_MODE is not None:
        pytest.skip("FNP_TILE forces the kernel")
    pts, off = syn.make_sweeps_batch((35,))
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
    d_p, d_o = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
    x3, ref = _net(cuda, "bf16x3"), _net(cuda, "fp32")
    aborts0 = _l.load().fnp_spconv_tiled_aborts()
    with torch.no_grad():
        a = x3.forward_points(d_p, d_o, 1, cfg)
        assert 1 in x3.engine()._heur_key(), x3.engine().tile_escape_share
        b = x3.forward_points(d_p, d_o, 1, cfg)
        want = ref.forward_points(d_p, d_o, 1, cfg)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        assert torch.equal(a[name].indices, want[name].indices) and torch.equal(b[name].indices, want[name].indices), name
        w = want[name].features
        scale = max(1.0, float(w.abs().max()))
        for got in (a, b):
            assert got[name].features.dtype == torch.float32
            err = float((got[name].features - w).abs().max())
            assert err <= 3e-5 * scale, (name, err, scale)
    assert _l.load().fnp_spconv_tiled_aborts() == aborts0


@pytest.mark.parametrize("mode,cin,cout,s,p", [("subm", 16, 16, 1, 1), ("strided", 16, 32, 2, 1), ("subm", 64, 64, 1, 1),
                                               ("subm", 5, 16, 1, 1)])   # (5 -> 16: conv_input, zero-padded to the 16 -> 16 MFMA kernels)
def test_backward_on_a_ten_sweep_scene(cuda, mode, cin, cout, s, p):
    """a26 at the density transfusion_lidar.yaml trains on (MAX_SWEEPS 10; MAX_NUMBER_OF_VOXELS 120 k in training,
    transfusion_lidar.yaml:54-59 — the cap fires on this scene): data and weight gradients of a SubM and a strided layer on the
    voxel coordinates of ONE 10-sweep scene at the full 41 x 1440 x 1440 range (the 64 -> 64 layer on that scene's stage-3 sites),
    fp16 — the reference's AMP dtype — against the oracle's conv_backward."""
    from findnpropagate_amd import spconv
    rng = np.random.default_rng(11)
    pts, off = syn.make_sweeps_batch((36,))
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, 120000)
    vox = S.voxelize(torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda), 1, cfg)
    n = int(vox["n"].item())
    assert n == 120000, "the training cap must fire"
    idx, shape = vox["coords"][:n].cpu().numpy(), list(GRID)
    if cin == 64:                                  # the stage-3 sites of the scene (two stride-2 layers down)
        for _ in range(2):
            idx, shape, _, _, _ = O.rulebook_strided(idx, shape, 3, 2, 1)
        shape = [int(v) for v in shape]
    n = idx.shape[0]
    td = torch.float16
    feats = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(td).float().numpy()
    kk, ss, pp = [3] * 3, [s] * 3, [p] * 3
    conv = (spconv.SubMConv3d(cin, cout, kk, padding=[1, 1, 1], bias=False, indice_key="a") if mode == "subm"
            else spconv.SparseConv3d(cin, cout, kk, stride=ss, padding=pp, bias=False)).to(cuda)
    w = conv.weight.detach().to(td).float().cpu().numpy()
    x = torch.from_numpy(feats).to(cuda).to(td).requires_grad_(True)
    out = conv(spconv.SparseConvTensor(x, torch.from_numpy(idx).to(cuda), shape, 1))
    oi = out.indices.cpu().numpy()
    dy = torch.from_numpy(rng.standard_normal((oi.shape[0], cout)).astype(np.float32)).to(td).float().numpy()
    (out.features.float() * torch.from_numpy(dy).to(cuda)).sum().backward()
    if mode == "subm":
        pin, pout, pnum = O.rulebook_subm(idx, shape, kk)
        order = np.arange(n)
    else:
        o_idx, o_shape, pin, pout, pnum = O.rulebook_strided(idx, shape, kk, ss, pp)
        assert np.array_equal(np.sort(_key(oi, o_shape)), np.sort(_key(o_idx, o_shape)))       # site set at the full range
        lut = {int(v): i for i, v in enumerate(_key(oi, o_shape))}
        order = np.array([lut[int(v)] for v in _key(o_idx, o_shape)])
    dx_w, dw_w = O.conv_backward(feats, w, pin, pout, pnum, dy[order])
    for got, want in ((x.grad.float().cpu().numpy(), dx_w), (conv.weight.grad.float().cpu().numpy(), dw_w)):
        scale = max(float(np.abs(want).max()), 1e-6)
        assert np.abs(got - want).max() <= 4e-3 * scale, (mode, cin, np.abs(got - want).max(), scale)


def test_amp_training_step_at_the_shipped_configuration(cuda, monkeypatch):
    """BATCH_SIZE_PER_GPU 4 x MAX_SWEEPS 10 under AMP (transfusion_lidar.yaml:147, nuscenes_dataset.yaml:5, train_utils.py:135-176) is
    what tools/bench_train.py --sweeps 10 --batch 4 --amp times; here ONE such scene through the same step: the training voxel
    cap fires, the step is not skipped (finite unscaled gradients), every parameter moves, site sets equal the oracle's.
    Every BatchNorm of the step runs on the library's fused kernels in the autocast dtype (torch's batch_norm is made to raise:
    until round 5 the activations were cast fp16 -> f32 -> bf16 -> fp16 around a torch batch_norm per layer)."""
    def _no_torch_bn(*a, **k):
        raise AssertionError("torch batch_norm ran inside the AMP step")
    monkeypatch.setattr(torch.nn.functional, "batch_norm", _no_torch_bn)
    net = _net(cuda, "bf16").train()
    pts, off = syn.make_sweeps_batch((37,))
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, 120000)
    vox = S.voxelize(torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda), 1, cfg)
    n = int(vox["n"].item())
    assert n == 120000
    bd = {"voxel_features": vox["mean"][:n], "voxel_coords": vox["coords"][:n].float(), "batch_size": 1}
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        out = net(bd)
        loss = (out["encoded_spconv_tensor"].features.float() ** 2).mean()
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    norm = torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
    s0 = scaler.get_scale()
    scaler.step(opt)
    scaler.update()
    assert torch.isfinite(loss) and torch.isfinite(norm) and scaler.get_scale() == s0, "the step was skipped"
    moved = [k for k, v in net.named_parameters() if not torch.equal(v.detach(), before[k])]
    assert len(moved) == len(before), sorted(set(before) - set(moved))[:5]
    res = {"out": out["encoded_spconv_tensor"], **out["multi_scale_3d_features"]}
    _check_sites(res, vox["coords"][:n].cpu().numpy(), 1)


def test_amp_module_forward_equals_the_fp16_module_forward(cuda):
    """Under autocast(fp16) the module path keeps its activations in fp16 from layer to layer, as spconv + nn.BatchNorm1d do in
    the reference's AMP step: its outputs are those of the same network built with FNP_DTYPE fp16 and run without autocast
    (same kernels, same order), up to conv_input, which autocast runs in fp16 and the fp16 network in f32."""
    pts, off = syn.make_batch((3, 4))
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, 160000)
    vox = S.voxelize(torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda), 2, cfg)
    n = int(vox["n"].item())
    bd = lambda: {"voxel_features": vox["mean"][:n], "voxel_coords": vox["coords"][:n].float(), "batch_size": 2}
    a_net, b_net = _net(cuda, "bf16").train(), _net(cuda, "fp16").train()
    b_net.load_state_dict(a_net.state_dict())
    with torch.autocast("cuda", dtype=torch.float16):
        a = a_net(bd())
    b = b_net(bd())
    for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        fa, fb = a["multi_scale_3d_features"][k].features.float(), b["multi_scale_3d_features"][k].features.float()
        assert torch.equal(a["multi_scale_3d_features"][k].indices, b["multi_scale_3d_features"][k].indices)
        scale = float(fb.abs().max())
        assert float((fa - fb).abs().max()) <= 2e-2 * scale, (k, float((fa - fb).abs().max()), scale)
    for (ka, pa), (kb, pb) in zip(a_net.named_buffers(), b_net.named_buffers()):      # running statistics advanced alike
        if ka.endswith("running_mean"):
            assert torch.allclose(pa, pb, rtol=2e-2, atol=2e-3), ka


@pytest.mark.parametrize("scene", ["single_sweep_x2", "ten_sweeps"])
def test_bf16x3_on_calibrated_weights_is_relative_to_the_feature_scale_not_1e4_absolute(cuda, scene):
    """BASELINE.json: "box regressions within 1e-4 fp32" — an ABSOLUTE bound.  VERDICT r04 asked for the test that shows it on
    features of a trained network's magnitude: BatchNorm statistics CALIBRATED on the input (synthetic.calibrate_batchnorm: one
    train-mode forward with momentum 1), so that every stage output has an rms of ~1 (SURVEY Appendix A.6 'values of O(1) after
    BN'; the largest features still reach 30-65 — a ReLU network's tails).  MEASURED, round 5 (two full-grid 30 k-point scenes; one
    10-sweep scene, first forward on the tile rulebooks, second on the gather kernels):

        bf16x3 vs the f32 engine, max |err|:  x_conv1 3.6e-4, x_conv2 8.1e-4, x_conv3 1.3e-3, x_conv4 1.8e-3, out 1.4e-3
        = 1.2-3.7e-5 of the largest feature of each output (conv_out inherits x_conv4's scale: 3.1e-5 of ITS largest feature)

    i.e. the bf16x3 engine does NOT meet the absolute 1e-4 on O(1)-rms features: its error is a property of the split (three
    products keep 2^-17 of every PRODUCT, and the largest products set the absolute error), 3e-5 of the feature scale at any
    magnitude.  The engine that meets the absolute bound is FNP_DTYPE fp32, which is the CPU oracle bit for bit
    (test_fused_backbone_f32_is_the_oracle_bit_for_bit).  This test pins what bf16x3 IS: within 4e-5 of the network's largest
    feature, and at least 95 % of the elements within 1e-4 absolute (measured: 98.3 % at the worst output, x_conv3)."""
    if scene == "ten_sweeps":
        pts, off = syn.make_sweeps_batch((38,))
    else:
        pts, off = syn.make_batch((40, 41))
    B = len(off) - 1
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
    d_p, d_o = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
    ref = _net(cuda, "fp32")
    vox = S.voxelize(d_p, d_o, B, cfg)
    n = int(vox["n"].item())
    syn.calibrate_batchnorm(ref, {"voxel_features": vox["mean"][:n], "voxel_coords": vox["coords"][:n].float(), "batch_size": B})
    x3 = _net(cuda, "bf16x3")
    x3.load_state_dict(ref.state_dict())
    x3.eval()
    with torch.no_grad():
        want = ref.forward_points(d_p, d_o, B, cfg)
        runs = [x3.forward_points(d_p, d_o, B, cfg) for _ in range(2 if scene == "ten_sweeps" else 1)]
    report = {}
    net_scale = max(float(want[k].features.abs().max()) for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"))   # (an output inherits its inputs' error)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        w = want[name].features
        rms, big = float(w.float().pow(2).mean().sqrt()), float(w.abs().max())
        assert 0.02 < rms < 5.0, (name, rms, "calibration should leave O(1) features")
        for got in runs:
            assert torch.equal(got[name].indices, want[name].indices), name
            d = (got[name].features - w).abs()
            err, over = float(d.max()), float((d > 1e-4).float().mean())
            report[name] = (f"{err:.2e}", f"{err / big:.1e} of max {big:.1f}", f"rms {rms:.2f}", f"{over:.1e} of elements over 1e-4")
            assert err <= 4e-5 * max(1.0, net_scale), (name, err, big, net_scale)
            assert over <= 0.05, (name, over)
    print(scene, "bf16x3 vs f32 engine on calibrated weights:", report)
