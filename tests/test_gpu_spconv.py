"""Rulebooks + sparse convolution kernels vs the CPU oracle (rows matched by coordinate, since the
strided output row order is implementation-defined in spconv), the spconv-shaped module API, and
the fused VoxelResBackBone8x in f32 (<= 1e-4, the north-star tolerance) and bf16 mode."""
import numpy as np
import pytest
import torch

from findnpropagate_amd import sparse as S
from findnpropagate_amd import spconv
from findnpropagate_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _random_sparse(rng, B, shape, n, C):
    cells = B * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=n, replace=False)
    b, rem = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    return rng.standard_normal((n, C)).astype(np.float32), np.stack([b, z, y, x], 1).astype(np.int32)


def _key(idx, shape):
    idx = idx.astype(np.int64)
    return ((idx[:, 0] * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]


def _by_coord(feats, idx, shape):
    o = np.argsort(_key(idx, shape))
    return feats[o], idx[o]


def _nbr_to_pairs(nbr, n_out):
    """(K,cap) nbr -> set of (k, in, out)."""
    nbr = nbr[:, :n_out]
    k, o = np.nonzero(nbr >= 0)
    return set(zip(k.tolist(), nbr[k, o].tolist(), o.tolist()))


@pytest.mark.parametrize("ksize", [3, (3, 1, 1)])
def test_subm_rulebook_bit_exact(cuda, oracle, rng, ksize):
    B, shape, n = 3, [13, 50, 47], 4000
    _, idx = _random_sparse(rng, B, shape, n, 1)
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)
    rb = S.rulebook_subm(d_idx, n_dev, grid, ksize)
    pin, pout, pn = oracle.rulebook_subm(idx, shape, ksize)
    want = set()
    for k in range(pin.shape[0]):
        want |= {(k, int(pin[k, p]), int(pout[k, p])) for p in range(pn[k])}
    assert _nbr_to_pairs(rb.nbr.cpu().numpy(), n) == want


@pytest.mark.parametrize("k,s,p", [(3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0), (2, 2, 0)])
def test_strided_rulebook_bit_exact(cuda, oracle, rng, k, s, p):
    B, shape, n = 2, [21, 40, 44], 3000
    _, idx = _random_sparse(rng, B, shape, n, 1)
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)
    rb = S.rulebook_strided(d_idx, n_dev, grid, k, s, p, cap_out=n * 27)
    n_out = int(rb.out_n.item())
    out_idx = rb.out_indices[:n_out].cpu().numpy()
    o_idx, o_shape, pin, pout, pn = oracle.rulebook_strided(idx, shape, k, s, p)
    assert rb.out_shape == o_shape and n_out == o_idx.shape[0]
    # same site set; our rows are in rank-grid (blocked) order, deterministic
    assert np.array_equal(np.sort(_key(out_idx, o_shape)), np.sort(_key(o_idx, o_shape)))
    # same pairs once output rows are renamed by coordinate
    ren = {int(kk): i for i, kk in enumerate(_key(o_idx, o_shape))}
    mine = {(kk, i, ren[int(_key(out_idx[o:o + 1], o_shape)[0])]) for (kk, i, o) in _nbr_to_pairs(rb.nbr.cpu().numpy(), n_out)}
    want = set()
    for kk in range(pin.shape[0]):
        want |= {(kk, int(pin[kk, q]), int(pout[kk, q])) for q in range(pn[kk])}
    assert mine == want


@pytest.mark.parametrize("k,s,p", [(3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)])
@pytest.mark.parametrize("how", ["masked", "lean32", "lean64"])
def test_next_stage_sites_marked_by_the_rulebook_kernel(cuda, rng, k, s, p, how):
    """rulebook_subm(mark_next=...) marks the output sites of the strided layer that consumes its rows while it resolves their
    neighbours; rulebook_strided(premarked=True) then skips its own marking launch: same output sites, same order, same
    table as the plain two-launch form (all three down-sampling geometries of the backbone, all three carrier kernels)."""
    B, shape, n = 2, [21, 40, 44], 5000
    _, idx = _random_sparse(rng, B, shape, n, 1)
    idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)
    plain = S.rulebook_strided(d_idx, n_dev, grid, k, s, p, cap_out=n * 8)
    out_grid = S.alloc_grid(B, plain.out_shape, cuda)
    kw = {"masked": dict(masks=True), "lean32": dict(tile_channels=32, lean_table=True), "lean64": dict(tile_channels=64, lean_table=True)}[how]
    rb = S.rulebook_subm(d_idx, n_dev, grid, 3, mark_next=(out_grid, k, s, p), **kw)
    assert rb._marked_next
    fused = S.rulebook_strided(d_idx, n_dev, grid, k, s, p, cap_out=n * 8, out_grid=out_grid, premarked=True)
    m = int(plain.out_n.item())
    assert int(fused.out_n.item()) == m
    assert torch.equal(fused.out_indices[:m], plain.out_indices[:m])
    assert torch.equal(fused.nbr[:, :m], plain.nbr[:, :m])
    # and the rulebook the carrier built is its usual self
    assert torch.equal(S.rulebook_subm(d_idx, n_dev, grid, 3).nbr[:, :n], rb.nbr[:, :n]) or how != "masked"


@pytest.mark.parametrize("n", [1, 63, 1500])
@pytest.mark.parametrize("Cin,Cout", [(5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (24, 40),
                                      (32, 16), (64, 32), (128, 64)])
def test_conv_f32_bit_exact_vs_oracle(cuda, oracle, rng, Cin, Cout, n):
    """f32 path == the oracle's k-ascending, cin-ascending fmaf chain, bit for bit: on the f32 MFMA kernel
    (v_mfma_f32_16x16x4_f32: the backbone's channel pairs and their transposes for the data gradients), on the first-layer
    kernel (5 -> 16) and on the thread-per-element chain (other shapes, and every shape with valu=True)."""
    B, shape = 2, [9, 30, 31]
    feats, idx = _random_sparse(rng, B, shape, n, Cin)
    w = (rng.standard_normal((Cout, 3, 3, 3, Cin)) * 0.1).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((n, Cout)).astype(np.float32)
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    wp = S.pack_weight(torch.from_numpy(w).to(cuda), torch.float32)
    got = S.conv_forward(torch.from_numpy(feats).to(cuda), wp, rb, n_dev, scale=torch.from_numpy(scale).to(cuda),
                         shift=torch.from_numpy(shift).to(cuda), residual=torch.from_numpy(res).to(cuda), relu=True)
    y = oracle.subm_conv(oracle.SparseTensor(feats, idx, shape, B), w)
    want = oracle.scale_shift_act(y.features, scale, shift, res, relu=True)
    assert np.array_equal(got.cpu().numpy(), want)
    raw = S.conv_forward(torch.from_numpy(feats).to(cuda), wp, rb, n_dev)
    assert np.array_equal(raw.cpu().numpy(), y.features)
    valu = S.conv_forward(torch.from_numpy(feats).to(cuda), wp, rb, n_dev, scale=torch.from_numpy(scale).to(cuda),
                          shift=torch.from_numpy(shift).to(cuda), residual=torch.from_numpy(res).to(cuda), relu=True, valu=True)
    assert torch.equal(valu, got), "matrix-pipe and thread-per-element chains are the same bits"
    wperm = S.pack_weight(torch.from_numpy(w).to(cuda), torch.float32, mfma_f32=True)
    assert isinstance(wperm, S.PermutedWeight) == ((Cin, Cout) in S.F32_MFMA_SHAPES)
    perm = S.conv_forward(torch.from_numpy(feats).to(cuda), wperm, rb, n_dev, scale=torch.from_numpy(scale).to(cuda),
                          shift=torch.from_numpy(shift).to(cuda), residual=torch.from_numpy(res).to(cuda), relu=True)
    assert torch.equal(perm, got), "the pre-permuted weight layout is a layout, not a different sum"


def _round16(a, td):
    a[...] = torch.from_numpy(a).to(td).float().numpy()


@pytest.mark.parametrize("td,out_tol", [(torch.bfloat16, 1e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize("Cin,Cout", [(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("n", [1, 63, 1500])
def test_conv_bf16_mfma_vs_oracle(cuda, oracle, rng, Cin, Cout, n, td, out_tol):
    """bf16 x bf16 (and fp16 x fp16: the reference's AMP mode) products are exact in f32; only the f32 summation order
    differs from the oracle (which is fed the same rounded inputs): tolerance 1e-4 relative to the row scale + one
    rounding of the stored output."""
    B, shape = 2, [9, 30, 31]
    feats, idx = _random_sparse(rng, B, shape, n, Cin)
    w = (rng.standard_normal((Cout, 3, 3, 3, Cin)) * 0.1).astype(np.float32)
    _round16(feats, td)
    _round16(w, td)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((n, Cout)).astype(np.float32)
    _round16(res, td)
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    wp = S.pack_weight(torch.from_numpy(w).to(cuda), td)
    x = torch.from_numpy(feats).to(cuda).to(td)
    y = oracle.subm_conv(oracle.SparseTensor(feats, idx, shape, B), w)
    want = oracle.scale_shift_act(y.features, scale, shift, res, relu=True)
    # f32 output: isolates the accumulation-order difference
    got32 = S.conv_forward(x, wp, rb, n_dev, out_dtype=torch.float32, scale=torch.from_numpy(scale).to(cuda),
                           shift=torch.from_numpy(shift).to(cuda), residual=torch.from_numpy(res).to(cuda), relu=True)
    np.testing.assert_allclose(got32.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    # bf16 output: one extra rounding
    got16 = S.conv_forward(x, wp, rb, n_dev, scale=torch.from_numpy(scale).to(cuda),
                           shift=torch.from_numpy(shift).to(cuda),
                           residual=torch.from_numpy(res).to(cuda).to(td), relu=True)
    assert got16.dtype == td
    np.testing.assert_allclose(got16.float().cpu().numpy(), want, rtol=out_tol, atol=out_tol)
    # run-to-run bit stability (no atomics)
    again = S.conv_forward(x, wp, rb, n_dev, out_dtype=torch.float32, scale=torch.from_numpy(scale).to(cuda),
                           shift=torch.from_numpy(shift).to(cuda), residual=torch.from_numpy(res).to(cuda), relu=True)
    assert torch.equal(got32, again)


def test_module_api_matches_oracle_and_dense(cuda, oracle, rng):
    """spconv-shaped modules (SubMConv3d -> SparseConv3d -> .dense()) in f32."""
    B, shape, n = 2, [11, 24, 26], 900
    feats, idx = _random_sparse(rng, B, shape, n, 8)
    m1 = spconv.SubMConv3d(8, 16, 3, padding=1, bias=False, indice_key="a").to(cuda).eval()
    m2 = spconv.SparseConv3d(16, 32, 3, stride=2, padding=(0, 1, 1), bias=True, indice_key="b").to(cuda).eval()
    with torch.no_grad():
        x = spconv.SparseConvTensor(torch.from_numpy(feats).to(cuda), torch.from_numpy(idx).to(cuda), shape, B)
        y1 = m1(x)
        y1b = m1(y1.replace_feature(x.features))          # second call reuses indice_key 'a'
        y2 = m2(y1)
        dense = y2.dense().cpu().numpy()
    assert "a" in x.indice_dict and torch.equal(y1.features, y1b.features)
    o1 = oracle.subm_conv(oracle.SparseTensor(feats, idx, shape, B), m1.weight.detach().cpu().numpy())
    assert np.array_equal(y1.features.cpu().numpy(), o1.features)
    o2 = oracle.sparse_conv(o1, m2.weight.detach().cpu().numpy(), 2, (0, 1, 1))
    o2.features += m2.bias.detach().cpu().numpy()[None, :]
    assert y2.spatial_shape == o2.spatial_shape and y2.features.shape[0] == o2.features.shape[0]
    got_f, got_i = _by_coord(y2.features.cpu().numpy(), y2.indices.cpu().numpy(), y2.spatial_shape)
    want_f, want_i = _by_coord(o2.features, o2.indices, o2.spatial_shape)
    assert np.array_equal(got_i, want_i)
    np.testing.assert_allclose(got_f, want_f, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dense, o2.dense(), rtol=1e-6, atol=1e-6)
    # a call with autograd enabled returns a tensor that carries the graph (tests/test_gpu_backward.py checks the values)
    m1.train()
    yt = m1(x)
    assert yt.features.requires_grad and yt.features.grad_fn is not None
    assert torch.equal(yt.features.detach(), y1.features)


def _small_net(cuda, dtype):
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

    grid_size = np.array([96, 88, 40])  # x, y, z -> sparse shape [41, 88, 96]
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": dtype}, 5, grid_size), seed=1)
    return net.to(cuda).eval()


def _check_stage(got, want, rtol, atol, bf16=False):
    gf, gi = _by_coord(got.features.float().cpu().numpy(), got.indices.cpu().numpy(), got.spatial_shape)
    wf, wi = _by_coord(want.features, want.indices, want.spatial_shape)
    assert got.spatial_shape == want.spatial_shape
    assert np.array_equal(gi, wi), "active site sets differ"
    if not bf16:
        np.testing.assert_allclose(gf, wf, rtol=rtol, atol=atol)
        return
    # bf16 storage: a value that lands within an ulp of a rounding boundary may round the other way
    # than in the oracle's emulation (different f32 summation order) and the flip propagates through
    # the following layers.  Gate: such elements are rare, and no element is off by more than a few
    # bf16 ulps (2^-8) of the tensor's scale.
    bad = np.abs(gf - wf) > atol + rtol * np.abs(wf)
    assert bad.mean() < 1e-2, f"{bad.sum()} of {bad.size} elements outside rtol/atol"
    assert np.abs(gf - wf).max() <= 4 * 2.0 ** -8 * max(1.0, np.abs(wf).max())


def test_fused_backbone_f32_within_1e4(cuda, oracle, rng):
    net = _small_net(cuda, "fp32")
    shape = net.sparse_shape
    feats, idx = _random_sparse(rng, 2, shape, 6000, 5)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want = oracle.backbone_forward(sd, feats, idx, 2, shape)
    bd = {"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda).float(),
          "batch_size": 2}
    with torch.no_grad():
        out = net(bd)
    assert out["encoded_spconv_tensor_stride"] == 8
    assert out["multi_scale_3d_strides"] == {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}
    x1 = out["multi_scale_3d_features"]["x_conv1"]
    assert np.array_equal(x1.indices.cpu().numpy(), idx), "SubM keeps the input row order"
    for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        _check_stage(out["multi_scale_3d_features"][k], want[k], 1e-4, 1e-4)
    _check_stage(out["encoded_spconv_tensor"], want["out"], 1e-4, 1e-4)
    assert out["encoded_spconv_tensor"].features.shape[1] == 128
    # HeightCompression: dense (B, 128, 2, H/8, W/8) -> (B, 256, H/8, W/8)
    d = out["encoded_spconv_tensor"].dense()
    assert list(d.shape) == [2, 128, 2, 11, 12]
    np.testing.assert_allclose(d.cpu().numpy(), want["out"].dense(), rtol=1e-4, atol=1e-4)
    # module path (unfused BN in torch) agrees with the fused path
    net.train()
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.eval()
        out2 = net({"voxel_features": bd["voxel_features"], "voxel_coords": bd["voxel_coords"], "batch_size": 2})
    net.eval()
    _check_stage(out2["encoded_spconv_tensor"], want["out"], 1e-4, 1e-4)


@pytest.mark.parametrize("sort_min_rows", [0, 1 << 30])
def test_fused_backbone_f32_is_the_oracle_bit_for_bit(cuda, oracle, rng, sort_min_rows, monkeypatch):
    """The fp32 engine (BASELINE.json's 1e-4 mode) is not merely within 1e-4 of the CPU oracle: every convolution is the
    oracle's k-ascending fmaf chain on v_mfma_f32_16x16x4_f32, the BatchNorm fold runs in the oracle's IEEE arithmetic on
    the host, the epilogue is one multiply, one add (+ residual add) and a compare — so all five stage outputs are EQUAL.
    Both with every SubM stage swept in class-sorted order (sort_min_rows 0: what full batches run) and in row order."""
    monkeypatch.setattr(S, "F32_SORT_MIN_ROWS", sort_min_rows)
    net = _small_net(cuda, "fp32")
    shape = net.sparse_shape
    feats, idx = _random_sparse(rng, 2, shape, 6000, 5)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want = oracle.backbone_forward(sd, feats, idx, 2, shape)
    with torch.no_grad():
        out = net({"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda), "batch_size": 2})
    stages = dict(out["multi_scale_3d_features"], out=out["encoded_spconv_tensor"])
    for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        gf, gi = _by_coord(stages[k].features.cpu().numpy(), stages[k].indices.cpu().numpy(), stages[k].spatial_shape)
        wf, wi = _by_coord(want[k].features, want[k].indices, want[k].spatial_shape)
        assert np.array_equal(gi, wi), k
        assert np.array_equal(gf, wf), (k, float(np.abs(gf - wf).max()))


def test_tile_timeout_is_an_error_not_wrong_features(cuda, rng):
    """A hand-over wait of the tiled 32-channel kernel that times out ends its workgroup with rows unwritten.  The engine
    reads the library's time-out counter with the per-stage counts of its one host sync and raises; the next forward,
    with the protocol intact again, is clean.  (fnp_debug_tile_hold makes the producers stop publishing images.)"""
    from findnpropagate_amd import lib as _lib
    net = _small_net(cuda, "bf16")
    shape = net.sparse_shape
    feats, idx = _random_sparse(rng, 1, shape, 3000, 5)
    bd = lambda: {"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda), "batch_size": 1}
    L = _lib.load()
    with torch.no_grad():
        ref = net(bd())["encoded_spconv_tensor"].features.clone()
        before = L.fnp_spconv_tiled_aborts()
        assert L.fnp_debug_tile_hold(1) == 0
        try:
            with pytest.raises(_lib.FnpError, match="timed out"):
                net(bd())
        finally:
            assert L.fnp_debug_tile_hold(0) == 0
        assert L.fnp_spconv_tiled_aborts() > before
        again = net(bd())["encoded_spconv_tensor"].features
    assert torch.equal(ref, again)


def test_fused_backbone_bf16(cuda, oracle, rng):
    net = _small_net(cuda, "bf16")
    shape = net.sparse_shape
    feats, idx = _random_sparse(rng, 2, shape, 6000, 5)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want = oracle.backbone_forward(sd, feats, idx, 2, shape, bf16=True)
    want32 = oracle.backbone_forward(sd, feats, idx, 2, shape)
    bd = {"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda), "batch_size": 2}
    with torch.no_grad():
        out = net(bd)
    # what crosses the reference boundary is float32 (the reference's BEV backbone and heads are f32 modules), whatever the
    # engine stores internally; conv_out's epilogue writes it directly (its last rounding to bf16 is simply skipped)
    assert out["encoded_spconv_tensor"].features.dtype == torch.float32
    assert all(t.features.dtype == torch.float32 for t in out["multi_scale_3d_features"].values())
    net.fnp_out_dtype = "native"
    with torch.no_grad():
        nat = net({"voxel_features": bd["voxel_features"], "voxel_coords": bd["voxel_coords"], "batch_size": 2})
    net.fnp_out_dtype = "fp32"
    assert nat["encoded_spconv_tensor"].features.dtype == torch.bfloat16
    assert torch.equal(nat["encoded_spconv_tensor"].features, out["encoded_spconv_tensor"].features.bfloat16())
    assert torch.equal(nat["multi_scale_3d_features"]["x_conv3"].features.float(), out["multi_scale_3d_features"]["x_conv3"].features)
    from findnpropagate_amd.backbones_2d import HeightCompression
    bev = HeightCompression({"NUM_BEV_FEATURES": 256})(dict(out))["spatial_features"]
    assert bev.dtype == torch.float32 and list(bev.shape) == [2, 256, shape[1] // 8, shape[2] // 8]
    # vs the bf16-emulating oracle: differences are isolated bf16 rounding flips (<= 1 bf16 ulp,
    # 2^-8 relative) that propagate through 21 layers
    for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        _check_stage(out["multi_scale_3d_features"][k], want[k], 3e-2, 3e-2, bf16=True)
    _check_stage(out["encoded_spconv_tensor"], want["out"], 3e-2, 3e-2, bf16=True)
    # vs the f32 oracle: report the bf16 storage error (not a parity gate)
    g, _ = _by_coord(out["encoded_spconv_tensor"].features.float().cpu().numpy(),
                     out["encoded_spconv_tensor"].indices.cpu().numpy(), want32["out"].spatial_shape)
    w, _ = _by_coord(want32["out"].features, want32["out"].indices, want32["out"].spatial_shape)
    rel = np.abs(g - w).max() / (np.abs(w).max() + 1e-6)
    assert rel < 0.1, rel
    # bit-identical across runs
    with torch.no_grad():
        out2 = net({"voxel_features": bd["voxel_features"], "voxel_coords": bd["voxel_coords"], "batch_size": 2})
    assert torch.equal(out["encoded_spconv_tensor"].features, out2["encoded_spconv_tensor"].features)


def test_forward_points_full_size_properties(cuda, oracle):
    """BASELINE-size run (3 scenes x 30k points, 41x1440x1440 grid): voxel rows bit-exact vs the
    oracle voxeliser, stage site counts == oracle rulebook counts, SubM keeps voxel order."""
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x

    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), seed=0).to(cuda).eval()
    assert net.sparse_shape == [41, 1440, 1440]
    seeds = (11, 12, 13)
    pts, off = syn.make_batch(seeds)
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    with torch.no_grad():
        res = net.forward_points(torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda), 3, cfg)
        res2 = net.forward_points(torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda), 3, cfg)
    coords = []
    for b, s in enumerate(seeds):
        _, c, _ = oracle.voxelize(syn.make_scene(s), syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 10, 160000)
        coords.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    coords = np.concatenate(coords)
    assert np.array_equal(res["voxel_coords"].cpu().numpy(), coords)
    assert np.array_equal(res["x_conv1"].indices.cpu().numpy(), coords)
    shape, idx = [41, 1440, 1440], coords
    for name, (k, s, p) in (("x_conv2", (3, 2, 1)), ("x_conv3", (3, 2, 1)), ("x_conv4", (3, 2, (0, 1, 1))), ("out", ((3, 1, 1), (2, 1, 1), 0))):
        idx, shape, _, _, _ = oracle.rulebook_strided(idx, shape, k, s, p)
        got = res[name]
        assert got.spatial_shape == shape and got.indices.shape[0] == idx.shape[0]
        assert np.array_equal(np.sort(_key(got.indices.cpu().numpy(), shape)), np.sort(_key(idx, shape)))
        assert torch.isfinite(got.features.float()).all()
    assert res["out"].spatial_shape == [2, 180, 180]
    assert torch.equal(res["out"].features, res2["out"].features), "persistent grids were left clean; rerun is bit-identical"
    # features of the multi-scene batch vs the oracle run scene by scene... as one batch (bf16 emulation)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want = oracle.backbone_forward(sd, res["voxel_features"].cpu().numpy(), coords, 3, [41, 1440, 1440], bf16=True)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        _check_stage(res[name], want[name], 3e-2, 3e-2, bf16=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape,C,B,n", [((2, 180, 180), 128, 3, 9000), ((2, 11, 12), 128, 2, 150), ((3, 7, 5), 6, 1, 40),
                                          ((1, 9, 70), 33, 2, 300), ((2, 16, 16), 16, 2, 0)])
def test_dense_writes_every_element_once(cuda, rng, dtype, shape, C, B, n):
    """fnp_sparse_to_dense with a workspace (height_compression.py:20-24): exact copy of the rows, zeros
    elsewhere, on a DIRTY output buffer; odd row sizes fall back to memset + scatter; and the
    HeightCompression module view."""
    from findnpropagate_amd import sparse as S
    from findnpropagate_amd import spconv
    from findnpropagate_amd.backbones_2d import HeightCompression
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    feats, idx = _random_sparse(rng, B, shape, max(n, 1), C)
    feats, idx = feats[:n], idx[:n]
    f = torch.from_numpy(feats).to(cuda).to(td)
    cap = max(n, 1) + 7                                   # rows beyond n are garbage and must be ignored
    fpad = torch.full((cap, C), 7.0, device=cuda, dtype=td)
    ipad = torch.full((cap, 4), 0, device=cuda, dtype=torch.int32)
    fpad[:n] = f
    ipad[:n] = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    out = torch.full((B, C, *shape), 3.0, device=cuda, dtype=td)          # dirty
    got = S.to_dense(fpad, ipad, n_dev, B, list(shape), out=out)
    want = torch.zeros((B, C, *shape), dtype=td)
    if n:
        ii = torch.from_numpy(idx).long()
        want[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]] = f.cpu()
    assert torch.equal(got.cpu(), want)
    if n:
        t = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(cuda), list(shape), B)
        hc = HeightCompression({"NUM_BEV_FEATURES": C * shape[0]})
        bd = hc({"encoded_spconv_tensor": t, "encoded_spconv_tensor_stride": 8})
        assert bd["spatial_features_stride"] == 8 and hc.num_bev_features == C * shape[0]
        assert torch.equal(bd["spatial_features"].cpu(), want.view(B, C * shape[0], shape[1], shape[2]))


@pytest.mark.gpu
def test_graphed_forward_equals_eager(cuda):
    """forward_points_graphed (hipGraph replay, padded static inputs) returns exactly what forward_points
    returns, for inputs of different sizes replayed on the same captured graph."""
    from findnpropagate_amd import sparse as S, synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    n_graphs = []
    for seeds in ([0, 1], [2, 3], [4, 5]):
        pts, off = syn.make_batch(seeds)
        if seeds == [4, 5]:
            pts, off = pts[: off[2] - 5000], off.copy()    # fewer points than the call before: stale padding must go
            off[2] = pts.shape[0]
        pts, off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
        with torch.no_grad():
            want = net.forward_points(pts, off, 2, cfg)
            want = {k: (v.features.clone(), v.indices.clone()) if hasattr(v, "features") else v for k, v in want.items()}
            got = net.forward_points_graphed(pts, off, 2, cfg, capacity=80000)
        assert got["counts"] == want["counts"]
        for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
            assert torch.equal(got[k].features, want[k][0]) and torch.equal(got[k].indices, want[k][1]), k
        assert torch.equal(got["voxel_coords"], want["voxel_coords"]) and torch.equal(got["voxel_features"], want["voxel_features"])
        n_graphs.append(len(net.engine()._graphs))
    assert n_graphs == [1, 1, 1], "one capture serves every call of that (batch, capacity)"


@pytest.mark.gpu
def test_counts_launch_hands_the_counts_to_the_host_itself(cuda):
    """fnp_gather_counts_host (include/fnp.h, ABI 14): the launch stores the counts into pinned host memory and, behind them, its own
    sequence number; a host thread that polls for that number finds the counts — inside a replayed hipGraph too (no event, no copy):
    how a captured forward hands its stage counts over in the middle of its one graph."""
    import ctypes
    import time
    from findnpropagate_amd import lib as _lib
    L = _lib.load()
    a = torch.tensor([7], dtype=torch.int32, device=cuda)
    b = torch.tensor([11], dtype=torch.int32, device=cuda)
    pool = torch.tensor([5], dtype=torch.int32, device=cuda)      # (reset after it is read: bit 3)
    dst = torch.zeros(4, dtype=torch.int32, device=cuda)
    seq = torch.zeros(1, dtype=torch.int32, device=cuda)
    pin = torch.zeros(32, dtype=torch.int32, pin_memory=True)
    word = pin.numpy()
    arr = (ctypes.c_void_p * 4)(a.data_ptr(), b.data_ptr(), None, pool.data_ptr())

    def launch():
        _lib.check(L.fnp_gather_counts_host(ctypes.cast(arr, ctypes.c_void_p), 4, 1 << 3, _lib.ptr(dst), ctypes.c_void_p(pin.data_ptr()), _lib.ptr(seq),
                                            _lib.stream()), "fnp_gather_counts_host")

    def wait(want):
        t0 = time.monotonic()
        while int(word[16]) != want:
            assert time.monotonic() - t0 < 20.0, "the sequence number never arrived"

    aborts = L.fnp_spconv_tiled_aborts()
    launch()
    wait(1)
    assert [int(v) for v in word[:4]] == [7, 11, aborts, 5]
    torch.cuda.synchronize()
    assert dst.tolist() == [7, 11, aborts, 5] and int(pool.item()) == 0 and int(seq.item()) == 1
    # as a node of a hipGraph, replayed: the sources change between replays, the sequence number counts the replays
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            a.add_(1)
            launch()
            b.mul_(2)       # (work behind the counts launch: the host does not wait for it)
        for r in range(3):
            g.replay()
            wait(2 + r)
            assert [int(v) for v in word[:4]] == [8 + r, 11 * 2 ** r, aborts, 0]
    torch.cuda.synchronize()
    assert L.fnp_gather_counts_host(None, 4, 0, _lib.ptr(dst), ctypes.c_void_p(pin.data_ptr()), _lib.ptr(seq), _lib.stream()) != 0
    assert L.fnp_gather_counts_host(ctypes.cast(arr, ctypes.c_void_p), 17, 0, _lib.ptr(dst), ctypes.c_void_p(pin.data_ptr()), _lib.ptr(seq), _lib.stream()) != 0


@pytest.mark.gpu
def test_a_frame_is_staged_into_static_inputs_by_one_launch(cuda, rng):
    """fnp_stage_points: rows [0, n) = the frame, rows [n, n_prev) = the padding value, the offsets copied — what three stream
    operations did before a captured forward's replay."""
    from findnpropagate_amd.backbones_3d.spconv_backbone import _PointsGraph
    dst = torch.full((5000, 5), -7.0, dtype=torch.float32, device=cuda)
    off_dst = torch.zeros(4, dtype=torch.int32, device=cuda)
    for n, n_prev in ((3000, 0), (1200, 3000), (0, 1200), (4999, 0), (5000, 4999)):
        pts = torch.from_numpy(rng.normal(size=(n, 5)).astype(np.float32)).to(cuda)
        off = torch.tensor([0, n // 3, n // 2, n], dtype=torch.int32, device=cuda)
        before = dst.clone()
        S.stage_points(pts, off, dst, off_dst, n_prev, _PointsGraph.FAR)
        want = before
        want[:n] = pts
        if n_prev > n:
            want[n:n_prev] = _PointsGraph.FAR
        assert torch.equal(dst, want) and torch.equal(off_dst, off), (n, n_prev)


@pytest.mark.gpu
def test_streams_are_tested_to_run_beside_each_other(cuda):
    """concurrent_streams: distinct streams, none of them the caller's, each seen to make progress while a spin kernel holds another."""
    from findnpropagate_amd.backbones_3d.spconv_backbone import concurrent_streams
    caller = torch.cuda.current_stream(cuda)
    st, ok = concurrent_streams(cuda, 3, beside=[caller], report=True)
    assert len(st) == 3 and len({s.cuda_stream for s in st}) == 3 and all(s != caller for s in st)
    assert isinstance(ok, bool)
    x = torch.zeros(8, device=cuda)
    for s in st:
        with torch.cuda.stream(s):
            x.add_(1)
        torch.cuda.current_stream(cuda).wait_stream(s)
    torch.cuda.synchronize()
    assert float(x.sum()) == 24.0
    assert concurrent_streams("cpu", 2) == []


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_points_pipeline_equals_eager_frame_by_frame(cuda, depth):
    """PointsPipeline (one-scene frames, `depth` hipGraph replays in flight on their own streams, each slot with its own
    engine): every frame's result equals forward_points', in submission order; a frame whose stage capacities are too small
    is rerun through the engine's own overflow loop."""
    from findnpropagate_amd import sparse as S, synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    frames, want = [], []
    for seed in (0, 1, 2, 3, 4, 5, 6):
        pts, off = syn.make_batch([seed])
        if seed == 4:
            pts, off = pts[:20000].copy(), np.array([0, 20000], np.int32)   # a shorter frame: the slot's stale padding must go
        pts, off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
        frames.append((pts, off))
        with torch.no_grad():
            w = net.forward_points(pts, off, 1, cfg)
        want.append({k: (w[k].features.clone(), w[k].indices.clone()) for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out")} | {"counts": w["counts"]})
    pipe = net.points_pipeline(1, cfg, depth=depth, capacity=65536)
    with torch.no_grad():
        for i, got in enumerate(pipe.map(frames)):
            assert got["counts"] == want[i]["counts"], i
            for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
                assert torch.equal(got[k].features, want[i][k][0]) and torch.equal(got[k].indices, want[i][k][1]), (i, k)
        assert i == len(frames) - 1 and not pipe.pending
        # capacities far too small in one slot: that frame goes through the engine's overflow loop, the others are untouched
        pipe.engines[0].cap_factor = [0.5, 0.2, 0.1, 0.05]
        pipe.slots[0] = None
        for i, got in enumerate(pipe.map(frames[:3])):
            for k in ("x_conv2", "out"):
                assert torch.equal(got[k].features, want[i][k][0]) and torch.equal(got[k].indices, want[i][k][1]), (i, k)
        assert pipe.engines[0].cap_factor[0] > 0.5


@pytest.mark.gpu
def test_batch_path_with_two_batches_in_flight_equals_forward_points(cuda):
    """forward_points_iter (round 6: the batch path's default — two batches in flight) and the PROBE form of the pipeline that
    bench.py times (every slot a two-graph capture, the four 128 -> 128 launches between event pairs on the slot's stream):
    batch by batch the tensors of forward_points."""
    from findnpropagate_amd import sparse as S, synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    batches, want = [], []
    for seeds in ([0, 1], [2, 3], [4, 5], [6, 7], [8, 9]):
        pts, off = syn.make_batch(seeds)
        pts, off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
        batches.append((pts, off))
        with torch.no_grad():
            w = net.forward_points(pts, off, 2, cfg)
        want.append({k: (w[k].features.clone(), w[k].indices.clone()) for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out")} | {"counts": w["counts"]})
    keys = ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out")
    with torch.no_grad():
        n = 0
        for i, got in enumerate(net.forward_points_iter(iter(batches), 2, cfg)):
            assert got["counts"] == want[i]["counts"], i
            for k in keys:
                assert torch.equal(got[k].features, want[i][k][0]) and torch.equal(got[k].indices, want[i][k][1]), (i, k)
            n += 1
        assert n == len(batches)
        # a second call with the same configuration reuses the first call's pipeline (captured slots, tested streams) — also after an
        # iterator that was dropped with batches still in flight — and a different configuration makes a new one
        kept = net._iter_pipeline[1]
        it = net.forward_points_iter(iter(batches), 2, cfg)
        next(it)
        del it
        for i, got in enumerate(net.forward_points_iter(iter(batches[::-1]), 2, cfg)):
            w_ = want[len(batches) - 1 - i]
            assert got["counts"] == w_["counts"], i
            for k in keys:
                assert torch.equal(got[k].features, w_[k][0]) and torch.equal(got[k].indices, w_[k][1]), (i, k)
        assert net._iter_pipeline[1] is kept and not kept.pending
        for got in net.forward_points_iter(iter(batches[:1]), 2, cfg, depth=3):
            assert got["counts"] == want[0]["counts"]
        assert net._iter_pipeline[1] is not kept and net._iter_pipeline[1].depth == 3
        # serial_convs: every slot's capture cut behind its index chain, the convolution graphs of all slots on one stream (the
        # default from 8 scenes per batch on); False: whole forwards in flight side by side
        for serial, depth in ((True, 2), (False, 2), (True, 3)):
            pipe = net.points_pipeline(2, cfg, depth=depth, capacity=131072, probe=True, serial_convs=serial)
            assert pipe.serial_convs == serial
            pipe.profile = []
            for rep in range(2):
                for i, got in enumerate(pipe.map(batches)):
                    assert got["counts"] == want[i]["counts"], (serial, i)
                    for k in keys:
                        assert torch.equal(got[k].features, want[i][k][0]) and torch.equal(got[k].indices, want[i][k][1]), (serial, i, k)
            torch.cuda.synchronize()
            assert len(pipe.profile) == 2 * 4 * len(batches) and all(t[:3] == (128, 128, 27) for t, _, _ in pipe.profile)
            assert all(e0.elapsed_time(e1) > 0 for _, e0, e1 in pipe.profile)
            del pipe


@pytest.mark.gpu
def test_capacity_overflow_regrows_eager_and_graphed(cuda):
    """Stage capacities that are too small are detected after the step (true counts live on the device),
    grown, the persistent grids wiped, and the step repeated — eager and hipGraph paths, same results."""
    from findnpropagate_amd import sparse as S, synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    pts, off = syn.make_batch([3])
    pts, off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
    with torch.no_grad():
        want = net.forward_points(pts, off, 1, cfg)
        want = {k: (want[k].features.clone(), want[k].indices.clone()) for k in ("x_conv2", "x_conv4", "out")}
        for graphed in (False, True):
            net.engine().cap_factor = [0.5, 0.2, 0.1, 0.05]          # far too small for stages 2-5
            net.engine()._graphs.clear()
            got = (net.forward_points_graphed(pts, off, 1, cfg) if graphed else net.forward_points(pts, off, 1, cfg))
            assert net.engine().cap_factor[0] > 0.5 and net.engine().cap_factor[3] > 0.05
            for k in want:
                assert torch.equal(got[k].features, want[k][0]) and torch.equal(got[k].indices, want[k][1]), (graphed, k)
            again = net.forward_points(pts, off, 1, cfg)              # the grids were left clean
            assert torch.equal(again["out"].features, want["out"][0])


@pytest.mark.gpu
def test_dropped_voxels_leave_persistent_grids_clean(cuda):
    """max_voxels cuts voxels: their cells were marked in the persistent stage-1 grid but have no row.  The
    voxeliser lists them behind the voxels (n_cells) so the sparse clear reaches them: a second call on the
    same engine (other scene, then the same scene) equals a fresh engine's result."""
    from findnpropagate_amd import sparse as S, synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 6000)     # ~19k voxels per scene: 2/3 dropped
    batches = []
    for seeds in ([0, 1], [2, 3]):
        pts, off = syn.make_batch(seeds)
        batches.append((torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)))

    def fresh(b):
        net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
        with torch.no_grad():
            r = net.forward_points(*batches[b], 2, cfg)
        return net, r

    net, r0 = fresh(0)
    _, r1 = fresh(1)
    assert r0["counts"][0] == 12000 and int(S.voxelize(*batches[0], 2, cfg)["n_cells"].item()) > 30000
    with torch.no_grad():
        a1 = net.forward_points(*batches[1], 2, cfg)     # same engine, grids left by batch 0
        a0 = net.forward_points(*batches[0], 2, cfg)
    for got, want in ((a1, r1), (a0, r0)):
        assert got["counts"] == want["counts"]
        assert torch.equal(got["voxel_coords"], want["voxel_coords"])
        for k in ("x_conv1", "x_conv3", "out"):
            assert torch.equal(got[k].indices, want[k].indices) and torch.equal(got[k].features, want[k].features), k
    for g in net.engine()._get_grids(2, cuda):
        assert int(g.bits.count_nonzero()) == 0 and int(g.summary.count_nonzero()) == 0


@pytest.mark.parametrize("B,shape,n", [(2, [41, 300, 300], 40000),      # 23 k summary words: one wave per word
                                       (3, [41, 1440, 1440], 150000),   # 67 k: 8 words per wave
                                       (16, [41, 1440, 1440], 400000)]) # 356 k: 64 words per wave
def test_rank_grid_prefix_forms_agree(cuda, rng, B, shape, n):
    """The rank-grid prefix picks its work split (1 / 8 / 64 summary words per wave) from the grid size.  Every
    split must yield the same thing: rank -> row of a clustered coordinate list is a bijection (a 1x1x1 SubM
    rulebook maps every row to itself), and the coordinates emitted in rank order by the strided path (1x1x1,
    stride 1: outputs = inputs) are the input set, each output row fed by the input row with its coordinates."""
    # clustered occupancy (blobs) so that summary words hold several blocks and blocks several cells
    centres = np.stack([rng.integers(0, B, 400), rng.integers(0, shape[0], 400), rng.integers(0, shape[1], 400),
                        rng.integers(0, shape[2], 400)], 1)
    pick = centres[rng.integers(0, 400, 3 * n)]
    jit = np.round(rng.standard_normal((3 * n, 4)) * np.array([0, 2, 12, 12])).astype(np.int64)
    c = pick + jit
    ok = (c[:, 1] >= 0) & (c[:, 1] < shape[0]) & (c[:, 2] >= 0) & (c[:, 2] < shape[1]) & (c[:, 3] >= 0) & (c[:, 3] < shape[2])
    c = np.unique(c[ok], axis=0)
    c = c[rng.permutation(c.shape[0])][:n].astype(np.int32)
    N = c.shape[0]
    idx = torch.from_numpy(c).to(cuda)
    n_dev = S.device_scalar(N, cuda)
    grid = S.build_grid(idx, n_dev, B, shape)
    rb = S.rulebook_subm(idx, n_dev, grid, 1)
    assert torch.equal(rb.nbr[0, :N].cpu(), torch.arange(N, dtype=torch.int32))
    rbs = S.rulebook_strided(idx, n_dev, grid, 1, 1, 0, N + 7)
    assert int(rbs.out_n.item()) == N
    out = rbs.out_indices[:N].cpu().numpy()
    assert np.array_equal(np.sort(_key(out, shape)), np.sort(_key(c, shape)))
    src = rbs.nbr[0, :N].cpu().numpy()
    assert src.min() >= 0 and np.array_equal(c[src], out)


@pytest.mark.parametrize("n", [200, 5000])
def test_window_kernel_equals_gather_kernel(cuda, rng, n):
    """FNP_HINT_ROWS_RANKED only changes where a 64 -> 64 layer finds its rows (LDS window + the lane-row swap
    epilogue instead of gathers + the strip epilogue): same products, same order, bit-identical output — also when
    the rows are NOT in rank order (the hint is a performance statement, never a correctness one)."""
    B, shape, C = 2, [9, 40, 41], 64
    feats, idx = _random_sparse(rng, B, shape, n, C)
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    wp = S.pack_weight(torch.from_numpy((rng.standard_normal((C, 3, 3, 3, C)) * 0.1).astype(np.float32)).to(cuda), torch.bfloat16)
    x = torch.from_numpy(feats).to(cuda).to(torch.bfloat16)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).to(cuda)
    sh = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(cuda)
    res = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)).to(cuda).to(torch.bfloat16)
    for residual in (None, res):
        a = S.conv_forward(x, wp, rb, n_dev, scale=sc, shift=sh, residual=residual, relu=True, ranked=False)
        b = S.conv_forward(x, wp, rb, n_dev, scale=sc, shift=sh, residual=residual, relu=True, ranked=True)
        assert torch.equal(a[:n], b[:n])


@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,order", [(1, "random"), (200, "random"), (257, "sorted"), (5000, "random"), (5000, "sorted"), (40000, "sorted")])
def test_tile_kernel_equals_gather_kernel(cuda, rng, n, order, dtype, C):
    """The tile-rulebook kernels of the ranked 32 -> 32 and 64 -> 64 layers (spconv_tile.hip: the rulebook restated per
    tile as LDS addresses of a window, deduplicated far rows and escapes) against spconv_mfma_kernel: same products, same
    order, bit-identical output.  Rows in random order put almost every neighbour outside the window (the overflow rows
    run out: the escape path carries the layer); rows sorted by cell put most of them inside."""
    B, shape = 2, [9, 40, 41] if n <= 5000 else [21, 80, 80]
    feats, idx = _random_sparse(rng, B, shape, n, C)
    if order == "sorted":
        idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    wp = S.pack_weight(torch.from_numpy((rng.standard_normal((C, 3, 3, 3, C)) * 0.1).astype(np.float32)).to(cuda), dtype)
    x = torch.from_numpy(feats).to(cuda).to(dtype)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).to(cuda)
    sh = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(cuda)
    res = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)).to(cuda).to(dtype)
    for residual, scale, relu in ((None, sc, True), (res, sc, True), (res, None, False)):
        shift = sh if scale is not None else None
        a = S.conv_forward(x, wp, rb, n_dev, scale=scale, shift=shift, residual=residual, relu=relu, ranked=False)
        b = S.conv_forward(x, wp, rb, n_dev, scale=scale, shift=shift, residual=residual, relu=relu, ranked=True, tile=True)
        c = S.conv_forward(x, wp, rb, n_dev, scale=scale, shift=shift, residual=residual, relu=relu, ranked=True, tile=False)
        assert torch.equal(a[:n], b[:n]) and torch.equal(a[:n], c[:n])
    # the rulebook kernel that writes the tile rulebook itself, with the full table and with the lean one (int32 rows of escape
    # tiles only: everything the tiled kernel may ask for): same output; any other kernel on the lean rulebook is refused
    grid2 = S.build_grid(d_idx, n_dev, B, shape)
    for lean in (False, True):
        rb2 = S.rulebook_subm(d_idx, n_dev, grid2, 3, tile_channels=C, lean_table=lean)
        d = S.conv_forward(x, wp, rb2, n_dev, scale=sc, shift=sh, residual=res, relu=True, ranked=True, tile=True)
        assert torch.equal(d[:n], S.conv_forward(x, wp, rb, n_dev, scale=sc, shift=sh, residual=res, relu=True, ranked=False)[:n]), lean
    with pytest.raises(S._l.FnpError, match="escape tiles only"):
        S.conv_forward(x, wp, rb2, n_dev, ranked=True, tile=False)


def test_bf16x3_engine_is_f32_grade_on_the_bf16_matrix_pipe(cuda, oracle):
    """FNP_DTYPE: bf16x3 — activations as f32 rows + their split x = hi + lo into two bf16 tensors (fnp_split_bf16), weights (with
    the BatchNorm scale folded in) as W_hi + W_lo, every convolution three launches of the bf16 MFMA kernels with f32 outputs
    chained through `residual` (lo x W_hi, hi x W_lo, hi x W_hi) — against the f32 CPU oracle: same site sets, and every one of
    the five outputs within 3e-5 of its feature scale (measured 0.7-1.5e-5; the bf16 engine: 4e-3; the f32 engine: 0).  The
    split itself: hi + lo == x to 2^-17 |x|, rows past n untouched."""
    x = torch.randn((1000, 64), device=cuda) * torch.logspace(-3, 3, 64, device=cuda)
    n_dev = S.device_scalar(900, cuda)
    hi, lo = S.split_bf16(x, n_dev)
    assert torch.equal(hi[:900], x[:900].bfloat16())
    assert ((hi[:900].float() + lo[:900].float() - x[:900]).abs() <= x[:900].abs() * 2.0 ** -16).all()
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    rng_hi = [-14.4, -14.4, -5.0, 14.4, 14.4, 3.0]
    grid = np.round((np.array(rng_hi[3:]) - np.array(rng_hi[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    pts = syn.make_scene(3)
    pts = pts[(np.abs(pts[:, 0]) < 14.4) & (np.abs(pts[:, 1]) < 14.4)]
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, rng_hi, 5, 10, 160000)
    off = torch.tensor([0, pts.shape[0]], dtype=torch.int32, device=cuda)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": "bf16x3"}, 5, grid), 0).to(cuda).eval()
    with torch.no_grad():
        res = net.forward_points(torch.from_numpy(pts).to(cuda), off, 1, cfg)
        res2 = net.forward_points(torch.from_numpy(pts).to(cuda), off, 1, cfg)
    v, c, n = oracle.voxelize(pts, syn.VOXEL_SIZE, rng_hi, 10, 160000)
    coords = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    assert np.array_equal(res["voxel_coords"].cpu().numpy(), coords)
    sd = {k: t.detach().cpu().numpy() for k, t in net.state_dict().items()}
    want = oracle.backbone_forward(sd, oracle.mean_vfe(v, n), coords, 1, net.sparse_shape)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        got, w = res[name], want[name]
        assert got.features.dtype == torch.float32
        gi, wi = got.indices.cpu().numpy(), w.indices
        go, wo = np.argsort(_key(gi, got.spatial_shape)), np.argsort(_key(wi, w.spatial_shape))
        assert np.array_equal(gi[go], wi[wo]), name
        g, f = got.features.cpu().numpy()[go], w.features[wo]
        assert np.abs(g - f).max() <= 3e-5 * max(1.0, np.abs(f).max()), (name, np.abs(g - f).max(), np.abs(f).max())
        assert torch.equal(got.features, res2[name].features), name     # (deterministic; grids left clean)
    with torch.no_grad():    # ... and replayed from the captured hipGraphs
        res3 = net.forward_points_graphed(torch.from_numpy(pts).to(cuda), off, 1, cfg)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        assert torch.equal(res[name].features, res3[name].features) and torch.equal(res[name].indices, res3[name].indices), name


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,n", [(16, 16, 3000), (16, 32, 3000), (32, 32, 1), (32, 32, 5000), (64, 64, 513), (64, 64, 5000), (64, 128, 3000),
                                        (128, 128, 3000), (128, 128, 40000), (32, 32, 40000), (64, 64, 40000)])
def test_split_epilogue_equals_convolution_then_split_pass(cuda, rng, cin, cout, n, dtype):
    """fnp_spconv_forward_split / fnp_spconv_forward_tiled_split (the bf16x3 engine's main product: f32 residual and a 16-bit addend
    joined before the ReLU, f32 rows AND their (hi, lo) split written by the convolution's epilogue) against the two launches they
    replace — fnp_spconv_forward with an f32 output and no ReLU, then fnp_split_bf16_add: bit-identical y, hi and lo, on the
    gather kernel and (32 -> 32, 64 -> 64, sorted rows) on the tile rulebook, with and without residual / addend; rows past n stay
    untouched."""
    B, shape = 2, [9, 40, 41] if n <= 5000 else [21, 80, 80]
    feats, idx = _random_sparse(rng, B, shape, n, cin)
    idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    wp = S.pack_weight(torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) * 0.1).astype(np.float32)).to(cuda), dtype)
    x = torch.from_numpy(feats).to(cuda).to(dtype)
    ones = torch.ones(cout, device=cuda)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(cuda)
    res = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(cuda)
    add = torch.from_numpy((rng.standard_normal((n, cout)) * 0.01).astype(np.float32)).to(cuda).to(dtype)
    tiled = cin == cout and cin in (32, 64)
    for residual, addend, relu in ((res, add, True), (None, add, True), (res, None, True), (None, None, False)):
        y0 = S.conv_forward(x, wp, rb, n_dev, out_dtype=torch.float32, scale=ones, shift=sh, residual=residual, relu=False, tile=False)
        if addend is not None:
            if dtype == torch.bfloat16:
                hi0, lo0 = S.split_bf16_add(y0, addend, n_dev, relu=relu)
            else:   # (the split pass is bf16 only: restated here)
                y0[:n] = y0[:n] + addend.float()
                if relu:
                    y0[:n].clamp_(min=0)
                hi0 = y0.to(dtype); lo0 = (y0 - hi0.float()).to(dtype)
        else:
            if relu:
                y0[:n].clamp_(min=0)
            hi0 = y0.to(dtype); lo0 = (y0 - hi0.float()).to(dtype)
        for tile in ((False, True, "sorted") if tiled else (False, "sorted") if (cin, cout) == (128, 128) else (False,)):
            if tile == "sorted":
                if (cin, cout) != (128, 128):
                    continue
                rbs = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3, masks=True)
                S.classsort(rbs, n_dev, 128)     # fnp_spconv_forward_sorted_split: the class-sorted sweep with the same epilogue
                y, hi, lo = S.conv_forward_split(x, wp, rbs, n_dev, scale=ones, shift=sh, residual=residual, addend=addend, relu=relu, ranked=True)
            else:
                y, hi, lo = S.conv_forward_split(x, wp, rb, n_dev, scale=ones, shift=sh, residual=residual, addend=addend, relu=relu, ranked=True, tile=tile)
            tag = (residual is not None, addend is not None, tile)
            assert torch.equal(y[:n], y0[:n]), tag
            assert torch.equal(hi[:n], hi0[:n]) and torch.equal(lo[:n], lo0[:n]), tag
            if dtype == torch.bfloat16:
                assert ((hi[:n].float() + lo[:n].float() - y[:n]).abs() <= y[:n].abs() * 2.0 ** -16).all()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("ksp", [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0)), ((2, 2, 2), (2, 2, 2), (0, 0, 0))])
def test_sparse_inverse_conv_undoes_the_paired_layer_sites(cuda, rng, oracle, ksp, dtype, tol):
    """spconv.SparseInverseConv3d (spconv_backbone.py:16-17 'inverseconv'; the UNet's up-sampling path): it runs on the indice
    pairs of the SparseConv3d that shares its indice_key with the two sides swapped, so its output sites are that layer's INPUT
    sites in their order; values against the oracle's pair-by-pair sum."""
    from findnpropagate_amd import spconv
    k, st, pd = ksp
    B, shape, n, C1, C2 = 2, [9, 20, 22], 1500, 16, 32
    feats, idx = _random_sparse(rng, B, shape, n, C1)
    down = spconv.SparseConv3d(C1, C2, k, stride=st, padding=pd, bias=False, indice_key="sp").to(cuda)
    up = spconv.SparseInverseConv3d(C2, C1, k, indice_key="sp", bias=False).to(cuda)
    x = spconv.SparseConvTensor(torch.from_numpy(feats).to(cuda).to(dtype), torch.from_numpy(idx).to(cuda), shape, B)
    with torch.no_grad():
        y = down(x)
        z = up(y)
    assert z.spatial_shape == shape and torch.equal(z.indices, x.indices) and z.features.shape == (n, C1)
    out_idx, out_shape, pin, pout, pn = oracle.rulebook_strided(idx, shape, k, st, pd)
    yo = oracle.SparseTensor(oracle.conv_apply(x.features.float().cpu().numpy(), down.weight.detach().cpu().numpy(), pin, pout, pn, out_idx.shape[0]),
                             out_idx, out_shape, B)
    # the product's strided output rows come in rank-grid order: compare through the cell keys
    go, wo = np.argsort(_key(y.indices.cpu().numpy(), out_shape)), np.argsort(_key(out_idx, out_shape))
    yf = y.features.float().cpu().numpy()
    assert np.allclose(yf[go], yo.features[wo], rtol=tol, atol=tol * max(1.0, np.abs(yo.features).max()))
    # inverse: feed the oracle the product's own y (row order mapped), so that only the inverse layer is under test
    y_in_oracle_order = np.empty_like(yo.features)
    y_in_oracle_order[wo] = yf[go]
    zo = oracle.inverse_conv(oracle.SparseTensor(y_in_oracle_order, out_idx, out_shape, B), up.weight.detach().cpu().numpy(), (idx, shape, pin, pout, pn))
    zf = z.features.float().cpu().numpy()
    assert np.allclose(zf, zo.features, rtol=tol, atol=tol * max(1.0, np.abs(zo.features).max()))
    with pytest.raises(ValueError, match="indice_key"):
        spconv.SparseInverseConv3d(C2, C1, k, indice_key="other", bias=False).to(cuda)(y)


def _surface_sites(rng, B, shape, n):
    """sites on a two-cell-thick wavy sheet: like a lidar surface after stride-2 layers, most sites have neighbours in only
    one of the two adjacent z planes (what the class sort separates)."""
    D, H, W = shape
    out = []
    for b in range(B):
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        z0 = ((D - 2) * 0.5 * (1 + np.sin(yy / 7.0 + b) * np.cos(xx / 9.0))).astype(np.int64).clip(0, D - 2)
        for dz in (0, 1):
            keep = rng.random((H, W)) < 0.8
            out.append(np.stack([np.full(keep.sum(), b), (z0 + dz)[keep], yy[keep], xx[keep]], 1))
    idx = np.concatenate(out).astype(np.int32)
    idx = idx[rng.permutation(idx.shape[0])[:n]]
    return idx


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,order", [(1, "random"), (17, "random"), (400, "sorted"), (5000, "random"), (5000, "sorted"), (60000, "sorted")])
def test_class_sorted_sweep_equals_plain_sweep(cuda, rng, n, order, dtype):
    """fnp_rulebook_classsort + fnp_spconv_forward_sorted (128 -> 128: rows processed class by class, tiles sweep only the
    offsets their rows use) against fnp_spconv_forward: perm is a permutation of each workgroup range with the z classes
    in order, blockmask is the union of the masks of 16 positions, and the convolution gives the same values."""
    C = 128
    B, shape = 2, [5, 60, 64] if n <= 5000 else [5, 200, 200]
    idx = _surface_sites(rng, B, shape, n)
    n = idx.shape[0]
    if order == "sorted":
        idx = idx[np.lexsort((idx[:, 1], idx[:, 3], idx[:, 2], idx[:, 0]))]
    feats = rng.standard_normal((n, C)).astype(np.float32)
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    wp = S.pack_weight(torch.from_numpy((rng.standard_normal((C, 3, 3, 3, C)) * 0.05).astype(np.float32)).to(cuda), dtype)
    x = torch.from_numpy(feats).to(cuda).to(dtype)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).to(cuda)
    sh = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(cuda)
    res = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)).to(cuda).to(dtype)
    plain = [S.conv_forward(x, wp, rb, n_dev, scale=a, shift=b, residual=r, relu=relu, ranked=True)
             for r, a, b, relu in ((None, sc, sh, True), (res, sc, sh, True), (res, None, None, False))]
    S.classsort(rb, n_dev, C)
    perm, bm = rb._sorted[0][:n].cpu().numpy(), rb._sorted[1].cpu().numpy().view(np.uint32)
    nbr = rb.nbr[:, :n].cpu().numpy()
    masks = ((nbr >= 0).astype(np.uint32) << np.arange(27, dtype=np.uint32)[:, None]).sum(0).astype(np.uint32)
    assert np.array_equal(rb._rowmask[:n].cpu().numpy().view(np.uint32), masks)
    assert np.array_equal(np.sort(perm), np.arange(n)), "perm is a permutation of the rows"
    for blk in range((n + 15) // 16):
        assert bm[blk] == np.bitwise_or.reduce(masks[perm[blk * 16:(blk + 1) * 16]]), blk
    lo, hi = (masks & 0x1ff) != 0, (masks >> 18) != 0
    cls = np.where(lo, np.where(hi, 2, 3), np.where(hi, 1, 0))[perm]
    # inside a workgroup range classes ascend and rows of a class keep their order; a drop marks the next range
    drops = np.nonzero(np.diff(cls) < 0)[0]
    assert len(drops) < 2048
    same = np.diff(cls) == 0
    assert np.all(np.diff(perm)[same] > 0)
    srt = [S.conv_forward(x, wp, rb, n_dev, scale=a, shift=b, residual=r, relu=relu, ranked=True)
           for r, a, b, relu in ((None, sc, sh, True), (res, sc, sh, True), (res, None, None, False))]
    for a, b in zip(plain, srt):
        assert torch.equal(a[:n], b[:n])
    if n >= 5000 and order == "sorted":
        skipped = 1.0 - np.mean([bin(int(v)).count("1") for v in bm[:(n + 15) // 16]]) / 27.0
        assert skipped > 0.2, "a two-cell sheet leaves at least a plane of offsets empty for most blocks"


@pytest.mark.parametrize("C", [16, 32, 64, 128])
@pytest.mark.parametrize("n,order", [(1, "random"), (17, "random"), (400, "sorted"), (5000, "random"), (60000, "sorted")])
def test_f32_class_sorted_sweep_equals_plain_sweep(cuda, rng, n, order, C):
    """fnp_rulebook_classsort_f32 + fnp_spconv_forward_f32_sorted against fnp_spconv_forward on f32 (plain and 4 x 4-transposed
    weights): perm is a permutation that keeps every row inside its workgroup range with the classes in order, and the values
    are the same bits (so the f32 engine stays bit-identical to the CPU oracle)."""
    B, shape = 2, [5, 60, 64] if n <= 5000 else [5, 200, 200]
    idx = _surface_sites(rng, B, shape, n)
    n = idx.shape[0]
    if order == "sorted":
        idx = idx[np.lexsort((idx[:, 1], idx[:, 3], idx[:, 2], idx[:, 0]))]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)
    rb0 = S.rulebook_subm(d_idx, n_dev, grid, 3)
    rb = S.rulebook_subm(d_idx, n_dev, grid, 3, masks=True)
    assert torch.equal(rb0.nbr, rb.nbr)
    w = torch.from_numpy((rng.standard_normal((C, 3, 3, 3, C)) * 0.05).astype(np.float32)).to(cuda)
    x = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)).to(cuda)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).to(cuda)
    sh = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(cuda)
    res = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)).to(cuda)
    cases = ((None, sc, sh, True), (res, sc, sh, True), (res, None, None, False))
    S.classsort_f32(rb, n_dev, C)
    perm = rb._perm_f32[C][:n].cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(n)), "perm is a permutation of the rows"
    masks = rb._rowmask[:n].cpu().numpy().view(np.uint32)
    lo, hi = (masks & 0x1ff) != 0, (masks >> 18) != 0
    cls = np.where(lo, np.where(hi, 2, 3), np.where(hi, 1, 0))[perm]
    same = np.diff(cls) == 0
    assert np.all(np.diff(perm)[same] > 0), "rows of a class keep their order"
    assert np.count_nonzero(np.diff(cls) < 0) < 2048, "classes ascend inside a range"
    for wp in (S.pack_weight(w, torch.float32), S.pack_weight(w, torch.float32, mfma_f32=True)):
        plain = [S.conv_forward(x, wp, rb0, n_dev, scale=a, shift=b, residual=r, relu=relu, ranked=True) for r, a, b, relu in cases]
        srt = [S.conv_forward(x, wp, rb, n_dev, scale=a, shift=b, residual=r, relu=relu, ranked=True) for r, a, b, relu in cases]
        for a, b in zip(plain, srt):
            assert torch.equal(a[:n], b[:n])


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_tile_kernel_in_the_backbone(cuda, mode):
    """Full-size grid, 3 real-shaped scenes: the backbone with the tile-rulebook kernels forced on its ranked 32 -> 32 and
    64 -> 64 layers (rows in rank-grid order: the window carries most neighbours, the overflow rows the rest) equals the
    backbone without them bit for bit, at every stage."""
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    net.fnp_dtype = mode
    from findnpropagate_amd import lib as _lib0
    aborts0 = _lib0.load().fnp_spconv_tiled_aborts()   # (the counter is the library's: test_tile_timeout_is_an_error raises it on purpose)
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    pts, off = syn.make_batch([0, 1, 2])
    pts, off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
    outs = []
    for tile in (False, True):
        S.TILE_MODE = tile
        try:
            with torch.no_grad():
                r = net.forward_points(pts, off, 3, cfg)
            outs.append({k: (r[k].features.clone(), r[k].indices.clone()) for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out")})
        finally:
            S.TILE_MODE = None
    for k in outs[0]:
        assert torch.equal(outs[0][k][1], outs[1][k][1]), k
        assert torch.equal(outs[0][k][0], outs[1][k][0]), k
    from findnpropagate_amd import lib as _lib
    assert _lib.load().fnp_spconv_tiled_aborts() == aborts0   # no hand-over wait of the tiled kernels timed out in these forwards


def test_forward_points_edge_batches(cuda):
    """Ragged and degenerate batches through the fused path: a scene with no points between two scenes, points that
    all fall outside the range, a single point.  A scene's result does not depend on what else is in the batch
    (same coordinates, bit-identical features), whatever rows and tiles its sites land on."""
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    p0 = syn.make_scene(0)

    def run(pts, off, B):
        with torch.no_grad():
            return net.forward_points(torch.from_numpy(pts).to(cuda), torch.from_numpy(np.array(off, np.int32)).to(cuda), B, cfg)

    a = run(p0, [0, len(p0)], 1)
    b = run(np.concatenate([p0, p0]), [0, len(p0), len(p0), 2 * len(p0)], 3)     # scene 1 is empty
    assert b["counts"] == [2 * c for c in a["counts"]]
    key = lambda i: (i[:, 1].astype(np.int64) * 4096 + i[:, 2]) * 4096 + i[:, 3]
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"):
        ia, ib = a[name].indices.cpu().numpy(), b[name].indices.cpu().numpy()
        fa, fb = a[name].features.float().cpu().numpy(), b[name].features.float().cpu().numpy()
        assert (ib[:, 0] != 1).all()
        oa = np.argsort(key(ia))
        for s in (0, 2):
            m = ib[:, 0] == s
            ob = np.argsort(key(ib[m]))
            assert np.array_equal(ia[oa][:, 1:], ib[m][ob][:, 1:]), (name, s)
            assert np.array_equal(fa[oa], fb[m][ob]), (name, s)
    c = run(np.full((100, 5), 1e6, np.float32), [0, 100], 1)
    assert c["counts"] == [0, 0, 0, 0, 0] and c["out"].features.shape == (0, 128)
    d = run(p0[:1], [0, 1], 1)
    assert d["counts"][0] == 1 and d["counts"][4] >= 1


@pytest.mark.parametrize("Cin,Cout,n", [(32, 16, 700), (64, 32, 1300), (128, 64, 900),   # transposed pairs (data gradients)
                                        (48, 48, 500), (16, 64, 300), (8, 24, 100),     # no MFMA instance: VALU kernel
                                        (64, 64, 0), (128, 128, 17), (32, 64, 255), (16, 32, 257)])
def test_conv_bf16_other_shapes_and_sizes(cuda, oracle, rng, Cin, Cout, n):
    """Channel pairs outside the backbone's forward set (the data-gradient instances, shapes that fall back to the
    VALU kernel), an empty tensor, sizes around the tile borders: bf16 in, f32 out against the oracle."""
    B, shape = 2, [7, 24, 25]
    feats, idx = _random_sparse(rng, B, shape, max(n, 1), Cin)
    feats, idx = feats[:n], idx[:n]
    w = (rng.standard_normal((Cout, 3, 3, 3, Cin)) * 0.1).astype(np.float32)
    oracle.round_bf16(feats)
    oracle.round_bf16(w)
    d_idx = torch.from_numpy(np.ascontiguousarray(idx if n else np.zeros((1, 4), np.int32))).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    wp = S.pack_weight(torch.from_numpy(w).to(cuda), torch.bfloat16)
    x = torch.from_numpy(np.ascontiguousarray(feats if n else np.zeros((1, Cin), np.float32))).to(cuda).to(torch.bfloat16)
    got = S.conv_forward(x, wp, rb, n_dev, out_dtype=torch.float32)
    if n == 0:
        return   # nothing to compare: the launch must simply be harmless
    want = oracle.subm_conv(oracle.SparseTensor(feats, idx, shape, B), w).features
    np.testing.assert_allclose(got[:n].cpu().numpy(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("td", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Cin,Cout", [(16, 32), (32, 64), (64, 128)])
@pytest.mark.parametrize("pad", [(1, 1, 1), (0, 1, 1)])
def test_strided_conv_with_in_kernel_rulebook_equals_table_path(cuda, rng, Cin, Cout, pad, td):
    """fnp_spconv_forward_strided computes the rulebook rows of a strided 3x3x3 layer inside the kernel: same output
    coordinates, bit-identical features as fnp_rulebook_strided's table + fnp_spconv_forward — on a grid whose rows
    are ranks and on one that carries a permutation (the voxeliser's grid)."""
    B, shape, n = 2, [11, 60, 61], 9000
    feats, idx = _random_sparse(rng, B, shape, n, Cin)
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)            # keeps the (random) row order: perm != None
    x = torch.from_numpy(feats).to(cuda).to(td)
    wp = S.pack_weight(torch.from_numpy((rng.standard_normal((Cout, 3, 3, 3, Cin)) * 0.1).astype(np.float32)).to(cuda), td)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32)).to(cuda)
    sh = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32)).to(cuda)
    a = S.rulebook_strided(d_idx, n_dev, grid, 3, 2, pad, 4 * n)
    ya = S.conv_forward(x, wp, a, a.out_n, scale=sc, shift=sh, relu=True)
    b = S.rulebook_strided(d_idx, n_dev, grid, 3, 2, pad, 4 * n, want_nbr=False)
    yb = S.conv_forward_strided(x, wp, b, scale=sc, shift=sh, relu=True)
    m = int(a.out_n.item())
    assert m == int(b.out_n.item()) and m > n // 4
    assert torch.equal(a.out_indices[:m], b.out_indices[:m])
    assert torch.equal(ya[:m], yb[:m])


def test_fused_backbone_fp16_and_autocast(cuda, oracle, rng):
    """FNP_DTYPE fp16 (the reference's AMP mode: fp16 features and weights, fp32 accumulate): the fused engine against
    the f32 oracle (fp16 storage keeps 11 bits: 10x closer than bf16), and the module path under torch autocast takes
    the fp16 kernels by itself (train_utils.py:172)."""
    from findnpropagate_amd import spconv
    net = _small_net(cuda, "fp16")
    shape = net.sparse_shape
    feats, idx = _random_sparse(rng, 2, shape, 6000, 5)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want32 = oracle.backbone_forward(sd, feats, idx, 2, shape)
    bd = {"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda), "batch_size": 2}
    net.fnp_out_dtype = "native"
    with torch.no_grad():
        out = net(dict(bd))
    assert out["encoded_spconv_tensor"].features.dtype == torch.float16
    g, _ = _by_coord(out["encoded_spconv_tensor"].features.float().cpu().numpy(), out["encoded_spconv_tensor"].indices.cpu().numpy(), want32["out"].spatial_shape)
    w, _ = _by_coord(want32["out"].features, want32["out"].indices, want32["out"].spatial_shape)
    rel16 = np.abs(g - w).max() / (np.abs(w).max() + 1e-6)
    assert rel16 < 2e-2, rel16
    # autocast: an f32 SubMConv3d module runs its features in fp16 and returns fp16
    conv = spconv.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key="k").to(cuda)
    f16, i16 = _random_sparse(rng, 2, [9, 30, 31], 1500, 16)
    x = spconv.SparseConvTensor(torch.from_numpy(f16).to(cuda), torch.from_numpy(i16).to(cuda), [9, 30, 31], 2)
    with torch.no_grad():
        y32 = conv(x).features
        with torch.autocast("cuda", dtype=torch.float16):
            y16 = conv(x).features
    assert y32.dtype == torch.float32 and y16.dtype == torch.float16
    np.testing.assert_allclose(y16.float().cpu().numpy(), y32.cpu().numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_index_chain_on_a_second_stream_changes_nothing(cuda, mode, monkeypatch):
    """The fused engine with its index chain (rank grids, rulebooks, records, class sort) on a side stream and the convolutions
    waiting on one event per stage — what a captured hipGraph runs by default — forced for plain stream launches too, against
    the one-stream engine: the same rows at every stage, bit for bit, twice in a row (the second forward starts from the grids
    and pool counters the first one left)."""
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    from findnpropagate_amd.backbones_3d import spconv_backbone as BB
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    net.fnp_dtype = mode
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    pts, off = syn.make_batch([0, 1, 2])
    pts, off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
    outs = []
    with torch.no_grad():
        for two in (False, True, True):
            monkeypatch.setattr(BB, "TWO_STREAMS", two)
            r = net.forward_points(pts, off, 3, cfg)
            torch.cuda.synchronize()
            outs.append([(r[k].features.clone(), r[k].indices.clone()) for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out")])
    for other in outs[1:]:
        for (fa, ia), (fb, ib) in zip(outs[0], other):
            assert torch.equal(ia, ib) and torch.equal(fa, fb)


def test_probe_graphs_equal_eager_and_time_the_dominant_launches(cuda):
    """forward_points_graphed(probe=True): the forward captured as two graphs with the four 128 -> 128 launches of stage 4 issued
    as plain launches between them (bench.py's timed step).  Same rows as the eager forward at every stage, twice in a row
    and for another frame through the same graphs; with a profile list set, every step appends four (tag, start, end) event
    pairs whose tags are the stage-4 SubM layers'."""
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    eng = net.engine()
    keys = ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out")
    with torch.no_grad():
        for seeds in ([0, 1, 2], [3, 4, 5], [0, 1, 2]):
            pts, off = syn.make_batch(seeds)
            pts, off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
            want = net.forward_points(pts, off, 3, cfg)
            want = [(want[k].features.clone(), want[k].indices.clone()) for k in keys]
            eng.profile = []
            try:
                got = net.forward_points_graphed(pts, off, 3, cfg, capacity=131072, probe=True)
                torch.cuda.synchronize()
                prof = eng.profile
            finally:
                eng.profile = None
            for (wf, wi), k in zip(want, keys):
                assert torch.equal(got[k].indices, wi) and torch.equal(got[k].features, wf), k
            assert [t[0] for t in prof] == [(128, 128, 27, False, True), (128, 128, 27, True, True)] * 2
            assert all(e0.elapsed_time(e1) > 0.0 for _, e0, e1 in prof)
    assert sum(1 for k in eng._graphs if k[-1] is True) == 1, "one pair of graphs served the three calls"


def test_forwards_that_return_early_are_safe_across_streams(cuda):
    """A forward returns once its counts are on the host, with the GPU still convolving.  Calling the engine again from
    ANOTHER stream right away (eager and replayed) must wait for the forward before it: alternating streams and frames, every
    result equals the one computed with a device synchronisation after each call."""
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    frames = []
    for seeds in ([0, 1], [2, 3], [4, 5]):
        pts, off = syn.make_batch(seeds)
        frames.append((torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)))
    keys = ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out")
    with torch.no_grad():
        want = []
        for pts, off in frames:
            r = net.forward_points(pts, off, 2, cfg)
            torch.cuda.synchronize()
            want.append([(r[k].features.clone(), r[k].indices.clone()) for k in keys])
        streams = [torch.cuda.Stream(cuda), torch.cuda.Stream(cuda)]
        torch.cuda.synchronize()
        # eager: six forwards issued back to back on alternating streams, nothing synchronised in between (each call's outputs
        # are its own tensors)
        got = []
        for i in range(6):
            pts, off = frames[i % 3]
            with torch.cuda.stream(streams[i % 2]):
                r = net.forward_points(pts, off, 2, cfg)
                got.append([(r[k].features, r[k].indices) for k in keys])
        torch.cuda.synchronize()
        for i, g in enumerate(got):
            for (gf, gi), (wf, wi) in zip(g, want[i % 3]):
                assert torch.equal(gi, wi) and torch.equal(gf, wf), i
        # replayed: the outputs are views of the graphs' buffers (valid until the next call), so each call's copies are taken on
        # its stream and the next stream waits for them; the replay itself still has to wait for the previous replay's end
        for i in range(6):
            pts, off = frames[i % 3]
            streams[i % 2].wait_stream(streams[(i + 1) % 2])
            with torch.cuda.stream(streams[i % 2]):
                r = net.forward_points_graphed(pts, off, 2, cfg, capacity=131072)
                g = [(r[k].features.clone(), r[k].indices.clone()) for k in keys]
            streams[i % 2].synchronize()
            for (gf, gi), (wf, wi) in zip(g, want[i % 3]):
                assert torch.equal(gi, wi) and torch.equal(gf, wf), i
