"""Compact rulebook (fnp_rulebook_ell) + the VALU convolution on it (fnp_spconv_forward_ell) for the sparse-neighbourhood
layers (conv_input, the 16 -> 16 SubM layers, the strided 16 -> 32 layer) against the (27, cap) table and the kernels on it."""
import numpy as np
import pytest
import torch

from findnpropagate_amd import sparse as S

pytestmark = pytest.mark.gpu
EMPTY, LINK = 0xFFFFFFFF, 31


def _sites(rng, B, shape, n, blob):
    """random sites; blob: plus a solid 6 x 6 x 6 cube (rows with 27 neighbours: three extension records)"""
    cells = B * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=n, replace=False)
    b, rem = np.divmod(lin, shape[0] * shape[1] * shape[2]); z, rem = np.divmod(rem, shape[1] * shape[2]); y, x = np.divmod(rem, shape[2])
    idx = np.stack([b, z, y, x], 1)
    if blob:
        zz, yy, xx = np.meshgrid(np.arange(1, 7), np.arange(3, 9), np.arange(2, 8), indexing="ij")
        idx = np.concatenate([idx, np.stack([np.zeros(zz.size, np.int64), zz.ravel(), yy.ravel(), xx.ravel()], 1)])
        idx = np.unique(idx, axis=0)
        idx = idx[rng.permutation(idx.shape[0])]
    return idx.astype(np.int32)


def _decode(rec, cap, n):
    """records -> list per row of (k, in_row), following links"""
    rec = rec.cpu().numpy().view(np.uint32).reshape(-1, 8)
    out, used = [], set()
    for r in range(n):
        row, cur = [], r
        while True:
            e = rec[cur]
            link = (e[7] >> 27) == LINK and e[7] != EMPTY
            for j in range(7 if link else 8):
                if e[j] != EMPTY:
                    row.append((int(e[j] >> 27), int(e[j] & ((1 << 27) - 1))))
            if not link:
                break
            ext = int(e[7] & ((1 << 27) - 1))
            assert ext not in used, "an extension record belongs to one row"
            used.add(ext)
            cur = cap + ext
        out.append(row)
    return out, used


def _n_records(cnt):
    m = 1
    while cnt > 8:
        cnt -= 7
        m += 1
    return m


@pytest.mark.parametrize("n,blob", [(1, False), (300, True), (5000, True), (20000, False)])
def test_subm_records_restate_the_table(cuda, rng, n, blob):
    B, shape = 2, [9, 40, 41]
    idx = _sites(rng, B, shape, n, blob)
    n = idx.shape[0]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)
    table = S.rulebook_subm(d_idx, n_dev, grid, 3).nbr[:, :n].cpu().numpy()
    rb = S.rulebook_subm_ell(d_idx, n_dev, grid, pool_records=n * 3 + 8, with_table=True)
    assert torch.equal(rb.nbr[:, :n].cpu(), torch.from_numpy(table)), "the table written beside the records is fnp_rulebook_subm's"
    rows, used = _decode(rb._ell[0], rb.cap_out, n)
    want_ext = 0
    for r in range(n):
        ks = np.nonzero(table[:, r] >= 0)[0]
        assert rows[r] == [(int(k), int(table[k, r])) for k in ks], r      # ascending k, the table's rows
        want_ext += _n_records(len(ks)) - 1
    assert int(rb._ell[2].item()) == want_ext == len(used)
    # the chunk orders: a permutation of every 256-row chunk, entry counts ascending
    perm = rb._ell[0].view(torch.uint8)[(rb.cap_out + rb._ell[1]) * 32:].cpu().numpy()
    counts = np.array([len(r) for r in rows] + [0] * (-n % 256))
    for c0 in range(0, n, 256):
        pc = perm[c0:c0 + 256].astype(np.int64)
        assert np.array_equal(np.sort(pc), np.arange(256))
        assert np.all(np.diff(counts[c0 + pc]) >= 0)
    if blob:
        assert max(len(r) for r in rows) == 27 and want_ext > 0


def test_strided_records_and_pool_overflow(cuda, rng):
    B, shape = 2, [11, 30, 32]
    idx = _sites(rng, B, shape, 6000, True)
    n = idx.shape[0]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)
    full = S.rulebook_strided(d_idx, n_dev, grid, 3, 2, 1, cap_out=n * 8)
    n_out = int(full.out_n.item())
    table = full.nbr[:, :n_out].cpu().numpy()
    lean = S.rulebook_strided(d_idx, n_dev, grid, 3, 2, 1, cap_out=n * 8, want_nbr=False)
    assert torch.equal(lean.out_indices[:n_out], full.out_indices[:n_out])
    S.ell_for_strided(lean, pool_records=n_out * 3)
    rows, used = _decode(lean._ell[0], lean.cap_out, n_out)
    for r in range(n_out):
        ks = np.nonzero(table[:, r] >= 0)[0]
        assert rows[r] == [(int(k), int(table[k, r])) for k in ks], r
    need = int(lean._ell[2].item())
    assert need == len(used) > 0
    # a pool that is too small: the counter still says what was needed, nothing is written beyond the records
    small = S.rulebook_strided(d_idx, n_dev, grid, 3, 2, 1, cap_out=n * 8, want_nbr=False)
    S.ell_for_strided(small, pool_records=max(need // 2, 1))
    assert int(small._ell[2].item()) == need > small._ell[1]
    assert small._ell[0].numel() == (small.cap_out + small._ell[1]) * 8 + ((small.cap_out + 255) // 256 * 256) // 4   # records + chunk orders


def _conv_inputs(rng, cuda, cin, cout, dtype, n=7000, blob=True):
    B, shape = 2, [9, 40, 41]
    idx = _sites(rng, B, shape, n, blob)
    n = idx.shape[0]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(cuda).to(dtype)
    w = S.pack_weight(torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) * 0.2).astype(np.float32)).to(cuda), dtype)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(cuda)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(cuda)
    return d_idx, n_dev, n, grid, x, w, sc, sh


@pytest.mark.parametrize("cin", [5, 4])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_first_layer_on_records_is_the_table_kernel_bit_for_bit(cuda, rng, cin, out_dtype):
    """f32 point features: the same k-ascending, cin-ascending fmaf chain as spconv_first_kernel (= the oracle's), same bits."""
    d_idx, n_dev, n, grid, x, w, sc, sh = _conv_inputs(rng, cuda, cin, 16, torch.float32)
    table = S.rulebook_subm(d_idx, n_dev, grid, 3)
    ell = S.rulebook_subm_ell(d_idx, n_dev, grid, pool_records=n)
    res = torch.from_numpy(rng.standard_normal((n, 16)).astype(np.float32)).to(cuda).to(out_dtype)
    for residual, scale, shift, relu in ((None, sc, sh, True), (res, sc, sh, True), (None, None, None, False)):
        a = S.conv_forward(x, w, table, n_dev, out_dtype=out_dtype, scale=scale, shift=shift, residual=residual, relu=relu)
        b = S.conv_forward_ell(x, w, ell, n_dev, out_dtype=out_dtype, scale=scale, shift=shift, residual=residual, relu=relu)
        assert torch.equal(a[:n], b[:n])


@pytest.mark.parametrize("cout", [16, 32])
@pytest.mark.parametrize("dtype,ulp", [(torch.bfloat16, 2.0 ** -8), (torch.float16, 2.0 ** -11)])
def test_16_channel_layers_on_records_match_the_table_kernels(cuda, rng, cout, dtype, ulp):
    """16-bit rows: exact products, f32 accumulation in another order than the MFMA kernel (dot2 pairs, neighbours that exist
    only): against the f32 sum of the same rounded operands within 1e-4 of the row scale + one rounding of the stored value."""
    d_idx, n_dev, n, grid, x, w, sc, sh = _conv_inputs(rng, cuda, 16, cout, dtype)
    table = S.rulebook_subm(d_idx, n_dev, grid, 3)
    ell = S.rulebook_subm_ell(d_idx, n_dev, grid, pool_records=n)
    res = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(cuda).to(dtype)
    want = S.conv_forward(x.float(), w.float(), table, n_dev, scale=sc, shift=sh, residual=res.float(), relu=True, valu=True)[:n]
    got = S.conv_forward_ell(x, w, ell, n_dev, scale=sc, shift=sh, residual=res, relu=True, mfma=False)[:n]
    mfma = S.conv_forward(x, w, table, n_dev, scale=sc, shift=sh, residual=res, relu=True)[:n]
    # the MATRIX kernel on the records (entries expanded per tile from the 32-byte records, chains included: the blob's rows
    # hold 27 neighbours): the table kernel's values bit for bit, with and without residual / BatchNorm / ReLU
    for residual, scale, shift, relu in ((res, sc, sh, True), (None, sc, sh, True), (None, None, None, False)):
        a = S.conv_forward(x, w, table, n_dev, scale=scale, shift=shift, residual=residual, relu=relu)
        b = S.conv_forward_ell(x, w, ell, n_dev, scale=scale, shift=shift, residual=residual, relu=relu, mfma=True)
        assert torch.equal(a[:n], b[:n])
    scale = float(want.abs().max())
    assert float((got.float() - want).abs().max()) <= (1e-4 + ulp) * scale
    assert float((mfma.float() - want).abs().max()) <= (1e-4 + ulp) * scale          # (the same bar holds for the table kernel)
    assert float((got.float() != mfma.float()).float().mean()) < 0.05                # and the two agree on all but a few roundings


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n", [1, 17, 300, 40000])
def test_strided_16_to_32_on_records_is_the_table_kernel_bit_for_bit(cuda, rng, n, dtype):
    """the strided 16 -> 32 layer: records of the OUTPUT rows (ell_for_strided) through the matrix kernel == the (27, cap) table
    through the matrix kernel, for sizes from one row (a partial block) to many tiles."""
    B, shape = 2, [11, 60, 64]
    idx = _sites(rng, B, shape, n, n >= 300)
    n_in = idx.shape[0]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n_in, cuda)
    grid = S.build_grid(d_idx, n_dev, B, shape)
    full = S.rulebook_strided(d_idx, n_dev, grid, 3, 2, 1, cap_out=n_in * 8 + 64)
    lean = S.rulebook_strided(d_idx, n_dev, grid, 3, 2, 1, cap_out=n_in * 8 + 64, want_nbr=False)
    n_out = int(full.out_n.item())
    S.ell_for_strided(lean, pool_records=n_out * 3 + 8)
    x = torch.from_numpy(rng.standard_normal((n_in, 16)).astype(np.float32)).to(cuda).to(dtype)
    w = S.pack_weight(torch.from_numpy((rng.standard_normal((32, 3, 3, 3, 16)) * 0.2).astype(np.float32)).to(cuda), dtype)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, 32).astype(np.float32)).to(cuda)
    sh = torch.from_numpy(rng.standard_normal(32).astype(np.float32)).to(cuda)
    a = S.conv_forward(x, w, full, full.out_n, scale=sc, shift=sh, relu=True)
    b = S.conv_forward_ell(x, w, lean, lean.out_n, scale=sc, shift=sh, relu=True, mfma=True)
    assert torch.equal(a[:n_out], b[:n_out])


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_backbone_on_compact_rulebooks_equals_backbone_on_tables(cuda, mode):
    """Full-size grid, three synthetic scenes: the fused backbone with stage 1 and the 16 -> 32 layer on the compact rulebook
    (default) against the same backbone on the (27, cap) tables: same site sets, features equal up to a few 16-bit roundings."""
    from findnpropagate_amd import synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(cuda).eval()
    net.fnp_dtype = mode
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    pts, off = syn.make_batch([0, 1, 2])
    pts, off = torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda)
    outs = []
    for ell in (False, True):   # (True: every sparse-neighbourhood layer on records; the default — conv_input only — lies between)
        S.ELL_MODE = ell
        try:
            with torch.no_grad():
                r = net.forward_points(pts, off, 3, cfg)
            outs.append({k: (r[k].features.float().clone(), r[k].indices.clone()) for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4", "out")})
        finally:
            S.ELL_MODE = None
    for k in outs[0]:   # the 16-channel layers run the matrix kernel on either rulebook: same values
        assert torch.equal(outs[0][k][1], outs[1][k][1]), k
        assert torch.equal(outs[0][k][0], outs[1][k][0]), k
    # ... and with the VALU kernel on the records (FNP_ELL_MFMA=0): another f32 summation order in five layers, a few roundings
    S.ELL_MODE, S.ELL_MFMA = True, False
    try:
        with torch.no_grad():
            r = net.forward_points(pts, off, 3, cfg)
        valu = {k: r[k].features.float().clone() for k in outs[0]}
    finally:
        S.ELL_MODE, S.ELL_MFMA = None, True
    ulp = 2.0 ** -8 if mode == "bf16" else 2.0 ** -11
    for k in outs[0]:
        a, b = outs[0][k][0], valu[k]
        assert float((a - b).abs().max()) <= 8 * ulp * max(1.0, float(a.abs().max())), k
        assert float((a != b).float().mean()) < 0.6, k
