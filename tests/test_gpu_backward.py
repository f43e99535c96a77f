"""Backward of the sparse convolution on the MI355X: rulebook transpose + forward kernel
(dgrad) and the two-stage weight gradient, through torch.autograd, against the oracle's conv_backward
(itself pinned to torch's dense conv3d autograd in tests/test_oracle_spconv.py).  Tolerances: f32 1e-4
relative to the gradient scale (summation order differs); bf16 / fp16 (the reference's AMP mode) against the same f32
oracle fed the rounded inputs: 3e-2 / 4e-3 (one rounding of the stored data gradient: 2^-8 / 2^-11 relative)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _random_sparse(rng, B, shape, n, C):
    cells = B * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=min(n, cells), replace=False)
    b, rem = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    return rng.standard_normal((len(lin), C)).astype(np.float32), np.stack([b, z, y, x], 1).astype(np.int32)


def _close(got, want, tol):
    scale = max(float(np.abs(want).max()), 1e-6)
    assert np.abs(got - want).max() <= tol * scale, (np.abs(got - want).max(), scale)


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("bf16", 3e-2), ("fp16", 4e-3)])
@pytest.mark.parametrize("mode,cin,cout,k,s,p", [("subm", 16, 16, 3, 1, 1), ("subm", 32, 32, 3, 1, 1), ("subm", 5, 16, 3, 1, 1),
                                                  ("strided", 16, 32, 3, 2, 1), ("strided", 64, 128, 3, 2, (0, 1, 1)),
                                                  ("strided", 128, 128, (3, 1, 1), (2, 1, 1), 0), ("subm", 64, 64, 3, 1, 1)])
def test_conv_autograd_matches_oracle(cuda, oracle, rng, dtype, tol, mode, cin, cout, k, s, p):
    from findnpropagate_amd import spconv
    B, shape, n = 2, [9, 20, 22], 1500
    feats, idx = _random_sparse(rng, B, shape, n, cin)
    kk = [k] * 3 if np.isscalar(k) else list(k)
    ss = [s] * 3 if np.isscalar(s) else list(s)
    pp = [p] * 3 if np.isscalar(p) else list(p)
    td = {"f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    conv = (spconv.SubMConv3d(cin, cout, kk, padding=[q // 2 for q in kk], bias=True, indice_key="a") if mode == "subm"
            else spconv.SparseConv3d(cin, cout, kk, stride=ss, padding=pp, bias=False)).to(cuda)
    w = conv.weight.detach().cpu().numpy()
    if dtype != "f32":   # the oracle sees the values the kernel sees
        feats = torch.from_numpy(feats).to(td).float().numpy()
        w = torch.from_numpy(w).to(td).float().numpy()
    x = torch.from_numpy(feats).to(cuda).to(td).requires_grad_(True)
    out = conv(spconv.SparseConvTensor(x, torch.from_numpy(idx).to(cuda), shape, B))
    oi = out.indices.cpu().numpy()
    dy = rng.standard_normal((oi.shape[0], cout)).astype(np.float32)
    if dtype != "f32":
        dy = torch.from_numpy(dy).to(td).float().numpy()
    (out.features.float() * torch.from_numpy(dy).to(cuda)).sum().backward()

    if mode == "subm":
        pin, pout, pnum = oracle.rulebook_subm(idx, shape, kk)
        order = np.arange(idx.shape[0])
    else:
        o_idx, o_shape, pin, pout, pnum = oracle.rulebook_strided(idx, shape, kk, ss, pp)
        key = lambda a: ((a[:, 0].astype(np.int64) * o_shape[0] + a[:, 1]) * o_shape[1] + a[:, 2]) * o_shape[2] + a[:, 3]
        assert np.array_equal(np.sort(key(oi)), np.sort(key(o_idx)))
        lut = {int(v): i for i, v in enumerate(key(oi))}
        order = np.array([lut[int(v)] for v in key(o_idx)])      # oracle row -> our row
    dx_w, dw_w = oracle.conv_backward(feats, w, pin, pout, pnum, dy[order])
    assert x.grad is not None and x.grad.dtype == td and conv.weight.grad.shape == conv.weight.shape
    _close(x.grad.float().cpu().numpy(), dx_w, tol)
    _close(conv.weight.grad.float().cpu().numpy(), dw_w, tol)
    if conv.bias is not None:
        _close(conv.bias.grad.float().cpu().numpy(), dy.sum(0), max(tol, 1e-3))
    # bit-reproducible: a second backward gives identical gradients (no atomics anywhere)
    g1 = conv.weight.grad.clone()
    conv.weight.grad = None
    x.grad = None
    out2 = conv(spconv.SparseConvTensor(x, torch.from_numpy(idx).to(cuda), shape, B))
    (out2.features.float() * torch.from_numpy(dy).to(cuda)).sum().backward()
    assert torch.equal(conv.weight.grad, g1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,n", [(16, 16, 1), (32, 32, 700), (16, 32, 5000), (64, 64, 5000), (128, 128, 9000), (64, 128, 3000)])
def test_wgrad_on_pair_lists_equals_the_table_sweep(cuda, rng, dtype, cin, cout, n):
    """fnp_rulebook_pairs + fnp_spconv_wgrad_pairs: the pair lists name, per offset and in ascending order, exactly the output
    rows with a neighbour there and those neighbours; the weight gradient summed over them equals the sweep over the whole
    table up to f32 rounding of another grouping (1e-5 of the gradient's scale), and twice the same bits."""
    from findnpropagate_amd import sparse as S
    B, shape = 2, [7, 40, 41]
    cells = B * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=n, replace=False)
    b, rem = np.divmod(lin, shape[0] * shape[1] * shape[2]); z, rem = np.divmod(rem, shape[1] * shape[2]); y, x = np.divmod(rem, shape[2])
    idx = torch.from_numpy(np.stack([b, z, y, x], 1).astype(np.int32)).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(idx, n_dev, S.build_grid(idx, n_dev, B, shape), 3)
    po, pi, cnt = S.rulebook_pairs(rb, n_dev)
    nbr = rb.nbr[:, :n].cpu().numpy()
    for k in range(27):
        rows = np.nonzero(nbr[k] >= 0)[0]
        assert int(cnt[k]) == len(rows)
        assert np.array_equal(po[k, :len(rows)].cpu().numpy(), rows) and np.array_equal(pi[k, :len(rows)].cpu().numpy(), nbr[k][rows])
    xin = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(cuda).to(dtype)
    dy = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(cuda).to(dtype)
    a = S.conv_wgrad(xin, dy, rb, n_dev, cin, cout, pairs=False)
    b1 = S.conv_wgrad(xin, dy, rb, n_dev, cin, cout, pairs=True)
    b2 = S.conv_wgrad(xin, dy, rb, n_dev, cin, cout, pairs=True)
    assert torch.equal(b1, b2)
    assert float((a - b1).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-6)


def test_backbone_training_step_runs_and_matches_torch_reference_ops(cuda, rng):
    """VoxelResBackBone8x in train mode (BatchNorm with batch statistics), f32: every parameter gradient
    equals the one obtained when the convolutions are evaluated with plain torch ops (index_select + matmul
    over the same rulebooks, torch's own autograd) — 'plain PyTorch fp32 reference of the same op'."""
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    from findnpropagate_amd.spconv import conv as C
    from findnpropagate_amd import synthetic as syn, sparse as S
    grid = np.array([96, 88, 40])
    net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": "fp32"}, 5, grid), 0).to(cuda).train()
    feats, idx = _random_sparse(rng, 2, net.sparse_shape, 5000, 5)
    bd = lambda: {"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda).float(), "batch_size": 2}

    def run():
        net.zero_grad()
        out = net(bd())
        loss = sum((t.features.float() ** 2).mean() for t in list(out["multi_scale_3d_features"].values()) + [out["encoded_spconv_tensor"]])
        loss.backward()
        return float(loss.detach()), {k: v.grad.clone() for k, v in net.named_parameters()}

    loss_a, grads_a = run()

    class RefFn:   # same signature as SparseConvFunction.apply, torch ops only
        @staticmethod
        def apply(f, weight, rb, n_out_dev, n_in_dev, ranked=False, rows=None, prepacked=None):
            n_out = int(n_out_dev.item())
            Cout, Cin = weight.shape[0], weight.shape[-1]
            wk = weight.reshape(Cout, rb.K, Cin)
            fz = torch.cat([f.float(), f.new_zeros((1, Cin), dtype=torch.float32)], 0)
            out = f.new_zeros((rb.cap_out if rows is None else rows, Cout), dtype=torch.float32)
            for k in range(rb.K):
                nb = rb.nbr[k, :n_out].long()
                nb = torch.where(nb < 0, torch.full_like(nb, f.shape[0]), nb)
                out[:n_out] = out[:n_out] + fz[nb] @ wk[:, k, :].t()
            return out.to(f.dtype)

    real = C.SparseConvFunction
    C.SparseConvFunction = RefFn
    try:
        loss_b, grads_b = run()
    finally:
        C.SparseConvFunction = real
    # ... and when, on top of that, BatchNorm / ReLU / residual run as the reference's separate torch modules
    from findnpropagate_amd.spconv import norm as N
    N.ENABLED = False
    C.SparseConvFunction = RefFn
    try:
        loss_c, grads_c = run()
    finally:
        C.SparseConvFunction = real
        N.ENABLED = True
    assert set(grads_a) == set(grads_b) == set(grads_c) and len(grads_a) > 60
    # gradients pass through 21 ReLUs and 21 batch normalisations: a summation-order difference of 1e-7 in a
    # pre-activation next to zero flips its mask, so the comparison is relative to each tensor's scale, 5e-3
    for tag, loss_x, grads_x in (("torch convs", loss_b, grads_b), ("torch convs + torch BN modules", loss_c, grads_c)):
        assert abs(loss_a - loss_x) <= 1e-4 * abs(loss_x), tag
        for name in grads_a:
            ga, gb = grads_a[name].float().cpu().numpy(), grads_x[name].float().cpu().numpy()
            assert np.isfinite(ga).all(), name
            assert np.abs(ga - gb).max() <= 5e-3 * max(np.abs(gb).max(), 1e-6), (tag, name, np.abs(ga - gb).max(), np.abs(gb).max())
            if ga.size >= 1000:     # (whole weight tensors: the typical element is far closer than the worst one)
                assert np.abs(ga - gb).mean() <= 3e-4 * max(np.abs(gb).max(), 1e-6), (tag, name)


def test_ddp_wraps_the_backbone(cuda, rng):
    """cfg 5 runs the detector under DistributedDataParallel (tools/train_st.py:245): the backbone's parameters are
    ordinary nn.Parameters behind a custom autograd Function, so DDP (RCCL all-reduce of the bucketed gradients;
    world size 1 here) must wrap it unchanged and give the same gradients."""
    import socket
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    from findnpropagate_amd import synthetic as syn
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=cuda)
    try:
        grid = np.array([96, 88, 40])
        net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": "bf16"}, 5, grid), 0).to(cuda).train()
        feats, idx = _random_sparse(rng, 2, net.sparse_shape, 4000, 5)
        bd = lambda: {"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda).float(), "batch_size": 2}
        loss_of = lambda out: sum((t.features.float() ** 2).mean() for t in out["multi_scale_3d_features"].values())
        net.zero_grad()
        loss_of(net(bd())).backward()
        want = {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}
        ddp = DDP(net, device_ids=[cuda.index])
        net.zero_grad()
        loss_of(ddp(bd())).backward()
        got = {k: v.grad for k, v in net.named_parameters() if v.grad is not None}
        assert set(got) == set(want) and len(got) > 50
        for k in want:
            assert torch.equal(got[k], want[k]), k          # deterministic kernels + world size 1: identical
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 2e-2), ("fp16", 3e-3)])
@pytest.mark.parametrize("C,n,with_res", [(16, 2, False), (32, 777, True), (64, 20000, True), (128, 5001, False),
                                          (8, 100, False), (256, 3001, True), (16, 70001, True)])   # (8 / 256: one / 32 threads per row in the statistics pass)
def test_fused_bn_relu_residual_matches_torch(cuda, rng, dtype, tol, C, n, with_res):
    """spconv/norm.bn_act (csrc/bnorm.hip) == nn.BatchNorm1d(train) -> (+ residual) -> ReLU evaluated by torch in f32 on
    the same stored values: output, input / residual / affine gradients, running statistics, num_batches_tracked;
    rows beyond the device-side row count are ignored; two runs are bit-identical."""
    from findnpropagate_amd.spconv import norm as N
    from findnpropagate_amd import sparse as S
    td = {"f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    cap = n + 37
    x0 = torch.from_numpy((rng.standard_normal((cap, C)) * 1.7 + rng.standard_normal(C)).astype(np.float32)).to(cuda).to(td)
    r0 = torch.from_numpy(rng.standard_normal((cap, C)).astype(np.float32)).to(cuda).to(td) if with_res else None
    dy = torch.from_numpy(rng.standard_normal((cap, C)).astype(np.float32)).to(cuda).to(td)
    bn_a = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(cuda).train()
    with torch.no_grad():
        bn_a.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)))
        bn_a.bias.copy_(torch.from_numpy(rng.standard_normal(C).astype(np.float32)))
    import copy
    bn_b = copy.deepcopy(bn_a)
    n_dev = S.device_scalar(n, cuda)

    def fused(bn):
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if with_res else None
        y = N.bn_act(x, n_dev, bn, residual=r, relu=True)
        (y[:n].float() * dy[:n].float()).sum().backward()
        return y.detach(), x.grad, None if r is None else r.grad

    y_a, dx_a, dr_a = fused(bn_a)
    x = x0[:n].float().clone().requires_grad_(True)
    r = r0[:n].float().clone().requires_grad_(True) if with_res else None
    z_t = bn_b(x)
    z_t = z_t + r if with_res else z_t
    y_t = torch.relu(z_t)
    (y_t * dy[:n].float()).sum().backward()
    scale = lambda t: max(float(t.abs().max()), 1e-6)
    assert (y_a[:n].float() - y_t.detach()).abs().max() <= tol * scale(y_t)
    # a pre-activation within rounding of zero may take the other side of the ReLU: those elements (a handful) carry a
    # whole dy instead of none; everything else must agree
    sure = z_t.detach().abs() > 1e-4 * scale(y_t)      # (the mask is decided on f32 values computed from the same stored inputs)
    assert float((~sure).float().mean()) < 0.01
    assert ((dx_a[:n].float() - x.grad).abs() * sure).max() <= tol * scale(x.grad) + 2.0 * float((~sure).sum()) / n * scale(dy.float())
    if with_res:
        assert ((dr_a[:n].float() - r.grad).abs() * sure).max() <= tol * scale(r.grad)
    # (an element whose ReLU fell on the other side carries its whole dy, and dy * xhat, into the affine gradients of its channel:
    #  with 70 k rows one such element is likely; the allowance is exactly those elements' contribution)
    flip = ((y_a[:n].float() > 0) != (y_t.detach() > 0)) & ~sure
    xd = x.detach()
    xhat = (xd - xd.mean(0)) / torch.sqrt(xd.var(0, unbiased=False) + bn_b.eps)
    slack_b = (flip * dy[:n].float().abs()).sum(0) * 1.01
    slack_g = (flip * (dy[:n].float() * xhat).abs()).sum(0) * 1.01
    assert ((bn_a.weight.grad - bn_b.weight.grad).abs() <= max(tol, 1e-4) * scale(bn_b.weight.grad) + slack_g).all()
    assert ((bn_a.bias.grad - bn_b.bias.grad).abs() <= max(tol, 1e-4) * scale(bn_b.bias.grad) + slack_b).all()
    assert int(flip.sum()) <= 3
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1
    bn_c = copy.deepcopy(bn_b)
    bn_c.load_state_dict(bn_a.state_dict())
    bn_d = copy.deepcopy(bn_c)
    y1, dx1, _ = fused(bn_c)
    y2, dx2, _ = fused(bn_d)
    assert torch.equal(y1[:n], y2[:n]) and torch.equal(dx1[:n], dx2[:n]) and torch.equal(bn_c.weight.grad, bn_d.weight.grad)


@pytest.mark.parametrize("mode,cin,cout,s,p", [("subm", 16, 16, 1, 1), ("strided", 16, 32, 2, 1)])
def test_conv_autograd_full_size_grid(cuda, oracle, mode, cin, cout, s, p):
    """The same gradients at BASELINE size: two synthetic 30k-point scenes voxelised on the 41 x 1440 x 1440 grid
    (~38k voxels), bf16, against the oracle's conv_backward on the voxel coordinates."""
    from findnpropagate_amd import spconv, sparse as S, synthetic as syn
    rng = np.random.default_rng(3)
    pts, off = syn.make_batch([70, 71])
    cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    vox = S.voxelize(torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda), 2, cfg)
    n = int(vox["n"].item())
    idx = vox["coords"][:n].cpu().numpy()
    shape = [41, 1440, 1440]
    td = torch.bfloat16
    feats = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(td).float().numpy()
    kk, ss, pp = [3] * 3, [s] * 3, [p] * 3
    conv = (spconv.SubMConv3d(cin, cout, kk, padding=[1, 1, 1], bias=False, indice_key="a") if mode == "subm"
            else spconv.SparseConv3d(cin, cout, kk, stride=ss, padding=pp, bias=False)).to(cuda)
    w = conv.weight.detach().to(td).float().cpu().numpy()
    x = torch.from_numpy(feats).to(cuda).to(td).requires_grad_(True)
    out = conv(spconv.SparseConvTensor(x, torch.from_numpy(idx).to(cuda), shape, 2))
    oi = out.indices.cpu().numpy()
    dy = torch.from_numpy(rng.standard_normal((oi.shape[0], cout)).astype(np.float32)).to(td).float().numpy()
    (out.features.float() * torch.from_numpy(dy).to(cuda)).sum().backward()
    if mode == "subm":
        pin, pout, pnum = oracle.rulebook_subm(idx, shape, kk)
        order = np.arange(n)
    else:
        o_idx, o_shape, pin, pout, pnum = oracle.rulebook_strided(idx, shape, kk, ss, pp)
        key = lambda a: ((a[:, 0].astype(np.int64) * o_shape[0] + a[:, 1]) * o_shape[1] + a[:, 2]) * o_shape[2] + a[:, 3]
        assert np.array_equal(np.sort(key(oi)), np.sort(key(o_idx)))
        lut = {int(v): i for i, v in enumerate(key(oi))}
        order = np.array([lut[int(v)] for v in key(o_idx)])
    dx_w, dw_w = oracle.conv_backward(feats, w, pin, pout, pnum, dy[order])
    assert n > 30000 and oi.shape[0] > 30000
    _close(x.grad.float().cpu().numpy(), dx_w, 3e-2)
    _close(conv.weight.grad.float().cpu().numpy(), dw_w, 3e-2)


def test_amp_training_step_as_the_reference_runs_it(cuda, rng):
    """tools/train_utils/train_utils.py:135,169-176: GradScaler + autocast around the forward, scaler.scale(loss).backward(),
    scaler.unscale_, clip_grad_norm_, scaler.step, scaler.update.  Under autocast the convolutions run fp16 features and
    weights with fp32 accumulation (the fp16 MFMA kernels), the fused BatchNorm + ReLU nodes take fp16 rows; the scaled
    loss goes back through SparseConvFunction and bn_act.  Against the same step in fp32 without AMP: finite, unskipped
    steps; unscaled gradients within 12 % (L2) of the fp32 ones; identical clipping decisions; weights that
    moved the same way."""
    import copy
    from findnpropagate_amd import synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.array([96, 88, 40])
    net32 = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": "fp32"}, 5, grid), 0).to(cuda).train()
    net16 = copy.deepcopy(net32)
    net16.fnp_dtype = "fp16"
    feats, idx = _random_sparse(rng, 2, net32.sparse_shape, 5000, 5)
    bd = lambda: {"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda).float(), "batch_size": 2}
    loss_of = lambda out: sum((t.features.float() ** 2).mean() for t in list(out["multi_scale_3d_features"].values()) + [out["encoded_spconv_tensor"]])
    clip = 10.0                                         # optim_cfg.GRAD_NORM_CLIP of the nuScenes configs
    opt32 = torch.optim.SGD(net32.parameters(), lr=1e-3)
    opt16 = torch.optim.SGD(net16.parameters(), lr=1e-3)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 12)
    p0 = {k: v.detach().clone() for k, v in net32.named_parameters()}
    for step in range(2):
        opt32.zero_grad()
        l32 = loss_of(net32(bd()))
        l32.backward()
        n32 = torch.nn.utils.clip_grad_norm_(net32.parameters(), clip)
        g32 = {k: v.grad.clone() for k, v in net32.named_parameters() if v.grad is not None}
        opt32.step()

        opt16.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            out = net16(bd())
            assert out["multi_scale_3d_features"]["x_conv2"].features.dtype in (torch.float16, torch.float32)
            l16 = loss_of(out)
        scaler.scale(l16).backward()
        scaler.unscale_(opt16)
        n16 = torch.nn.utils.clip_grad_norm_(net16.parameters(), clip)
        g16 = {k: v.grad.clone() for k, v in net16.named_parameters() if v.grad is not None}
        scale_before = scaler.get_scale()
        scaler.step(opt16)
        scaler.update()
        assert scaler.get_scale() == scale_before, "no inf / nan in the unscaled gradients: the step was not skipped"
        assert torch.isfinite(l16) and abs(float(l16) - float(l32)) <= 2e-2 * abs(float(l32))
        assert set(g16) == set(g32) and len(g16) > 50
        assert torch.isfinite(n16) and abs(float(n16) - float(n32)) <= 3e-2 * float(n32)
        for k in g32:
            # fp16 storage through up to 21 layers (+ ReLU decisions within rounding of zero): worst element within 30 % of the
            # tensor's scale, the tensor as a whole within 12 % in L2 (measured: up to 7 %, a BatchNorm bias behind 17 layers; single elements of the
            # small stage-4 weight gradients up to 18 % of the tensor's largest)
            s = max(float(g32[k].abs().max()), 1e-8)
            assert float((g16[k].float() - g32[k]).abs().max()) <= 3e-1 * s + 1e-7, (step, k)
            assert float((g16[k].float() - g32[k]).norm()) <= 1.2e-1 * float(g32[k].norm()) + 1e-7, (step, k)
    moved = 0
    for (k, a), (_, b) in zip(net32.named_parameters(), net16.named_parameters()):
        d32, d16 = a.detach() - p0[k], b.detach() - p0[k]
        if float(d32.abs().max()) > 0:
            moved += 1
            assert float((d16 - d32).abs().max()) <= 3e-1 * float(d32.abs().max()) + 1e-9, k   # (the bound of the gradients above)
    assert moved > 50


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Cout,K3,Cin", [(16, (3, 3, 3), 5), (32, (3, 3, 3), 16), (128, (3, 1, 1), 128), (64, (3, 3, 3), 64)])
def test_pack_weight_kernel_equals_the_torch_ops(cuda, rng, dtype, Cout, K3, Cin):
    """fnp_pack_weight: the packed slabs and both mirror forms of the training path, in one launch, against pack_weight + flip
    (+ transpose): the same bits."""
    from findnpropagate_amd import sparse as S
    w = torch.from_numpy(rng.standard_normal((Cout, *K3, Cin)).astype(np.float32)).to(cuda)
    ref = S.pack_weight(w, dtype)
    for mirror, want in ((0, None), (1, ref.flip(0)), (2, ref.flip(0).transpose(1, 2).contiguous()), (3, ref.transpose(1, 2).contiguous())):
        p, m = S.pack_weight_train(w, dtype, mirror)
        assert torch.equal(p, ref) and p.is_contiguous()
        assert (m is None) if want is None else torch.equal(m, want)


@pytest.mark.gpu
@pytest.mark.parametrize("C,dtype", [(5, torch.float32), (16, torch.bfloat16), (64, torch.bfloat16), (128, torch.float16)])
def test_wgrad_in_the_module_layout(cuda, rng, C, dtype):
    """conv_wgrad(module_shape=...): the final reduction writes the parameter's (Cout, kD, kH, kW, Cin) layout — the packed
    (K, Cout, Cin) result permuted, value for value."""
    from findnpropagate_amd import sparse as S
    Cout = 16 if C == 5 else C
    B, shape = 2, [9, 40, 44]
    idx = np.unique(np.stack([rng.integers(0, B, 4000), rng.integers(0, shape[0], 4000), rng.integers(0, shape[1], 4000),
                              rng.integers(0, shape[2], 4000)], 1).astype(np.int32), axis=0)
    n = idx.shape[0]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    x = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)).to(cuda).to(dtype)
    dy = torch.from_numpy(rng.standard_normal((n, Cout)).astype(np.float32)).to(cuda).to(dtype)
    packed = S.conv_wgrad(x, dy, rb, n_dev, C, Cout)
    module = S.conv_wgrad(x, dy, rb, n_dev, C, Cout, module_shape=(Cout, 3, 3, 3, C))
    assert module.shape == (Cout, 3, 3, 3, C)
    assert torch.equal(module, packed.permute(1, 0, 2).reshape(Cout, 3, 3, 3, C))


@pytest.mark.gpu
def test_prefetched_strided_rulebooks_change_nothing(cuda, rng):
    """Training forward + backward with the strided layers' rulebooks asked for ahead (counts back on a side stream,
    spconv/conv.py prefetch) against the synchronous path: same outputs, same gradients, same BatchNorm buffers, and the
    module's parameter names are still the reference's."""
    import copy
    from findnpropagate_amd import synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    from findnpropagate_amd.spconv import conv as C
    grid = np.array([96, 88, 40])
    net_a = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": "bf16"}, 5, grid), 0).to(cuda).train()
    net_b = copy.deepcopy(net_a)
    names = [k for k, _ in net_a.named_parameters()]
    feats, idx = _random_sparse(rng, 2, net_a.sparse_shape, 5000, 5)
    bd = lambda: {"voxel_features": torch.from_numpy(feats).to(cuda), "voxel_coords": torch.from_numpy(idx).to(cuda).float(), "batch_size": 2}
    loss_of = lambda out: sum((t.features.float() ** 2).mean() for t in list(out["multi_scale_3d_features"].values()) + [out["encoded_spconv_tensor"]])
    outs = []
    for net, pf in ((net_a, True), (net_b, False)):
        C.PREFETCH = pf
        try:
            for _ in range(2):
                net.zero_grad()
                out = net(bd())
                loss_of(out).backward()
        finally:
            C.PREFETCH = True
        outs.append(out)
    assert [k for k, _ in net_a.named_parameters()] == names and not any("_fnp" in k for k in net_a.state_dict())
    for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        a, b = outs[0]["multi_scale_3d_features"][k], outs[1]["multi_scale_3d_features"][k]
        assert torch.equal(a.indices, b.indices) and torch.equal(a.features, b.features), k
    assert torch.equal(outs[0]["encoded_spconv_tensor"].features, outs[1]["encoded_spconv_tensor"].features)
    for (k, pa), (_, pb) in zip(net_a.named_parameters(), net_b.named_parameters()):
        assert torch.equal(pa.grad, pb.grad), k
    for (k, ba), (_, bb) in zip(net_a.named_buffers(), net_b.named_buffers()):
        assert torch.equal(ba, bb), k
        if k.endswith("num_batches_tracked"):
            assert int(ba) == 2, k


def test_conv_input_weight_gradient_on_pair_lists(cuda, rng):
    """The few-pairs weight-gradient kernel (conv_input: 5 -> 16, f32 features and gradients) on the rulebook's pair lists
    against its sweep over the table: the same sum in another grouping of the rows — equal to f32 rounding, and identical
    from run to run."""
    from findnpropagate_amd import sparse as S
    B, shape = 2, [9, 60, 64]
    idx = np.unique(np.stack([rng.integers(0, B, 9000), rng.integers(0, shape[0], 9000), rng.integers(0, shape[1], 9000),
                              rng.integers(0, shape[2], 9000)], 1).astype(np.int32), axis=0)
    n = idx.shape[0]
    d_idx = torch.from_numpy(idx).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(d_idx, n_dev, S.build_grid(d_idx, n_dev, B, shape), 3)
    x = torch.from_numpy(rng.standard_normal((n, 5)).astype(np.float32)).to(cuda)
    dy = torch.from_numpy(rng.standard_normal((n, 16)).astype(np.float32)).to(cuda)
    table = S.conv_wgrad(x, dy, rb, n_dev, 5, 16, pairs=False)
    lists = S.conv_wgrad(x, dy, rb, n_dev, 5, 16, pairs=True)
    again = S.conv_wgrad(x, dy, rb, n_dev, 5, 16, pairs=True)
    assert getattr(rb, "_pairs", None) is not None
    assert torch.equal(lists, again)
    scale = float(table.abs().max())
    assert float((lists - table).abs().max()) <= 1e-5 * scale
    nbr = rb.nbr[:, :n].cpu().numpy()
    want = np.zeros((27, 16, 5), np.float64)
    xs, dys = x.cpu().numpy().astype(np.float64), dy.cpu().numpy().astype(np.float64)
    for k in range(27):
        o = np.nonzero(nbr[k] >= 0)[0]
        want[k] = dys[o].T @ xs[nbr[k][o]]
    assert np.abs(lists.cpu().numpy() - want).max() <= 1e-5 * scale
