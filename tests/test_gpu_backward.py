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
        def apply(f, weight, rb, n_out_dev, n_in_dev):
            n_out = int(n_out_dev.item())
            Cout, Cin = weight.shape[0], weight.shape[-1]
            wk = weight.reshape(Cout, rb.K, Cin)
            fz = torch.cat([f.float(), f.new_zeros((1, Cin), dtype=torch.float32)], 0)
            out = f.new_zeros((rb.cap_out, Cout), dtype=torch.float32)
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
    assert abs(loss_a - loss_b) <= 1e-4 * abs(loss_b)
    assert set(grads_a) == set(grads_b) and len(grads_a) > 60
    for name in grads_a:
        ga, gb = grads_a[name].float().cpu().numpy(), grads_b[name].float().cpu().numpy()
        assert np.isfinite(ga).all(), name
        assert np.abs(ga - gb).max() <= 2e-3 * max(np.abs(gb).max(), 1e-6), (name, np.abs(ga - gb).max(), np.abs(gb).max())


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
