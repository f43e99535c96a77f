"""SURVEY 8 (f)1, second half: the first BaseBEVBackbone block (ZeroPad2d(1) + Conv2d(256 -> 128, 3x3) + BatchNorm2d + ReLU,
base_bev_backbone.py:31-40) evaluated on the sparse rows of the encoded tensor, against torch's own modules on the densified
map (height_compression.py:20-24).  f32 rows: <= 1e-4 (only the summation order differs); bf16 rows: bf16 products."""
import numpy as np
import pytest
import torch

from findnpropagate_amd import spconv
from findnpropagate_amd.backbones_2d import BaseBEVBackbone, HeightCompression

pytestmark = pytest.mark.gpu
CFG = {"LAYER_NUMS": [1, 1], "LAYER_STRIDES": [1, 2], "NUM_FILTERS": [128, 256], "UPSAMPLE_STRIDES": [1, 2],
       "NUM_UPSAMPLE_FILTERS": [256, 256], "USE_CONV_FOR_NO_STRIDE": True}


def _encoded(rng, cuda, B, shape, n, dtype=torch.float32, clustered=True):
    D, H, W = shape
    if clustered:   # a few blobs + scattered sites: cells with full, partial and empty 3x3 windows, map borders included
        cy, cx = rng.integers(0, H, 6), rng.integers(0, W, 6)
        pts = []
        for b in range(B):
            yy = np.clip((cy[:, None] + rng.normal(0, 4, (6, n // (6 * B) + 1))).astype(int), 0, H - 1).ravel()
            xx = np.clip((cx[:, None] + rng.normal(0, 4, (6, n // (6 * B) + 1))).astype(int), 0, W - 1).ravel()
            zz = rng.integers(0, D, yy.shape[0])
            pts.append(np.stack([np.full_like(yy, b), zz, yy, xx], 1))
        idx = np.unique(np.concatenate(pts), axis=0)
        corners = np.array([[0, 0, 0, 0], [0, D - 1, H - 1, W - 1], [B - 1, 0, 0, W - 1], [B - 1, D - 1, H - 1, 0]])
        idx = np.unique(np.concatenate([idx, corners]), axis=0)
    else:
        lin = rng.choice(B * D * H * W, size=n, replace=False)
        b, rem = np.divmod(lin, D * H * W); z, rem = np.divmod(rem, H * W); y, x = np.divmod(rem, W)
        idx = np.stack([b, z, y, x], 1)
    idx = idx[rng.permutation(idx.shape[0])].astype(np.int32)
    feats = torch.from_numpy(rng.standard_normal((idx.shape[0], 128)).astype(np.float32)).to(cuda).to(dtype)
    return spconv.SparseConvTensor(feats, torch.from_numpy(idx).to(cuda), shape, B)


def _net(cuda, rng, cfg=CFG, **extra):
    net = BaseBEVBackbone(dict(cfg, **extra), 256).to(cuda).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, m.num_features).astype(np.float32)))
                m.bias.copy_(torch.from_numpy(rng.standard_normal(m.num_features).astype(np.float32) * 0.3))
                m.running_mean.copy_(torch.from_numpy(rng.standard_normal(m.num_features).astype(np.float32) * 0.2))
                m.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, m.num_features).astype(np.float32)))
    return net


def test_module_tree_is_the_references(cuda, rng):
    net = _net(cuda, rng)
    keys = set(net.state_dict())
    for k in ("blocks.0.1.weight", "blocks.0.2.running_var", "blocks.0.4.weight", "blocks.1.1.weight", "deblocks.0.0.weight",
              "deblocks.1.0.weight", "deblocks.1.1.num_batches_tracked"):
        assert k in keys, k
    assert net.blocks[0][1].weight.shape == (128, 256, 3, 3) and net.num_bev_features == 512
    assert isinstance(net.deblocks[0][0], torch.nn.Conv2d) and isinstance(net.deblocks[1][0], torch.nn.ConvTranspose2d)


@pytest.mark.parametrize("B,shape,n,clustered", [(2, [2, 180, 180], 9000, True), (1, [2, 37, 41], 300, True), (3, [2, 64, 60], 4000, False),
                                                 (1, [2, 24, 24], 1, False)])
def test_first_block_on_sparse_rows_equals_torch_on_the_dense_map(cuda, rng, B, shape, n, clustered):
    net = _net(cuda, rng)
    t = _encoded(rng, cuda, B, shape, n, clustered=clustered)
    with torch.no_grad():
        got = net.first_block_from_sparse(t)
        dense = HeightCompression({"NUM_BEV_FEATURES": 256})({"encoded_spconv_tensor": t, "encoded_spconv_tensor_stride": 8})["spatial_features"]
        want = net.blocks[0][:4](dense.float())
    assert got.dtype == torch.float32 and tuple(got.shape) == tuple(want.shape) == (B, 128, shape[1], shape[2])
    err = (got - want).abs().max().item()
    assert err <= 1e-4 * max(1.0, want.abs().max().item()), err
    # cells without any input row in their 3x3 window hold relu(shift) exactly
    occ = (dense != 0).any(dim=1, keepdim=True).float()
    reach = torch.nn.functional.max_pool2d(occ, 3, stride=1, padding=1) > 0
    bn = net.blocks[0][2]
    shift = bn.bias - bn.running_mean * bn.weight / torch.sqrt(bn.running_var + bn.eps)
    bg = torch.relu(shift).view(1, -1, 1, 1).expand_as(got)
    far = (~reach).expand_as(got)
    assert far.any() and torch.allclose(got[far], bg[far], rtol=0, atol=1e-6)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
def test_first_block_on_16_bit_rows(cuda, rng, dtype, tol):
    """rows in the engine's storage dtype (FNP_OUT_DTYPE native): 16-bit products with f32 accumulation against the f32 modules
    fed the same rounded rows and weights rounded the same way."""
    net = _net(cuda, rng)
    t = _encoded(rng, cuda, 2, [2, 90, 90], 5000, dtype=dtype)
    with torch.no_grad():
        got = net.first_block_from_sparse(t)
        dense = t.replace_feature(t.features.float()).dense().view(2, 256, 90, 90)
        conv = net.blocks[0][1]
        w = conv.weight.to(dtype).float()
        want = torch.relu(net.blocks[0][2](torch.nn.functional.conv2d(torch.nn.functional.pad(dense, (1, 1, 1, 1)), w)))
    assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())


def test_forward_takes_the_sparse_rows_and_equals_the_dense_path(cuda, rng):
    """forward(data_dict) in eval mode with the encoded tensor present == the reference's dense path (spatial_features through
    every torch module); train mode and FNP_SPARSE_FIRST False take the dense path."""
    net = _net(cuda, rng)
    t = _encoded(rng, cuda, 2, [2, 96, 88], 6000)
    dense = HeightCompression({"NUM_BEV_FEATURES": 256})({"encoded_spconv_tensor": t, "encoded_spconv_tensor_stride": 8})["spatial_features"]
    calls = []
    orig = net.first_block_from_sparse
    net.first_block_from_sparse = lambda tt: (calls.append(1), orig(tt))[1]
    with torch.no_grad():
        a = net({"encoded_spconv_tensor": t, "spatial_features": dense})["spatial_features_2d"]
        assert calls == [1]
        b = net({"spatial_features": dense})["spatial_features_2d"]
        net.sparse_first = False
        c = net({"encoded_spconv_tensor": t, "spatial_features": dense})["spatial_features_2d"]
    assert calls == [1] and torch.equal(b, c)
    assert tuple(a.shape) == (2, 512, 96, 88)
    assert (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())
