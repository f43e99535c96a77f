"""Host logic that needs no GPU: BaseBEVBackbone's module tree / dense path (the reference's contract), sparse.tiled_fits,
the sorted / compact-rulebook switches."""
import numpy as np
import torch

from findnpropagate_amd import sparse as S
from findnpropagate_amd.backbones_2d import BaseBEVBackbone

CFG = {"LAYER_NUMS": [5, 5], "LAYER_STRIDES": [1, 2], "NUM_FILTERS": [128, 256], "UPSAMPLE_STRIDES": [1, 2],
       "NUM_UPSAMPLE_FILTERS": [256, 256], "USE_CONV_FOR_NO_STRIDE": True}     # transfusion_lidar.yaml:75-82


def test_module_tree_matches_base_bev_backbone():
    """base_bev_backbone.py:27-80: blocks[i] = ZeroPad2d, Conv2d(3x3, stride), BN, ReLU, then LAYER_NUMS x (Conv2d, BN, ReLU);
    deblocks[i] = Conv2d / ConvTranspose2d + BN + ReLU; num_bev_features = sum of the up-sample filters."""
    net = BaseBEVBackbone(CFG, 256)
    assert len(net.blocks) == 2 and len(net.deblocks) == 2 and net.num_bev_features == 512
    assert [type(m).__name__ for m in net.blocks[0]][:4] == ["ZeroPad2d", "Conv2d", "BatchNorm2d", "ReLU"] and len(net.blocks[0]) == 4 + 3 * 5
    assert net.blocks[0][1].weight.shape == (128, 256, 3, 3) and net.blocks[1][1].stride == (2, 2) and net.blocks[1][1].weight.shape == (256, 128, 3, 3)
    assert net.blocks[0][2].eps == 1e-3 and net.blocks[0][2].momentum == 0.01
    assert isinstance(net.deblocks[0][0], torch.nn.Conv2d) and net.deblocks[0][0].kernel_size == (1, 1)      # USE_CONV_FOR_NO_STRIDE
    assert isinstance(net.deblocks[1][0], torch.nn.ConvTranspose2d) and net.deblocks[1][0].stride == (2, 2)
    keys = list(net.state_dict())
    assert keys[0] == "blocks.0.1.weight" and "blocks.0.16.weight" in keys and "deblocks.1.1.running_mean" in keys


def test_dense_path_is_the_reference_forward():
    torch.manual_seed(0)
    net = BaseBEVBackbone(dict(CFG, LAYER_NUMS=[1, 1]), 256).eval()
    x = torch.randn(2, 256, 12, 12)
    with torch.no_grad():
        out = net({"spatial_features": x})
        b0 = net.blocks[0](x)
        b1 = net.blocks[1](b0)
        want = torch.cat([net.deblocks[0](b0), net.deblocks[1](b1)], dim=1)
    assert set(out) == {"spatial_features", "spatial_features_2d"} and torch.equal(out["spatial_features_2d"], want)


def test_switches_and_size_limits():
    assert S.tiled_fits(1 << 20, 64, 1 << 20, 1 << 20)
    assert not S.tiled_fits(1 << 25, 64, 1 << 20, 1 << 20)          # 2^25 rows x 128 bytes = 4 GiB of features
    assert not S.tiled_fits(1 << 20, 64, 1 << 25, 1 << 20)          # the int32 table beyond 32-bit offsets
    assert S.sorted_by_default(128, 128, torch.bfloat16, 1 << 20) and not S.sorted_by_default(128, 128, torch.bfloat16, 65536)
    assert not S.sorted_by_default(64, 64, torch.bfloat16, 1 << 20) and not S.sorted_by_default(128, 128, torch.float32, 1 << 20)
    assert (5, 16) in S.ELL_SHAPES and (16, 32) in S.ELL_SHAPES and (32, 32) not in S.ELL_SHAPES
