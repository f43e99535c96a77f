"""Seeded input recipes shared by tests/golden/make_backbone_extra_golden.py (which runs the reference on them) and
tests/test_backbone_tree.py (which runs the product and the oracle on them).  Data recipes only."""
import numpy as np

from findnpropagate_amd import synthetic as syn
from oracle import oracle as O

ADAPTER_GRID = [64, 64, 16]     # grid_size (x, y, z) of the model the checkpoints are loaded into (weights do not depend on it)
ADAPTER_SEED = 3
BATCH_SCENES = 3                # scenes 0 and 2 hold voxels, scene 1 is EMPTY
BATCH_HALF = 4.8
BATCH_STRIDE = 8


def disk_layouts(state_dict, conv_keys):
    """checkpoints of one model in the three convolution-weight layouts detector3d_template.py:401-433 tells apart; the model's
    own layout is (Cout, kD, kH, kW, Cin)"""
    base = {k: v.detach().clone() for k, v in state_dict.items()}
    one_x, native = dict(base), dict(base)
    for k in conv_keys:
        one_x[k] = base[k].permute(1, 2, 3, 4, 0).contiguous()      # spconv 1.x: (kD, kH, kW, Cin, Cout)
        native[k] = base[k].permute(1, 2, 3, 0, 4).contiguous()     # spconv 2.x native: (kD, kH, kW, Cout, Cin)
    return {"implicit_gemm_2x": base, "spconv_1x": one_x, "native_2x": native}


def batch_scene():
    """voxel features / coords of a 3-scene batch whose middle scene is empty, and the grid (x, y, z)"""
    rng = [-BATCH_HALF, -BATCH_HALF, -5.0, BATCH_HALF, BATCH_HALF, 3.0]
    feats, coords = [], []
    for b, seed in ((0, 11), (2, 12)):
        pts = syn.make_scene(seed)
        pts = pts[(np.abs(pts[:, 0]) < BATCH_HALF) & (np.abs(pts[:, 1]) < BATCH_HALF)][::BATCH_STRIDE]
        v, c, n = O.voxelize(pts, syn.VOXEL_SIZE, rng, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
        coords.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
        feats.append(O.mean_vfe(v, n))
    grid = np.round((np.array(rng[3:]) - np.array(rng[:3])) / np.array(syn.VOXEL_SIZE)).astype(np.int64)
    return np.concatenate(feats), np.concatenate(coords), grid
