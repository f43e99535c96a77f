#!/usr/bin/env python3
"""Pin the backbone's LAYER GRAPH with the reference's own constructors and its own forward().

Runs in the build container only (needs /root/reference).  What is executed is the reference's
pcdet/models/backbones_3d/spconv_backbone.py (post_act_block, SparseBasicBlock, VoxelBackBone8x,
VoxelResBackBone8x: __init__ AND forward) and pcdet/utils/spconv_utils.py, imported from where they
lie, with `spconv` resolved exactly as INTEGRATION.md §2 prescribes:

    sys.modules["spconv"] = findnpropagate_amd.spconv
    sys.modules["spconv.pytorch"] = findnpropagate_amd.spconv.pytorch

so every SubMConv3d / SparseConv3d / SparseSequential in the dumped tree is the product's class
constructed with the arguments the REFERENCE passes.  Two fixtures (data, never reference source):

  backbone_tree.json     per backbone class: every module of named_modules() (class, channels, kernel,
                         stride, padding, indice_key, bias, BN eps / momentum / affine), state_dict keys +
                         shapes + dtypes, the order in which forward() calls the leaf modules, the
                         constructor's public attributes (sparse_shape, num_point_features, backbone_channels)
  backbone_forward.npz   (backbone_forward_plain.npz: the same for VoxelBackBone8x)
                         one small scene through the reference's VoxelResBackBone8x.forward in eval mode:
                         inputs (voxel features + coords), weight seed, and the five outputs
                         (encoded_spconv_tensor, x_conv1..x_conv4: indices + f32 features)

There is no spconv here and no GPU, so for the forward fixture the harness (this file, not shipped
code) backs SparseConvolution.forward by the CPU oracle's convolution primitive (oracle.subm_conv /
sparse_conv, self-pinned in tests/test_oracle_spconv.py).  Everything AROUND the primitive — which conv
follows which, indice_key reuse, BatchNorm1d(eps, momentum) in eval mode, ReLU placement, the residual
wiring of SparseBasicBlock, the outputs dictionary — is the reference's code, run as is.  The script
asserts oracle.backbone_forward (the oracle's own restatement of that graph) reproduces it before writing.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("FNP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import findnpropagate_amd as fnp  # noqa: E402,F401
from findnpropagate_amd import spconv as fnp_spconv  # noqa: E402
from findnpropagate_amd import synthetic as syn  # noqa: E402
from oracle import oracle as O  # noqa: E402

# ---- INTEGRATION.md §2, verbatim -----------------------------------------------------------------
sys.modules["spconv"] = fnp_spconv                      # pcdet/utils/spconv_utils.py:3-10 imports it
sys.modules["spconv.pytorch"] = fnp_spconv.pytorch

HALF = 4.8                 # crop of the synthetic scene: 128 x 128 x 40 voxels
STRIDE = 3                 # every third point: ~1.5 k voxels keeps the fixture small
SEED_SCENE, SEED_WEIGHTS = 0, 0


class Cfg(dict):           # model_cfg is an EasyDict in the reference; only .get() is used on this path
    __getattr__ = dict.get


def shell(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def load_reference():
    """pcdet.models.backbones_3d.spconv_backbone without running the package __init__s (they import the
    whole detector zoo and its compiled ops)."""
    p = os.path.join(REF, "pcdet")
    shell("pcdet", p)
    shell("pcdet.utils", os.path.join(p, "utils"))
    shell("pcdet.models", os.path.join(p, "models"))
    shell("pcdet.models.backbones_3d", os.path.join(p, "models", "backbones_3d"))
    return importlib.import_module("pcdet.models.backbones_3d.spconv_backbone")


def describe(m):
    d = {"class": type(m).__name__}
    if isinstance(m, fnp_spconv.conv.SparseConvolution):
        d.update(in_channels=m.in_channels, out_channels=m.out_channels, kernel_size=list(m.kernel_size),
                 stride=list(m.stride), padding=list(m.padding), subm=bool(m.subm), indice_key=m.indice_key,
                 bias=m.bias is not None)
    elif isinstance(m, nn.BatchNorm1d):
        d.update(num_features=m.num_features, eps=m.eps, momentum=m.momentum, affine=m.affine,
                 track_running_stats=m.track_running_stats)
    return d


def oracle_conv_forward(self, input):
    """harness stand-in for the convolution primitive (see the module docstring)"""
    x = O.SparseTensor(input.features.detach().numpy(), input.indices.numpy(), input.spatial_shape, input.batch_size)
    x.rulebooks = input.indice_dict
    w = self.weight.detach().numpy()
    o = O.subm_conv(x, w, self.indice_key) if self.subm else O.sparse_conv(x, w, self.stride, self.padding)
    assert self.bias is None
    return fnp_spconv.SparseConvTensor(torch.from_numpy(o.features), torch.from_numpy(o.indices), o.spatial_shape,
                                       input.batch_size, indice_dict=input.indice_dict)


def small_scene():
    pts = syn.make_scene(SEED_SCENE)
    pts = pts[(np.abs(pts[:, 0]) < HALF) & (np.abs(pts[:, 1]) < HALF)][::STRIDE]
    rng = [-HALF, -HALF, -5.0, HALF, HALF, 3.0]
    v, c, n = O.voxelize(pts, syn.VOXEL_SIZE, rng, syn.MAX_POINTS_PER_VOXEL, syn.MAX_VOXELS_TEST)
    coords = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    grid = np.round((np.array(rng[3:]) - np.array(rng[:3])) / np.array(syn.VOXEL_SIZE)).astype(np.int64)
    return O.mean_vfe(v, n), coords, grid


def main():
    ref = load_reference()
    out_dir = os.path.dirname(os.path.abspath(__file__))
    full_grid = np.array([1440, 1440, 40])                       # transfusion_lidar.yaml:6,54 -> grid_size (x, y, z)
    tree = {"_made_by": "tests/golden/make_backbone_tree_golden.py", "_reference": "pcdet/models/backbones_3d/spconv_backbone.py",
            "_shim": {"spconv": sys.modules["spconv"].__name__, "spconv.pytorch": sys.modules["spconv.pytorch"].__name__}}
    feats, coords, grid = small_scene()

    for cls_name, cfg in (("VoxelResBackBone8x", {"NAME": "VoxelResBackBone8x", "USE_BIAS": False}),   # transfusion_lidar.yaml:67-69
                          ("VoxelBackBone8x", {"NAME": "VoxelBackBone8x"})):
        net = getattr(ref, cls_name)(model_cfg=Cfg(**cfg), input_channels=5, grid_size=full_grid)
        assert type(net).__module__.startswith("pcdet."), type(net).__module__
        entry = {"ctor": {"model_cfg": cfg, "input_channels": 5, "grid_size": full_grid.tolist()},
                 "sparse_shape": [int(v) for v in net.sparse_shape], "num_point_features": net.num_point_features,
                 "backbone_channels": dict(net.backbone_channels),
                 "modules": [{"name": n, **describe(m)} for n, m in net.named_modules() if n],
                 "state_dict": {k: [list(t.shape), str(t.dtype).replace("torch.", "")] for k, t in net.state_dict().items()}}

        # order in which forward() reaches the leaf modules: run the reference's forward on the small scene
        small = getattr(ref, cls_name)(model_cfg=Cfg(**cfg), input_channels=5, grid_size=grid)
        syn.init_backbone_weights(small, SEED_WEIGHTS).eval()
        names = {m: n for n, m in small.named_modules()}
        calls, hooks = [], []
        for m in small.modules():
            if not list(m.children()):
                hooks.append(m.register_forward_hook(lambda mod, i, o: calls.append(names[mod])))
        plain = fnp_spconv.conv.SparseConvolution.forward
        fnp_spconv.conv.SparseConvolution.forward = oracle_conv_forward
        try:
            with torch.no_grad():
                bd = small({"voxel_features": torch.from_numpy(feats), "voxel_coords": torch.from_numpy(coords),
                            "batch_size": 1})
        finally:
            fnp_spconv.conv.SparseConvolution.forward = plain
            for h in hooks:
                h.remove()
        entry["call_order"] = calls
        entry["forward_keys"] = sorted(k for k in bd if k not in ("voxel_features", "voxel_coords", "batch_size"))
        entry["encoded_spconv_tensor_stride"] = bd["encoded_spconv_tensor_stride"]
        entry["multi_scale_3d_strides"] = dict(bd["multi_scale_3d_strides"])
        tree[cls_name] = entry

        # the reference's forward() output itself (over the oracle's convolution primitive), held against oracle.backbone_forward
        res = cls_name == "VoxelResBackBone8x"
        got = {"out": bd["encoded_spconv_tensor"], **bd["multi_scale_3d_features"]}
        sd = {k: t.detach().numpy() for k, t in small.state_dict().items()}
        want = O.backbone_forward(sd, feats, coords, 1, small.sparse_shape, table=None if res else O.PLAIN_BACKBONE8X)
        arrays = {"voxel_features": feats, "voxel_coords": coords, "grid_size": grid,
                  "weight_seed": np.int64(SEED_WEIGHTS), "scene": np.array([SEED_SCENE, STRIDE], np.int64),
                  "half_range": np.float32(HALF)}
        for k, t in got.items():
            f, i = t.features.numpy(), t.indices.numpy()
            assert np.array_equal(i, want[k].indices), k
            err = np.abs(f - want[k].features).max()
            assert err <= 1e-5 * max(1.0, np.abs(f).max()), (k, err)   # torch's BatchNorm1d vs the folded scale / shift
            print(f"{cls_name} {k}: {i.shape[0]} sites x {f.shape[1]}, |reference forward - oracle.backbone_forward| max {err:.2e}")
            arrays[k + "_indices"], arrays[k + "_features"] = i, f.astype(np.float32)
            arrays[k + "_spatial_shape"] = np.array(t.spatial_shape, np.int64)
        # the weights are a function of (tree, seed); a checksum guards the recipe itself
        arrays["state_checksum"] = np.array([float(np.abs(v).astype(np.float64).sum()) for v in sd.values()])
        np.savez_compressed(os.path.join(out_dir, "backbone_forward.npz" if res else "backbone_forward_plain.npz"), **arrays)

    with open(os.path.join(out_dir, "backbone_tree.json"), "w") as f:
        json.dump(tree, f, indent=1, sort_keys=True)
    print("wrote backbone_tree.json, backbone_forward.npz, backbone_forward_plain.npz")


if __name__ == "__main__":
    main()
