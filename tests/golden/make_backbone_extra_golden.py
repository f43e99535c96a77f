#!/usr/bin/env python3
"""Two more fixtures from the reference's own code (build container only: needs /root/reference).

  state_dict_adapter.json   pcdet/models/detectors/detector3d_template.py:401-433 `_load_state_dict` — the adapter
                            that turns spconv 1.x convolution weights (kD, kH, kW, Cin, Cout) into the layout of the model
                            it loads into — RUN on checkpoints in the 1.x layout, the 2.x "native" layout
                            (kD, kH, kW, Cout, Cin) and this framework's (= spconv 2.x implicit-gemm) layout
                            (Cout, kD, kH, kW, Cin), for VoxelResBackBone8x; per key the sha256 of the parameter the
                            reference's function loaded, plus the keys its find_all_spconv_keys walk found.  The
                            function is not imported with its module (detector3d_template.py imports the whole detector
                            zoo and its compiled ops): its `def` is cut out of the source file where it lies with `ast`
                            and compiled against the reference's own find_all_spconv_keys (pcdet/utils/spconv_utils.py,
                            imported under INTEGRATION.md §2's alias).  Nothing of its text is written anywhere.
  backbone_forward_batch.npz  the reference's VoxelResBackBone8x.forward (eval) on a batch of THREE scenes of which
                            the middle one is EMPTY (batch_size 3, no voxel carries batch index 1), over the oracle's
                            convolution primitive like backbone_forward.npz.
"""
import ast
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_backbone_tree_golden as T  # noqa: E402  (aliases spconv, shells pcdet.*)
from make_backbone_tree_golden import O, REF, fnp_spconv, syn  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from backbone_recipes import ADAPTER_GRID, ADAPTER_SEED, BATCH_SCENES, batch_scene, disk_layouts  # noqa: E402


def reference_adapter():
    """the reference's _load_state_dict as a plain function (self, model_state_disk, *, strict=True)"""
    ref = T.load_reference()                                      # (pcdet.utils is shelled: spconv_utils imports cleanly)
    import importlib
    su = importlib.import_module("pcdet.utils.spconv_utils")
    path = os.path.join(REF, "pcdet", "models", "detectors", "detector3d_template.py")
    tree = ast.parse(open(path).read(), path)
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "_load_state_dict")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"find_all_spconv_keys": su.find_all_spconv_keys, "torch": torch}
    exec(compile(mod, path, "exec"), ns)
    return ns["_load_state_dict"], su.find_all_spconv_keys, ref


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def adapter_fixture():
    load, find_keys, ref = reference_adapter()
    # the model the checkpoint is loaded INTO is built by the reference's constructor over the aliased spconv
    mk = lambda: ref.VoxelResBackBone8x(model_cfg=T.Cfg(NAME="VoxelResBackBone8x", USE_BIAS=False), input_channels=5,
                                       grid_size=np.array(ADAPTER_GRID))
    src = syn.init_backbone_weights(mk(), ADAPTER_SEED)
    keys = sorted(find_keys(src))
    out = {"_made_by": "tests/golden/make_backbone_extra_golden.py",
           "_reference": "pcdet/models/detectors/detector3d_template.py:401-433, pcdet/utils/spconv_utils.py:15-29",
           "spconv_keys": keys, "layouts": {}}
    for name, disk in disk_layouts(src.state_dict(), keys).items():
        dst = mk()
        try:
            _, updated = load(dst, disk, strict=True)
        except RuntimeError as e:
            # the 2.x NATIVE layout is not one the adapter knows: non-square weights match neither of its two candidates, are
            # dropped ("has invalid shape") and the strict load that follows reports them missing.  (Square ones have the 1.x
            # shape and are permuted as if they were 1.x.)  The product must refuse the same checkpoint for the same keys.
            import re
            missing = re.search(r"Missing key\(s\) in state_dict: (.*?)\.\s*$", str(e), re.S)
            rejected = sorted(re.findall(r'"([^"]+)"', missing.group(1)))
            out["layouts"][name] = {"raises": "RuntimeError", "rejected_keys": rejected}
            print(f"adapter[{name}]: the reference raises RuntimeError; rejected keys: {rejected}")
            continue
        got = dst.state_dict()
        same = all(torch.equal(got[k], v) for k, v in src.state_dict().items())
        out["layouts"][name] = {"updated_keys": sorted(updated), "equals_source_model": bool(same),
                                "sha256": {k: sha(v) for k, v in got.items()}}
        print(f"adapter[{name}]: {len(updated)} keys loaded, equals the source model: {same}")
        assert same, name
    json.dump(out, open(os.path.join(HERE, "state_dict_adapter.json"), "w"), indent=1, sort_keys=True)


def batch_fixture():
    ref = T.load_reference()
    feats, coords, grid = batch_scene()
    assert set(np.unique(coords[:, 0]).tolist()) == {0, 2}, "the middle scene must be empty"
    net = ref.VoxelResBackBone8x(model_cfg=T.Cfg(NAME="VoxelResBackBone8x", USE_BIAS=False), input_channels=5, grid_size=grid)
    syn.init_backbone_weights(net, T.SEED_WEIGHTS).eval()
    plain = fnp_spconv.conv.SparseConvolution.forward
    fnp_spconv.conv.SparseConvolution.forward = T.oracle_conv_forward
    try:
        with torch.no_grad():
            bd = net({"voxel_features": torch.from_numpy(feats), "voxel_coords": torch.from_numpy(coords), "batch_size": BATCH_SCENES})
    finally:
        fnp_spconv.conv.SparseConvolution.forward = plain
    got = {"out": bd["encoded_spconv_tensor"], **bd["multi_scale_3d_features"]}
    sd = {k: t.detach().numpy() for k, t in net.state_dict().items()}
    want = O.backbone_forward(sd, feats, coords, BATCH_SCENES, net.sparse_shape)
    arrays = {"voxel_features": feats, "voxel_coords": coords, "grid_size": grid, "weight_seed": np.int64(T.SEED_WEIGHTS),
              "batch_size": np.int64(BATCH_SCENES)}
    for k, t in got.items():
        f, i = t.features.numpy(), t.indices.numpy()
        assert t.batch_size == BATCH_SCENES
        assert np.array_equal(i, want[k].indices), k
        err = np.abs(f - want[k].features).max()
        assert err <= 1e-5 * max(1.0, np.abs(f).max()), (k, err)
        assert 1 not in set(np.unique(i[:, 0]).tolist()), k
        print(f"batch {k}: {i.shape[0]} sites x {f.shape[1]} (scenes {sorted(set(i[:, 0].tolist()))}), |reference - oracle| max {err:.2e}")
        arrays[k + "_indices"], arrays[k + "_features"] = i, f.astype(np.float32)
        arrays[k + "_spatial_shape"] = np.array(t.spatial_shape, np.int64)
    arrays["state_checksum"] = np.array([float(np.abs(v).astype(np.float64).sum()) for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, "backbone_forward_batch.npz"), **arrays)


if __name__ == "__main__":
    adapter_fixture()
    batch_fixture()
    print("wrote state_dict_adapter.json, backbone_forward_batch.npz")
