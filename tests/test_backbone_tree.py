"""The backbone's layer graph held to what the REFERENCE's own constructors and forward() produce.

tests/golden/backbone_tree.json and backbone_forward.npz were made by
tests/golden/make_backbone_tree_golden.py: the reference's pcdet/models/backbones_3d/spconv_backbone.py
(:8-67 post_act_block / SparseBasicBlock, :70-181 VoxelBackBone8x, :184-295 VoxelResBackBone8x), imported
from /root/reference with `spconv` aliased exactly as INTEGRATION.md §2 prescribes, constructed for
transfusion_lidar.yaml's arguments, and run once (forward, eval mode) on a small scene with the
convolution primitive backed by the CPU oracle.  Three things are held to it here: the product's
module classes, the oracle's layer table (oracle.RES_BACKBONE8X / backbone_layers, which
backbone_forward executes), and — in the -m gpu set — the product's engines' outputs.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TREE = json.load(open(os.path.join(HERE, "golden", "backbone_tree.json")))
OUTPUTS = ("out", "x_conv1", "x_conv2", "x_conv3", "x_conv4")


def _fixture(plain=False):
    return np.load(os.path.join(HERE, "golden", "backbone_forward_plain.npz" if plain else "backbone_forward.npz"))


def _describe(m):
    from findnpropagate_amd import spconv
    d = {"class": type(m).__name__}
    if isinstance(m, spconv.conv.SparseConvolution):
        d.update(in_channels=m.in_channels, out_channels=m.out_channels, kernel_size=list(m.kernel_size),
                 stride=list(m.stride), padding=list(m.padding), subm=bool(m.subm), indice_key=m.indice_key,
                 bias=m.bias is not None)
    elif isinstance(m, nn.BatchNorm1d):
        d.update(num_features=m.num_features, eps=m.eps, momentum=m.momentum, affine=m.affine,
                 track_running_stats=m.track_running_stats)
    return d


@pytest.mark.parametrize("cls_name", ["VoxelResBackBone8x", "VoxelBackBone8x"])
def test_product_module_tree_is_the_reference_constructors(cls_name):
    from findnpropagate_amd import backbones_3d
    want = TREE[cls_name]
    ctor = want["ctor"]
    net = getattr(backbones_3d, cls_name)(ctor["model_cfg"], ctor["input_channels"], np.array(ctor["grid_size"]))
    got = [{"name": n, **_describe(m)} for n, m in net.named_modules() if n]
    assert [g["name"] for g in got] == [w["name"] for w in want["modules"]]
    for g, w in zip(got, want["modules"]):
        assert g == w, (g, w)
    sd = {k: [list(t.shape), str(t.dtype).replace("torch.", "")] for k, t in net.state_dict().items()}
    assert list(sd) == list(want["state_dict"]) or sorted(sd) == sorted(want["state_dict"])
    assert sd == want["state_dict"]
    assert [int(v) for v in net.sparse_shape] == want["sparse_shape"]
    assert net.num_point_features == want["num_point_features"]
    assert dict(net.backbone_channels) == want["backbone_channels"]


def test_oracle_layer_table_is_the_reference_forward_order():
    """oracle.backbone_layers() — the table oracle.backbone_forward executes — against the order in which the
    reference's forward() reached its convolutions and BatchNorms, and the arguments its constructor gave them."""
    want = TREE["VoxelResBackBone8x"]
    mods = {m["name"]: m for m in want["modules"]}
    calls = want["call_order"]
    convs = [n for n in calls if mods[n]["class"] in ("SubMConv3d", "SparseConv3d")]
    layers = O.backbone_layers(want["ctor"]["input_channels"], 0)
    assert [L["conv"] for L in layers] == convs and len(convs) == 21
    for L in layers:
        m = mods[L["conv"]]
        assert (m["in_channels"], m["out_channels"]) == (L["cin"], L["cout"]), L
        assert m["kernel_size"] == L["kernel"] and m["subm"] == L["subm"] and m["indice_key"] == L["indice_key"], L
        assert m["bias"] is False
        if not L["subm"]:                       # (a SubM layer's stride / padding arguments are not used by spconv)
            assert m["stride"] == L["stride"] and m["padding"] == L["padding"], L
        i = calls.index(L["conv"])
        assert calls[i + 1] == L["bn"], (calls[i:i + 3], L)            # conv -> its BatchNorm, nothing between
        bn = mods[L["bn"]]
        assert bn["class"] == "BatchNorm1d" and bn["num_features"] == L["cout"] and bn["eps"] == 1e-3 and bn["momentum"] == 0.01
        # ReLU follows directly except behind bn2 of a block, where the residual add comes first (then the block's relu)
        assert mods[calls[i + 2]]["class"] == "ReLU"
    # a SparseBasicBlock shares one ReLU module: relu is called twice per block, once after bn1 and once after the add
    assert sum(1 for n in calls if mods[n]["class"] == "ReLU") == 21
    assert want["forward_keys"] == ["encoded_spconv_tensor", "encoded_spconv_tensor_stride", "multi_scale_3d_features",
                                    "multi_scale_3d_strides"]
    assert want["encoded_spconv_tensor_stride"] == 8
    assert want["multi_scale_3d_strides"] == {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}


def _weights(fx, dtype="fp32", plain=False):
    from findnpropagate_amd import synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelBackBone8x, VoxelResBackBone8x
    net = (VoxelBackBone8x if plain else VoxelResBackBone8x)({"USE_BIAS": False, "FNP_DTYPE": dtype}, 5, fx["grid_size"])
    syn.init_backbone_weights(net, int(fx["weight_seed"])).eval()
    chk = np.array([float(np.abs(v.detach().numpy()).astype(np.float64).sum()) for v in net.state_dict().values()])
    assert np.allclose(chk, fx["state_checksum"], rtol=1e-12), "weight recipe drifted from the fixture's"
    return net


def test_oracle_backbone_forward_reproduces_the_reference_forward():
    fx = _fixture()
    net = _weights(fx)
    sd = {k: t.detach().numpy() for k, t in net.state_dict().items()}
    got = O.backbone_forward(sd, fx["voxel_features"], fx["voxel_coords"], 1, net.sparse_shape)
    for k in OUTPUTS:
        assert np.array_equal(got[k].indices, fx[k + "_indices"]), k
        assert got[k].spatial_shape == fx[k + "_spatial_shape"].tolist()
        w = fx[k + "_features"]
        # torch's BatchNorm1d (eval) against the oracle's folded scale / shift: rounding only
        assert np.abs(got[k].features - w).max() <= 1e-5 * max(1.0, np.abs(w).max()), k


def test_oracle_plain_table_is_the_reference_forward_order_and_reproduces_its_forward():
    """oracle.PLAIN_BACKBONE8X (VoxelBackBone8x, spconv_backbone.py:70-181) against the reference's own constructor, the order its
    forward() reached the layers in, and the forward's five outputs on the fixture scene (backbone_forward_plain.npz)."""
    want = TREE["VoxelBackBone8x"]
    mods = {m["name"]: m for m in want["modules"]}
    calls = want["call_order"]
    convs = [n for n in calls if mods[n]["class"] in ("SubMConv3d", "SparseConv3d")]
    layers = O.backbone_layers(want["ctor"]["input_channels"], 0, table=O.PLAIN_BACKBONE8X)
    assert [L["conv"] for L in layers] == convs and len(convs) == 12
    for L in layers:
        m = mods[L["conv"]]
        assert (m["in_channels"], m["out_channels"], m["kernel_size"], m["subm"], m["indice_key"]) == \
               (L["cin"], L["cout"], L["kernel"], L["subm"], L["indice_key"]), L
        if not L["subm"]:
            assert m["stride"] == L["stride"] and m["padding"] == L["padding"], L
        i = calls.index(L["conv"])
        assert calls[i + 1] == L["bn"] and mods[calls[i + 2]]["class"] == "ReLU" and not L["residual"]
    fx = _fixture(plain=True)
    net = _weights(fx, plain=True)
    sd = {k: t.detach().numpy() for k, t in net.state_dict().items()}
    got = O.backbone_forward(sd, fx["voxel_features"], fx["voxel_coords"], 1, net.sparse_shape, table=O.PLAIN_BACKBONE8X)
    for k in OUTPUTS:
        assert np.array_equal(got[k].indices, fx[k + "_indices"]), k
        w = fx[k + "_features"]
        assert np.abs(got[k].features - w).max() <= 1e-5 * max(1.0, np.abs(w).max()), k


def test_integration_shim_runs_as_documented():
    """INTEGRATION.md §2's first code block, executed verbatim in a fresh interpreter, followed by what
    pcdet/utils/spconv_utils.py:3-10 does with the aliased module."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):]
    block = sec[sec.index("```python") + len("```python"):]
    block = block[:block.index("```")]
    assert 'sys.modules["spconv"]' in block and 'sys.modules["spconv.pytorch"]' in block
    tail = (
        "\nimport spconv\n"
        "assert float(spconv.__version__[2:]) >= 2.2\n"
        "spconv.constants.SPCONV_USE_DIRECT_TABLE = False\n"
        "import spconv.pytorch as spconv\n"
        "import torch.nn as nn\n"
        "assert issubclass(spconv.SubMConv3d, spconv.conv.SparseConvolution) and issubclass(spconv.SparseConv3d, spconv.conv.SparseConvolution)\n"
        "assert issubclass(spconv.SparseSequential, nn.Module) and issubclass(spconv.SparseModule, nn.Module)\n"
        "m = spconv.SparseSequential(spconv.SubMConv3d(5, 16, 3, padding=1, bias=False, indice_key='subm1'), nn.BatchNorm1d(16), nn.ReLU())\n"
        "assert list(m.state_dict()) [0] == '0.weight' and tuple(m[0].weight.shape) == (16, 3, 3, 3, 5)\n"
        "import pcdet.ops.iou3d_nms.iou3d_nms_cuda as a, pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda as b\n"
        "assert hasattr(a, 'boxes_overlap_bev_gpu') and hasattr(a, 'nms_gpu') and hasattr(b, 'points_in_boxes_gpu')\n"
        "print('shim ok')\n")
    # `import pcdet.ops...` resolves through sys.modules only if the parents exist: the reference's own package does that;
    # here two empty parents stand for it
    head = ("import sys, types\n"
            "for n in ('pcdet', 'pcdet.ops', 'pcdet.ops.iou3d_nms', 'pcdet.ops.roiaware_pool3d'):\n"
            "    m = types.ModuleType(n); m.__path__ = []; sys.modules[n] = m\n")
    r = subprocess.run([sys.executable, "-c", head + block + tail], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shim ok" in r.stdout, r.stderr[-2000:]


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fused_fp32", "module_fp32", "fused_bf16"])
def test_product_engines_reproduce_the_reference_forward(path):
    """The product on the fixture's inputs against what the reference's forward() produced (over the oracle's
    convolution primitive): site sets equal, f32 features within 1e-4 of the feature scale (the reference runs torch's
    BatchNorm1d, the engines a folded scale / shift), bf16 within its storage precision."""
    fx = _fixture()
    dev = torch.device("cuda", 0)
    net = _weights(fx, "bf16" if path.endswith("bf16") else "fp32").to(dev)
    feats = torch.from_numpy(fx["voxel_features"]).to(dev)
    coords = torch.from_numpy(fx["voxel_coords"]).to(dev)
    bd = {"voxel_features": feats, "voxel_coords": coords, "batch_size": 1}
    with torch.no_grad():
        if path.startswith("module"):
            from findnpropagate_amd.backbones_3d.spconv_backbone import _module_forward
            bd = _module_forward(net, bd)        # the layer-by-layer path INTEGRATION §2's plain shim gives the reference
        else:
            bd = net(bd)
    got = {"out": bd["encoded_spconv_tensor"], **bd["multi_scale_3d_features"]}
    assert bd["encoded_spconv_tensor_stride"] == 8
    tol = 1e-4 if path.endswith("fp32") else 3e-2
    for k in OUTPUTS:
        gi, wi = got[k].indices.cpu().numpy(), fx[k + "_indices"]
        s = got[k].spatial_shape
        assert list(s) == fx[k + "_spatial_shape"].tolist()
        key = lambda i: ((i[:, 1].astype(np.int64) * s[1]) + i[:, 2]) * s[2] + i[:, 3]
        go, wo = np.argsort(key(gi)), np.argsort(key(wi))
        assert np.array_equal(gi[go], wi[wo]), (path, k)
        g, w = got[k].features.float().cpu().numpy()[go], fx[k + "_features"][wo]
        err = np.abs(g - w).max()
        assert err <= tol * max(1.0, np.abs(w).max()), (path, k, err)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
def test_product_plain_backbone_reproduces_the_reference_forward(dtype, tol):
    """VoxelBackBone8x (spconv_backbone.py:70-181; module path) on the fixture's inputs against what the reference's own
    VoxelBackBone8x.forward produced: site sets equal, features within the engine's precision."""
    fx = _fixture(plain=True)
    dev = torch.device("cuda", 0)
    net = _weights(fx, dtype, plain=True).to(dev)
    with torch.no_grad():
        bd = net({"voxel_features": torch.from_numpy(fx["voxel_features"]).to(dev), "voxel_coords": torch.from_numpy(fx["voxel_coords"]).to(dev),
                  "batch_size": 1})
    got = {"out": bd["encoded_spconv_tensor"], **bd["multi_scale_3d_features"]}
    for k in OUTPUTS:
        gi, wi = got[k].indices.cpu().numpy(), fx[k + "_indices"]
        s = got[k].spatial_shape
        assert list(s) == fx[k + "_spatial_shape"].tolist()
        key = lambda i: ((i[:, 1].astype(np.int64) * s[1]) + i[:, 2]) * s[2] + i[:, 3]
        go, wo = np.argsort(key(gi)), np.argsort(key(wi))
        assert np.array_equal(gi[go], wi[wo]), (dtype, k)
        g, w = got[k].features.float().cpu().numpy()[go], fx[k + "_features"][wo]
        assert np.abs(g - w).max() <= tol * max(1.0, np.abs(w).max()), (dtype, k, np.abs(g - w).max())


def test_checkpoints_in_the_spconv_1x_weight_layout_load():
    """detector3d_template.py:401-433 adapts convolution weights saved with spconv 1.x — (kD, kH, kW, Cin, Cout) — to this layout
    (Cout, kD, kH, kW, Cin); a plain load_state_dict() of such a checkpoint must give the same parameters
    (SparseConvolution._load_from_state_dict), and the reference's own find_all_spconv_keys walk (spconv_utils.py:15-29:
    isinstance(child, spconv.conv.SparseConvolution)) finds all 21 weights."""
    from findnpropagate_amd import spconv, synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    grid = np.array([64, 64, 16])
    a = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 3)
    sd = a.state_dict()
    keys = [n + ".weight" for n, m in a.named_modules() if isinstance(m, spconv.conv.SparseConvolution)]
    assert len(keys) == 21 and all(k in sd for k in keys)
    disk = {k: v.clone() for k, v in sd.items()}
    for k in keys:
        disk[k] = sd[k].permute(1, 2, 3, 4, 0).contiguous()          # (Cout, kD, kH, kW, Cin) -> 1.x (kD, kH, kW, Cin, Cout)
        assert disk[k].shape != sd[k].shape
    b = VoxelResBackBone8x({"USE_BIAS": False}, 5, grid)
    b.load_state_dict(disk)
    for k, v in sd.items():
        assert torch.equal(b.state_dict()[k], v), k


# ---- round 5: the reference's checkpoint adapter and a batch with an empty scene ------------------------------------------------
ADAPTER = json.load(open(os.path.join(HERE, "golden", "state_dict_adapter.json")))


def _sha(t):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def test_load_state_dict_equals_the_reference_adapter_on_every_layout():
    """tests/golden/state_dict_adapter.json: the reference's own `_load_state_dict` (detector3d_template.py:401-433) and its
    find_all_spconv_keys (spconv_utils.py:15-29), RUN by tests/golden/make_backbone_extra_golden.py on checkpoints of one seeded
    model in three convolution-weight layouts.  A plain load_state_dict() of the product's class must end with the parameters the
    reference's adapter ended with (sha256 per key) for the 1.x and the implicit-gemm layouts, and must refuse the 2.x native
    layout for the keys the reference refused (neither knows that layout; both raise RuntimeError)."""
    from backbone_recipes import ADAPTER_GRID, ADAPTER_SEED, disk_layouts
    from findnpropagate_amd import spconv, synthetic as syn
    from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
    mk = lambda: VoxelResBackBone8x({"USE_BIAS": False}, 5, np.array(ADAPTER_GRID))
    src = syn.init_backbone_weights(mk(), ADAPTER_SEED)
    keys = sorted(n + ".weight" for n, m in src.named_modules() if isinstance(m, spconv.conv.SparseConvolution))
    assert keys == ADAPTER["spconv_keys"] and len(keys) == 21
    for name, disk in disk_layouts(src.state_dict(), keys).items():
        want = ADAPTER["layouts"][name]
        dst = mk()
        if want.get("raises"):
            with pytest.raises(RuntimeError) as e:
                dst.load_state_dict(disk)
            for k in want["rejected_keys"]:
                assert k in str(e.value), (name, k)
            square = [k for k in keys if k not in want["rejected_keys"]]
            assert not any(("size mismatch for " + k) in str(e.value) for k in square), name
            continue
        dst.load_state_dict(disk)
        got = dst.state_dict()
        assert sorted(got) == sorted(want["sha256"]) == want["updated_keys"]
        for k, h in want["sha256"].items():
            assert _sha(got[k]) == h, (name, k)


def _batch_fixture():
    return np.load(os.path.join(HERE, "golden", "backbone_forward_batch.npz"))


def test_batch_recipe_is_the_fixture_and_the_oracle_reproduces_the_reference_on_it():
    """backbone_forward_batch.npz: the reference's VoxelResBackBone8x.forward on THREE scenes whose middle one is empty."""
    from backbone_recipes import BATCH_SCENES, batch_scene
    fx = _batch_fixture()
    feats, coords, grid = batch_scene()
    assert int(fx["batch_size"]) == BATCH_SCENES == 3 and set(np.unique(coords[:, 0]).tolist()) == {0, 2}
    assert np.array_equal(feats, fx["voxel_features"]) and np.array_equal(coords, fx["voxel_coords"]) and np.array_equal(grid, fx["grid_size"])
    net = _weights(fx)
    sd = {k: t.detach().numpy() for k, t in net.state_dict().items()}
    got = O.backbone_forward(sd, feats, coords, BATCH_SCENES, net.sparse_shape)
    for k in OUTPUTS:
        assert np.array_equal(got[k].indices, fx[k + "_indices"]), k
        w = fx[k + "_features"]
        assert np.abs(got[k].features - w).max() <= 1e-5 * max(1.0, np.abs(w).max()), k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol,path", [("fp32", 1e-4, "engine"), ("fp32", 1e-4, "module"), ("bf16", 3e-2, "engine"), ("bf16x3", 1e-4, "engine")])
def test_product_on_a_batch_with_an_empty_scene_equals_the_reference_forward(dtype, tol, path):
    """the fused engines and the module path (the reference's forward() signature) on the three-scene fixture: same sites in
    every scene — none in the empty one —, features within the engine's bound of the reference's forward"""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    fx = _batch_fixture()
    dev = torch.device("cuda", 0)
    net = _weights(fx, dtype).to(dev)
    B = int(fx["batch_size"])
    bd = {"voxel_features": torch.from_numpy(fx["voxel_features"]).to(dev), "voxel_coords": torch.from_numpy(fx["voxel_coords"]).to(dev), "batch_size": B}
    with torch.no_grad():
        if path == "module":
            from findnpropagate_amd.backbones_3d import spconv_backbone as sb
            bd = sb._module_forward(net, bd)
        else:
            bd = net(bd)
    got = {"out": bd["encoded_spconv_tensor"], **bd["multi_scale_3d_features"]}
    for k in OUTPUTS:
        assert got[k].batch_size == B
        gi, wi = got[k].indices.cpu().numpy(), fx[k + "_indices"]
        s = got[k].spatial_shape
        assert list(s) == fx[k + "_spatial_shape"].tolist()
        key = lambda i: (((i[:, 0].astype(np.int64) * s[0] + i[:, 1]) * s[1]) + i[:, 2]) * s[2] + i[:, 3]
        go, wo = np.argsort(key(gi)), np.argsort(key(wi))
        assert np.array_equal(gi[go], wi[wo]), (dtype, k)
        assert 1 not in set(gi[:, 0].tolist())
        g, w = got[k].features.float().cpu().numpy()[go], fx[k + "_features"][wo]
        assert np.abs(g - w).max() <= tol * max(1.0, np.abs(w).max()), (dtype, path, k, np.abs(g - w).max())
