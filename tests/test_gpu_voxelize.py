"""GPU voxeliser + MeanVFE vs the sequential oracle: coords / order / counts / the (M,P,C) block
bit-exact, mean features bit-exact too (same slot-order summation)."""
import numpy as np
import pytest
import torch

from findnpropagate_amd import sparse as S
from findnpropagate_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _run(points_list, cfg_args, cuda, want_voxels=True):
    pts = np.concatenate(points_list, 0) if points_list else np.zeros((0, cfg_args[2]), np.float32)
    off = np.zeros(len(points_list) + 1, np.int32)
    off[1:] = np.cumsum([p.shape[0] for p in points_list])
    cfg = S.make_voxel_cfg(*cfg_args)
    r = S.voxelize(torch.from_numpy(pts).to(cuda), torch.from_numpy(off).to(cuda), len(points_list), cfg,
                   want_voxels=want_voxels)
    n = int(r["n"].item())
    return {k: (r[k][:n].cpu().numpy() if r[k] is not None else None) for k in ("coords", "num_points", "mean", "voxels")}, n


def _oracle(points_list, cfg_args, oracle):
    vs, rg, C, mp, mv = cfg_args
    cs, ns, vs_, ms = [], [], [], []
    for b, p in enumerate(points_list):
        v, c, n = oracle.voxelize(p, vs, rg, mp, mv)
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
        ns.append(n)
        vs_.append(v)
        ms.append(oracle.mean_vfe(v, n))
    return np.concatenate(cs), np.concatenate(ns), np.concatenate(vs_), np.concatenate(ms)


def test_voxelize_nuscenes_scene_bit_exact(cuda, oracle):
    args = (syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    scenes = [syn.make_scene(s) for s in (0, 1, 2)]
    got, n = _run(scenes, args, cuda)
    c, num, vox, mean = _oracle(scenes, args, oracle)
    assert n == c.shape[0] and n > 50000
    assert np.array_equal(got["coords"], c), "voxel coords / first-come order"
    assert np.array_equal(got["num_points"], num)
    assert np.array_equal(got["voxels"], vox)
    assert np.array_equal(got["mean"], mean)


@pytest.mark.parametrize("max_points,max_voxels", [(3, 100000), (10, 300), (1, 77)])
def test_voxelize_overflow_rules(cuda, oracle, rng, max_points, max_voxels):
    """Dense cloud: voxels overflow max_points; max_voxels cuts the first-come list per scene."""
    scenes = []
    for s in range(3):
        p = rng.uniform(-3, 3, size=(5000 + 311 * s, 5)).astype(np.float32)
        p[:, 2] = rng.uniform(-1.2, 1.2, size=p.shape[0])
        p[::11, 1] = -50.0   # out of range
        scenes.append(p)
    args = ([0.25, 0.25, 0.5], [-2.5, -2.5, -1.0, 2.5, 2.5, 1.0], 5, max_points, max_voxels)
    got, n = _run(scenes, args, cuda)
    c, num, vox, mean = _oracle(scenes, args, oracle)
    assert n == c.shape[0]
    assert np.array_equal(got["coords"], c) and np.array_equal(got["num_points"], num)
    assert np.array_equal(got["voxels"], vox) and np.array_equal(got["mean"], mean)
    assert num.max() <= max_points and (max_points > 3 or num.max() == max_points)


def test_voxelize_shuffled_points_follow_point_order(cuda, oracle, rng):
    """The reference shuffles points before voxelising (data_processor.py:96-106): order matters."""
    p = syn.make_scene(5)
    p = p[rng.permutation(p.shape[0])]
    args = (syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    got, n = _run([p], args, cuda)
    c, num, vox, mean = _oracle([p], args, oracle)
    assert np.array_equal(got["coords"], c) and np.array_equal(got["voxels"], vox)


def test_voxelize_edge_inputs(cuda, oracle):
    args = ([0.1, 0.1, 0.1], [0, 0, 0, 1, 1, 1], 5, 10, 100)
    got, n = _run([np.zeros((0, 5), np.float32)], args, cuda)
    assert n == 0
    p = np.array([[0.55, 0.25, 0.95, 7, 0], [1.0, 0.5, 0.5, 0, 0], [-1e-9, 0.5, 0.5, 0, 0], [0.55, 0.25, 0.95, 9, 1]], np.float32)
    got, n = _run([p], args, cuda)
    assert n == 1 and got["coords"].tolist() == [[0, 9, 2, 5]] and got["num_points"].tolist() == [2]
    assert got["mean"][0, 3] == 8.0
    # a scene with no valid point between two non-empty scenes
    got, n = _run([p, np.full((3, 5), 50.0, np.float32), p], args, cuda)
    assert got["coords"].tolist() == [[0, 9, 2, 5], [2, 9, 2, 5]]


def test_voxel_generator_wrapper_dropin(cuda, oracle):
    """VoxelGeneratorWrapper with the reference's signature (data_processor.py:17-62)."""
    from findnpropagate_amd.processor import VoxelGeneratorWrapper

    p = syn.make_scene(3)
    v, c, n = oracle.voxelize(p, syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 10, 120000)
    for device in ("cuda", None):      # the device generator and the host generator (dataloader workers) agree bit for bit
        g = VoxelGeneratorWrapper(vsize_xyz=syn.VOXEL_SIZE, coors_range_xyz=syn.POINT_CLOUD_RANGE, num_point_features=5,
                                  max_num_points_per_voxel=10, max_num_voxels=120000, device=device)
        voxels, coords, num = g.generate(p)
        assert coords.dtype == np.int32 and coords.shape[1] == 3
        assert np.array_equal(coords, c) and np.array_equal(num, n) and np.array_equal(voxels, v)


def test_rerun_is_deterministic(cuda):
    args = (syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
    scenes = [syn.make_scene(7)]
    a, _ = _run(scenes, args, cuda)
    b, _ = _run(scenes, args, cuda)
    for k in a:
        assert np.array_equal(a[k], b[k])


def test_fused_mean_equals_the_reference_meanvfe_fixture(cuda, oracle):
    """tests/golden/meanvfe_golden.npz: MeanVFE.forward of the reference's own class (mean_vfe.py:14-31, torch CPU) on the voxel
    block of a scene whose time column is that of a ten-sweep aggregation.  The voxeliser's fused mean (vox_emit_kernel) must be
    that tensor bit for bit — which it is only in torch's summation order (163 rows of this fixture differ in a slot-order sum)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meanvfe_golden.npz"))
    n_scene = int(d["n_scene_voxels"])
    args = (syn.VOXEL_SIZE, [float(v) for v in d["range"]], 5, 10, 160000)
    got, n = _run([d["points"]], args, cuda)
    assert n == n_scene
    assert np.array_equal(got["voxels"], d["voxels"][:n_scene]) and np.array_equal(got["num_points"], d["num_points"][:n_scene])
    assert np.array_equal(got["mean"], d["mean"][:n_scene])
    seq = np.zeros((n_scene, 5), np.float32)
    for p in range(10):
        seq = (seq + d["voxels"][:n_scene, p]).astype(np.float32)
    assert (seq / np.maximum(d["num_points"][:n_scene], 1)[:, None].astype(np.float32) != got["mean"]).any(), "the fixture tells the two orders apart"


@pytest.mark.parametrize("C,max_points", [(4, 10), (5, 5), (6, 10), (7, 12), (5, 20), (4, 35), (3, 10), (8, 10)])
def test_fused_mean_in_torchs_summation_order_for_other_shapes(cuda, oracle, rng, C, max_points):
    """The generic path of vox_emit_kernel (any feature count, max_points >= 16: the cascade's accumulator levels) against
    oracle.mean_vfe, which tests/test_meanvfe_golden.py holds to torch's own CPU sum(dim=1) for the same shapes."""
    p = rng.uniform(-3, 3, size=(20000, C)).astype(np.float32)
    p[:, 2] = rng.uniform(-1.2, 1.2, size=p.shape[0])
    p[:, 3:] = (rng.normal(size=(p.shape[0], C - 3)) * rng.choice([1e-3, 1.0, 100.0], size=(p.shape[0], C - 3))).astype(np.float32)
    args = ([0.5, 0.5, 0.5], [-2.5, -2.5, -1.0, 2.5, 2.5, 1.0], C, max_points, 100000)
    got, n = _run([p], args, cuda)
    c, num, vox, mean = _oracle([p], args, oracle)
    assert n == c.shape[0] and num.max() == max_points
    assert np.array_equal(got["voxels"], vox) and np.array_equal(got["mean"], mean)


@pytest.mark.parametrize("n", [4095, 4096, 4097, 12288, 65535, 65536, 65537])
def test_first_point_scan_forms_at_their_size_boundaries(cuda, oracle, rng, n):
    """The first-point flags are scanned in ONE launch up to 16 tiles of 4,096 points (scan.hip small_scan_kernel: a workgroup sums
    what lies in front of its tile itself) and by the three-launch tile scan above: same voxels, same first-come order, at the
    tile boundaries of the first and the size boundary between the two."""
    n0 = n // 3
    scenes = []
    for m in (n0, n - n0):
        p = rng.uniform(-6, 6, size=(m, 5)).astype(np.float32)
        p[:, 2] = rng.uniform(-1.0, 1.0, size=m)
        p[::13, 0] = 99.0   # out of range
        scenes.append(p)
    args = ([0.1, 0.1, 0.25], [-5.0, -5.0, -1.0, 5.0, 5.0, 1.0], 5, 4, 60000)
    got, nv = _run(scenes, args, cuda)
    c, num, vox, mean = _oracle(scenes, args, oracle)
    assert nv == c.shape[0] and nv > n // 4
    assert np.array_equal(got["coords"], c) and np.array_equal(got["num_points"], num)
    assert np.array_equal(got["voxels"], vox) and np.array_equal(got["mean"], mean)
