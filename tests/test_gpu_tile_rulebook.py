"""The tile rulebook (fnp_tile_rulebook_build, fnp_rulebook_subm_tiled) against the int32 table it restates, decoded on
the host by oracle/tile_rulebook.py: every entry names the row the table names (window row, overflow row through the far-row
list, no neighbour), escapes are flagged per 32-row group and only occur where a tile has more distinct far rows than
overflow rows.  Bit-exact (integer work)."""
import numpy as np
import pytest
import torch

from findnpropagate_amd import lib as _l
from findnpropagate_amd import sparse as S
from oracle import tile_rulebook as TR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _sites(rng, B, shape, n, order):
    cells = B * shape[0] * shape[1] * shape[2]
    lin = np.sort(rng.choice(cells, size=n, replace=False)) if order == "sorted" else rng.choice(cells, size=n, replace=False)
    b, rem = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32)


@pytest.mark.parametrize("channels", [32, 64])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("n,order", [(1, "sorted"), (255, "sorted"), (3000, "random"), (20000, "sorted"), (20000, "random")])
def test_tile_rulebook_restates_the_table(cuda, n, order, fused, channels):
    ch = channels
    rng = np.random.default_rng(n + (7 if fused else 0) + ch)
    aborts0 = _l.load().fnp_spconv_tiled_aborts()   # (library-wide counter: another test raises it on purpose)
    B, shape = 2, [11, 60, 61]
    idx = torch.from_numpy(_sites(rng, B, shape, n, order)).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    grid = S.build_grid(idx, n_dev, B, shape)
    rb = S.rulebook_subm(idx, n_dev, grid, 3, tile_channels=ch if fused else None)
    tile_rb = S.tile_rulebook(rb, n_dev, ch)
    assert tile_rb.numel() == TR.record_bytes(channels) * ((rb.cap_out + TR.GEOMETRY[channels][0] - 1) // TR.GEOMETRY[channels][0])
    got, esc = TR.decode(tile_rb.cpu().numpy(), n, channels)
    want = rb.nbr[:, :n].cpu().numpy().astype(np.int64)
    escaped = got == -2
    assert np.array_equal(np.where(escaped, want, got), want)
    tile, _, ovf, _ = TR.GEOMETRY[channels]
    # escape flags: exactly the 32-row groups that hold an escape entry
    pad = np.zeros((TR.K, esc.shape[0] * tile), dtype=bool)
    pad[:, :n] = escaped
    groups = pad.reshape(TR.K, esc.shape[0], tile // 32, 32).any(axis=(0, 3))
    assert np.array_equal(groups, esc.astype(bool))
    # an escape only where the tile's distinct far rows exceed (or crowd) its overflow rows
    halo = TR.GEOMETRY[channels][1]
    for t in np.nonzero(groups.any(axis=1))[0]:
        w = want[:, t * tile:min(n, (t + 1) * tile)]
        wlo = max(0, t * tile - halo)
        far = np.unique(w[(w >= 0) & ((w < wlo) | (w >= wlo + tile + 2 * halo))])
        assert far.size > ovf // 2
    assert _l.load().fnp_spconv_tiled_aborts() == aborts0


def test_tiled_entry_points_refuse_what_they_do_not_cover(cuda):
    """fnp_spconv_forward_tiled / fnp_tile_rulebook_build return FNP_ERR_ARG (no launch) for shapes and buffers outside
    their contract; sparse.conv_forward then takes the gather kernel by itself."""
    L = _l.load()
    n, C = 300, 32
    rng = np.random.default_rng(5)
    idx = torch.from_numpy(_sites(rng, 1, [5, 20, 21], n, "sorted")).to(cuda)
    n_dev = S.device_scalar(n, cuda)
    rb = S.rulebook_subm(idx, n_dev, S.build_grid(idx, n_dev, 1, [5, 20, 21]), 3)
    x = torch.randn((n, C), device=cuda).bfloat16()
    w = torch.randn((27, C, C), device=cuda).bfloat16()
    y = torch.empty_like(x)
    t = S.tile_rulebook(rb, n_dev, C)
    args = lambda **kw: dict(dict(feat=_l.ptr(x), dtype=_l.FNP_BF16, rows=n, w=_l.ptr(w), t=_l.ptr(t), nbr=_l.ptr(rb.nbr), stride=rb.nbr.shape[1],
                                   n=_l.ptr(n_dev), cap=rb.cap_out, out=_l.ptr(y), cin=C, cout=C), **kw)

    def call(a):
        return L.fnp_spconv_forward_tiled(a["feat"], a["dtype"], a["rows"], a["w"], a["t"], a["nbr"], a["stride"], a["n"], a["cap"], a["out"],
                                          None, None, None, 0, a["cin"], a["cout"], _l.stream())
    assert call(args()) == 0
    assert call(args(cin=16, cout=16)) == -1            # channels
    assert call(args(cin=32, cout=64)) == -1
    assert call(args(dtype=_l.FNP_F32)) == -1           # f32 features
    assert call(args(t=_l.ptr(t) + 4)) == -1            # misaligned tile rulebook
    assert call(args(t=None)) == -1
    assert call(args(stride=rb.cap_out - 1)) == -1      # table narrower than the capacity
    assert L.fnp_tile_rulebook_bytes(rb.cap_out, 48) == 0
    assert L.fnp_tile_rulebook_build(_l.ptr(rb.nbr), rb.nbr.shape[1], 27, _l.ptr(n_dev), rb.cap_out, 48, _l.ptr(t), _l.stream()) == -1
    assert L.fnp_tile_rulebook_build(_l.ptr(rb.nbr), rb.nbr.shape[1], 8, _l.ptr(n_dev), rb.cap_out, 32, _l.ptr(t), _l.stream()) == -1
    torch.cuda.synchronize()
    # the Python layer falls back to the gather kernel for a layer the tiled kernels do not cover, whatever `tile` says
    w16 = torch.randn((27, 16, 16), device=cuda).bfloat16()
    x16 = torch.randn((n, 16), device=cuda).bfloat16()
    a = S.conv_forward(x16, w16, rb, n_dev, tile=True)
    b = S.conv_forward(x16, w16, rb, n_dev, tile=False)
    assert torch.equal(a[:n], b[:n])
