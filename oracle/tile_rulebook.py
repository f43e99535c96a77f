"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 2): a numpy reading of the tile rulebook that
fnp_tile_rulebook_build / fnp_rulebook_subm_tiled write (include/fnp.h, findnpropagate_amd/csrc/tilerb.cuh), used by
tests/test_gpu_tile_rulebook.py to check that a tile rulebook says exactly what the (27, cap) int32 table says.

The tile rulebook has no counterpart in the reference (spconv keeps indice pairs, spconv_backbone.py:12-17 call sites): it
is this framework's own restatement of its output-stationary table `nbr`, so the table is what it is checked against."""
import numpy as np

K = 27
ESCAPE = 0xFFFF
GEOMETRY = {  # channels -> (TILE, HALO, OVF, ROW_BYTES)
    32: (256, 32, 256, 64),
    64: (256, 32, 128, 128),
}
# image slots (tilerb.cuh G32 / G64): 32 channels — window rows, even ones first, then odd ones; overflow rows; the row of zeros.
# 64 channels — four window quarters (row position mod 4) of 96 slots, each starting 8 slots in; overflow rows from slot 384;
# the row of zeros is slot 0.


def record_bytes(channels):
    tile, _, ovf, _ = GEOMETRY[channels]
    return K * tile * 2 + ovf * 4 + 16


def swizzle(channels, rs):
    """Swizzle bits (entry bits 4..) of the image row at slot rs."""
    if channels == 32:
        return ((-(rs >> 2)) & 3) << 4
    return ((rs >> 1) & 7) << 4


def decode(tile_rb, n, channels):
    """tile_rb: uint8 array; returns (nbr (27, n) int64 with -2 where the entry is an escape, escape flags (tiles, TILE//32))."""
    tile, halo, ovf, rowb = GEOMETRY[channels]
    win, rec = tile + 2 * halo, record_bytes(channels)
    if channels == 32:
        ovf_base, zero, nslots = win, win + ovf, win + ovf + 1
        win_pos = np.full(nslots, -1, dtype=np.int64)          # slot -> window position
        d = np.arange(win)
        win_pos[(d & 1) * (win // 2) + (d >> 1)] = d
    else:
        ovf_base, zero, nslots = 384, 0, 512
        win_pos = np.full(nslots, -1, dtype=np.int64)
        d = np.arange(win)
        win_pos[(d & 3) * 96 + 8 + (d >> 2)] = d
    ntiles = (n + tile - 1) // tile
    recs = np.asarray(tile_rb[: ntiles * rec], dtype=np.uint8).reshape(ntiles, rec)
    codes = recs[:, : K * tile * 2].copy().view(np.uint16).reshape(ntiles, K, tile).astype(np.int64)
    far = recs[:, K * tile * 2: K * tile * 2 + ovf * 4].copy().view(np.int32).reshape(ntiles, ovf).astype(np.int64)
    esc = recs[:, rec - 16: rec - 16 + tile // 32].copy()
    out = np.full((K, ntiles * tile), -1, dtype=np.int64)
    for t in range(ntiles):
        c = codes[t]
        is_esc = c == ESCAPE
        rs = c // rowb
        assert np.all(is_esc | ((c % rowb) == swizzle(channels, rs))), "entry swizzle bits"
        wlo = max(0, t * tile - halo)
        assert np.all(is_esc | (rs < nslots)), "entry beyond the image"
        rsc = np.clip(rs, 0, nslots - 1)
        d = win_pos[rsc]
        in_ovf = (rsc >= ovf_base) & (rsc < ovf_base + ovf) & ~is_esc
        assert np.all(is_esc | in_ovf | (rsc == zero) | (d >= 0)), "entry names an unused image slot"
        ids = np.where(d >= 0, wlo + d, -1)
        ids = np.where(in_ovf, far[t][np.clip(rsc - ovf_base, 0, ovf - 1)], ids)
        ids = np.where(rsc == zero, -1, ids)
        ids = np.where(is_esc, -2, ids)
        out[:, t * tile:(t + 1) * tile] = ids
    return out[:, :n], esc
