"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 2): a numpy reading of the tile rulebook that
fnp_tile_rulebook_build / fnp_rulebook_subm_tiled write (include/fnp.h, findnpropagate_amd/csrc/tilerb.h), used by
tests/test_gpu_tile_rulebook.py to check that a tile rulebook says exactly what the (27, cap) int32 table says.

The tile rulebook has no counterpart in the reference (spconv keeps indice pairs, spconv_backbone.py:12-17 call sites): it
is this framework's own restatement of its output-stationary table `nbr`, so the table is what it is checked against."""
import numpy as np

K = 27
ESCAPE = 0xFFFF
GEOMETRY = {  # channels -> (TILE, HALO, OVF, ROW_BYTES)
    32: (256, 32, 256, 64),
    64: (128, 64, 128, 128),
}


def record_bytes(channels):
    tile, _, ovf, _ = GEOMETRY[channels]
    return K * tile * 2 + ovf * 4 + 16


def swizzle(channels, rs):
    """Swizzle bits of the image row at slot rs: byte-address bits 4.. (XOR swizzle)."""
    if channels == 32:
        return ((-(rs >> 2)) & 3) << 4
    return ((rs >> 1) & 7) << 4


def decode(tile_rb, n, channels):
    """tile_rb: uint8 array; returns (nbr (27, n) int64 with -2 where the entry is an escape, escape flags (tiles, TILE//32))."""
    tile, halo, ovf, rowb = GEOMETRY[channels]
    win, zero, rec = tile + 2 * halo, tile + 2 * halo + ovf, record_bytes(channels)
    unit = rowb       # what one image row spans in the entry's unit
    split = 2      # tilerb.h G32::SPLIT (= FNP_TILE32_MB, 2 as shipped) / G64::SPLIT: window rows kept apart by residue
    ntiles = (n + tile - 1) // tile
    recs = np.asarray(tile_rb[: ntiles * rec], dtype=np.uint8).reshape(ntiles, rec)
    codes = recs[:, : K * tile * 2].copy().view(np.uint16).reshape(ntiles, K, tile).astype(np.int64)
    far = recs[:, K * tile * 2: K * tile * 2 + ovf * 4].copy().view(np.int32).reshape(ntiles, ovf).astype(np.int64)
    esc = recs[:, rec - 16: rec - 16 + tile // 32].copy()
    out = np.full((K, ntiles * tile), -1, dtype=np.int64)
    for t in range(ntiles):
        c = codes[t]
        is_esc = c == ESCAPE
        rs = c // unit
        assert np.all(is_esc | ((c % unit) == swizzle(channels, rs))), "entry swizzle bits"
        wlo = max(0, t * tile - halo)
        part = win // split
        d = (rs % part) * split + rs // part                                      # window slot -> window position
        ids = np.where(rs < win, wlo + d, -1)
        in_ovf = (rs >= win) & (rs < zero) & ~is_esc
        ids = np.where(in_ovf, far[t][np.clip(rs - win, 0, ovf - 1)], ids)
        assert np.all(is_esc | (rs <= zero)), "entry beyond the image"
        ids = np.where(is_esc, -2, ids)
        out[:, t * tile:(t + 1) * tile] = ids
    return out[:, :n], esc
