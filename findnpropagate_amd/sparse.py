"""Functional host layer over the C ABI (include/fnp.h) for voxelisation, rank grids, rulebooks
and sparse convolution.  torch is used for device memory and streams only; every function
enqueues HIP kernels from libfnp_hip.so on the current stream and never synchronises unless its
docstring says so.

Data layout in HBM:
  features  (cap, C)   row-major, bf16 or f32            rows >= n are undefined
  indices   (cap, 4)   int32 [b, z, y, x]
  n         (1,)       int32 device scalar: number of valid rows
  grid      bits (nblk,) u64 occupancy, base (nblk,) u32 popcount prefix, summary (nblk/64,) u64,
            perm (cap,) int32|None
  rulebook  nbr (K, cap) int32: input row per (kernel offset, output row) or -1
"""
import ctypes
import os
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import lib as _l


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


@dataclass
class RankGrid:
    bits: torch.Tensor            # (nblk,) int64 view of u64 occupancy words
    base: torch.Tensor            # (nblk,) int32 view of u32 popcount prefixes
    summary: torch.Tensor         # (nsum,) int64: one bit per block
    perm: Optional[torch.Tensor]  # (cap,) int32 or None (rows already in rank order)
    batch_size: int
    shape: List[int]              # [D, H, W]
    counters: Optional[torch.Tensor] = None   # (fnp_rankgrid_counter_words,) int32, zero between builds: the counted marks (fnp.h)

    def c(self, with_perm=True):
        """struct fnp_rankgrid for the ABI."""
        g = _l.RankGridC()
        g.B, g.D, g.H, g.W = self.batch_size, self.shape[0], self.shape[1], self.shape[2]
        g.bits, g.base, g.summary = self.bits.data_ptr(), self.base.data_ptr(), self.summary.data_ptr()
        g.perm = self.perm.data_ptr() if (with_perm and self.perm is not None) else None
        g.counters = self.counters.data_ptr() if (self.counters is not None and COUNTED_MARKS) else None
        return g

    def zero_(self):
        self.bits.zero_()
        self.summary.zero_()
        if self.counters is not None:
            self.counters.zero_()


@dataclass
class Rulebook:
    nbr: torch.Tensor             # (K, cap_out) int32
    K: int
    cap_out: int
    geom: _l.ConvGeom
    out_indices: Optional[torch.Tensor] = None   # strided only
    out_n: Optional[torch.Tensor] = None
    out_grid: Optional[RankGrid] = None
    out_shape: Optional[List[int]] = None
    in_grid: Optional[RankGrid] = None           # strided, nbr is None: the grid the convolution looks its inputs up in


# counted marks (round 6, fnp.h fnp_rankgrid.counters): the marking kernels count the cells they set and the rank prefix is one
# launch instead of three.  FNP_COUNTED_MARKS=0: the three-launch prefix (development A/B; same grids, same ranks)
COUNTED_MARKS = os.environ.get("FNP_COUNTED_MARKS", "1") != "0"


def num_blocks(batch_size, shape):
    return int(_l.load().fnp_rankgrid_num_blocks(batch_size, *shape))


def alloc_grid(batch_size, shape, device, with_perm_cap=None):
    """Zeroed occupancy words + uninitialised prefix (+ perm) for a (B, D, H, W) grid."""
    nblk = num_blocks(batch_size, shape)
    bits = torch.zeros((nblk,), dtype=torch.int64, device=device)
    base = torch.empty((nblk,), dtype=torch.int32, device=device)
    summary = torch.zeros(((nblk + 63) // 64,), dtype=torch.int64, device=device)
    perm = torch.empty((with_perm_cap,), dtype=torch.int32, device=device) if with_perm_cap else None
    counters = torch.zeros((int(_l.load().fnp_rankgrid_counter_words(batch_size, *shape)),), dtype=torch.int32, device=device)
    return RankGrid(bits, base, summary, perm, batch_size, [int(v) for v in shape], counters)


def make_geom(ksize, stride, padding, in_shape, out_shape=None):
    g = _l.ConvGeom()
    k, s, p = _triple(ksize), _triple(stride), _triple(padding)
    if out_shape is None:
        out_shape = [(in_shape[d] + 2 * p[d] - k[d]) // s[d] + 1 for d in range(3)]
    for d in range(3):
        g.ksize[d], g.stride[d], g.padding[d] = k[d], s[d], p[d]
        g.in_shape[d], g.out_shape[d] = int(in_shape[d]), int(out_shape[d])
    return g, [int(x) for x in out_shape]


def device_scalar(value, device):
    return torch.full((1,), int(value), dtype=torch.int32, device=device)


# --------------------------------------------------------------------------------- voxelise
def make_voxel_cfg(voxel_size, point_cloud_range, num_features, max_points, max_voxels):
    """grid = round((max - min) / voxel_size) as in data_processor.py:257-258."""
    import numpy as np

    cfg = _l.VoxelCfg()
    rng = np.asarray(point_cloud_range, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    grid = np.round((rng[3:6] - rng[0:3]) / vs).astype(np.int64)
    for d in range(3):
        cfg.range_min[d] = float(rng[d])
        cfg.voxel_size[d] = float(vs[d])
        cfg.grid[d] = int(grid[d])
    cfg.num_features = int(num_features)
    cfg.max_points = int(max_points)
    cfg.max_voxels = int(max_voxels)
    return cfg


def concurrent_streams(device, n, beside=(), tries=24, report=False):
    """n HIP streams that really run BESIDE each other and beside the streams in `beside`.

    HIP spreads its streams over a few hardware queues — 4 (GPU_MAX_HW_QUEUES) — in the order of their first use, torch hands its
    streams out of a pool, and two streams of one hardware queue run strictly one after the other: a batch's index chain behind
    another batch's convolutions.  Nothing says so: the same pipeline measured 12.2 k or 13.4 k scenes/s at 128 scenes, 1.7 k or 3.0 k
    frames/s with two one-scene frames in flight, by what the process had done with streams before (in a fresh process the first
    three pool streams share the queue of the default stream).  So candidate streams are TESTED: a run of chip-filling kernels
    (48 in-place adds over 128 MB, ~1.8 ms) goes to one stream, a one-word fill and an event to the other; the event completes at once unless the
    two share a queue (0.02-0.3 ms against the whole run: tools/probe/queue_map.py sorts 14 pool streams into exactly four classes
    this way, round robin in pool order).  A spin kernel of one thread does NOT show it.  Streams that fail against any already chosen
    one are dropped (back into torch's pool) and the next is tried; after `tries` candidates the rest is filled with untested ones
    (report=True: returns (streams, whether every one of them passed))."""
    import time
    device = torch.device(device)
    chosen, rejected = [], []
    if device.type != "cuda" or n <= 0:
        return (chosen, True) if report else chosen
    with torch.cuda.device(device):
        # (elementwise KERNELS, never zero_(): that is hipMemsetAsync for a large tensor, and on this stack an eager memset changes what
        #  the memset NODES of graphs captured earlier do when they are replayed — DESIGN.md section 2; with a 256 MB one in between,
        #  the replay of a graph that torch had given a 4-byte memset node segfaulted)
        big = torch.empty((32 << 20,), dtype=torch.int32, device=device)     # 128 MB: an add over it holds every CU for ~60 us
        word = torch.ones((1,), dtype=torch.int32, device=device)
        big.fill_(1)
        torch.cuda.synchronize(device)

        def shares_queue(a, b):
            """does work on b wait for the work on a (one hardware queue)?"""
            if a == b:
                return True
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            with torch.cuda.stream(a):
                for _ in range(48):      # (~1.8 ms: well above the ~0.2 ms the fill + event below cost the host on a free queue)
                    big.add_(1)
            ev = torch.cuda.Event()
            with torch.cuda.stream(b):
                word.fill_(1)
            ev.record(b)
            ev.synchronize()
            waited = time.perf_counter() - t0
            torch.cuda.synchronize(device)
            return waited > 0.5 * (time.perf_counter() - t0)

        others = [s for s in beside if s is not None]
        for _ in range(max(tries, n)):
            if len(chosen) == n:
                break
            c = torch.cuda.Stream(device)
            if any(shares_queue(o, c) for o in others + chosen):
                rejected.append(c)
                continue
            chosen.append(c)
        complete = len(chosen) == n
        while len(chosen) < n:      # (more streams asked for than the card has queues to give: take what there is)
            chosen.append(rejected.pop(0) if rejected else torch.cuda.Stream(device))
        del big
    return (chosen, complete) if report else chosen


def stage_points(points, batch_offsets, dst_points, dst_offsets, n_prev, pad):
    """one launch: dst_points[:n] = points, dst_points[n:n_prev] = pad, dst_offsets[:] = batch_offsets (fnp_stage_points: a frame into the
    static inputs of a captured forward).  Everything on the device; the caller's stream."""
    L = _l.load()
    _l.require_device(points, batch_offsets, dst_points, dst_offsets)
    assert points.dtype == torch.float32 and points.dim() == 2 and points.is_contiguous() and dst_points.is_contiguous()
    assert dst_points.dtype == torch.float32 and dst_points.shape[1] == points.shape[1] and dst_points.shape[0] >= max(points.shape[0], n_prev)
    assert batch_offsets.dtype == torch.int32 and dst_offsets.dtype == torch.int32 and dst_offsets.numel() == batch_offsets.numel()
    assert batch_offsets.is_contiguous() and dst_offsets.is_contiguous()
    _l.check(L.fnp_stage_points(_l.ptr(points) if points.shape[0] else None, points.shape[0], int(n_prev), points.shape[1], float(pad), _l.ptr(dst_points),
                                _l.ptr(batch_offsets), batch_offsets.numel(), _l.ptr(dst_offsets), _l.stream()), "fnp_stage_points")


def voxelize(points, batch_offsets, batch_size, cfg, grid=None, want_voxels=False, workspace=None):
    """points (N,C) f32 device, batch_offsets (B+1,) int32 device.

    Returns dict(coords (N,4) i32, num_points (N,), mean (N,C) f32, voxels|None, n (1,) i32 device,
    grid RankGrid with perm, n_cells (1,) i32 device).  Rows >= n are undefined, except that coords rows
    [n, n_cells) list the cells of voxels dropped by max_voxels (for clear_grid).  No host sync.
    """
    L = _l.load()
    _l.require_device(points, batch_offsets)
    assert points.dtype == torch.float32 and points.dim() == 2 and points.is_contiguous()
    assert batch_offsets.dtype == torch.int32 and batch_offsets.numel() == batch_size + 1
    n, C = points.shape
    assert C == cfg.num_features
    dev = points.device
    cap = max(n, 1)
    if grid is None:
        grid = alloc_grid(batch_size, [cfg.grid[2], cfg.grid[1], cfg.grid[0]], dev, with_perm_cap=cap)
    elif grid.perm is None or grid.perm.numel() < cap:
        grid.perm = torch.empty((cap,), dtype=torch.int32, device=dev)
    assert grid.batch_size == batch_size
    gc = grid.c()   # the rank grid may be larger than the voxel grid (sparse_shape has z + 1)
    ws_bytes = int(L.fnp_voxelize_workspace_bytes(n, cfg, gc))
    _l.check(min(ws_bytes, 0), "fnp_voxelize_workspace_bytes")
    if workspace is None or workspace.numel() < ws_bytes:
        workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num_points = torch.empty((cap,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap, C), dtype=torch.float32, device=dev)
    voxels = torch.empty((cap, cfg.max_points, C), dtype=torch.float32, device=dev) if want_voxels else None
    n_vox = torch.empty((2,), dtype=torch.int32, device=dev)    # always written by fnp_voxelize: voxels, cells
    rc = L.fnp_voxelize(_l.ptr(points), n, _l.ptr(batch_offsets), cfg, gc,
                        _l.ptr(workspace), workspace.numel(),
                        _l.ptr(coords), _l.ptr(num_points), _l.ptr(mean), _l.ptr(voxels), _l.ptr(n_vox),
                        n_vox.data_ptr() + 4, cap, _l.stream())
    _l.check(rc, "fnp_voxelize")
    return dict(coords=coords, num_points=num_points, mean=mean, voxels=voxels, n=n_vox[:1], n_cells=n_vox[1:],
                grid=grid, workspace=workspace, cap=cap)


# --------------------------------------------------------------------------------- grids
def build_grid(indices, n_dev, batch_size, shape, keep_order=True, grid=None):
    """Index an explicit (cap,4) coordinate list.  perm maps rank -> row (keep_order=True)."""
    L = _l.load()
    _l.require_device(indices, n_dev)
    assert indices.dtype == torch.int32 and indices.is_contiguous()
    cap = max(indices.shape[0], 1)
    dev = indices.device
    if grid is None:
        grid = alloc_grid(batch_size, shape, dev, with_perm_cap=cap if keep_order else None)
    assert grid.batch_size == batch_size and list(grid.shape) == [int(v) for v in shape]
    if keep_order and (grid.perm is None or grid.perm.numel() < cap):
        grid.perm = torch.empty((cap,), dtype=torch.int32, device=dev)
    ws = torch.empty((int(L.fnp_rankgrid_workspace_bytes(batch_size, *grid.shape)) + 256,), dtype=torch.uint8, device=dev)
    rc = L.fnp_rankgrid_build(_l.ptr(indices), _l.ptr(n_dev), cap, grid.c(with_perm=keep_order), _l.ptr(ws), ws.numel(),
                              _l.stream())
    _l.check(rc, "fnp_rankgrid_build")
    return grid


def clear_grid(grid, indices, n_dev):
    """Sparse clear of the occupancy words touched by `indices` (O(rows))."""
    L = _l.load()
    rc = L.fnp_rankgrid_clear(_l.ptr(indices), _l.ptr(n_dev), max(indices.shape[0], 1), grid.c(), _l.stream())
    _l.check(rc, "fnp_rankgrid_clear")


# the clear of the fused engine's persistent grids walks their summary level (fnp_rankgrid_clear_summary: 79 -> ~25 us at 128 scenes,
# and nothing a coordinate list could miss); FNP_CLEAR_ROWS=1: the row form of rounds 1-5 (development A/B)
CLEAR_BY_SUMMARY = os.environ.get("FNP_CLEAR_ROWS", "0") != "1"


def clear_grids(jobs):
    """clear_grid for several grids in ONE launch.  jobs: list of (grid, indices, n_dev), at most 8."""
    L = _l.load()
    k = len(jobs)
    if CLEAR_BY_SUMMARY:
        grids = (_l.RankGridC * k)(*[g.c() for g, _, _ in jobs])
        rc = L.fnp_rankgrid_clear_summary(k, ctypes.cast(grids, ctypes.c_void_p), _l.stream())
        _l.check(rc, "fnp_rankgrid_clear_summary")
        return
    P = ctypes.c_void_p * k
    coords = P(*[_l.ptr(idx) for _, idx, _ in jobs])
    rows = P(*[_l.ptr(n) for _, _, n in jobs])
    caps = (ctypes.c_int * k)(*[max(idx.shape[0], 1) for _, idx, _ in jobs])
    grids = (_l.RankGridC * k)(*[g.c() for g, _, _ in jobs])
    rc = L.fnp_rankgrid_clear_multi(k, ctypes.cast(coords, ctypes.c_void_p), ctypes.cast(rows, ctypes.c_void_p),
                                    ctypes.cast(caps, ctypes.c_void_p), ctypes.cast(grids, ctypes.c_void_p), _l.stream())
    _l.check(rc, "fnp_rankgrid_clear_multi")


# --------------------------------------------------------------------------------- rulebooks
def rulebook_subm(indices, n_dev, grid, ksize, tile_channels=None, masks=False, lean_table=False, mark_next=None, esc_counter=None):
    """tile_channels (32 or 64, 3x3x3 only): also write the tile rulebook the `tile_channels`-channel layers of this rulebook
    run on (conv_forward's tiled path finds it with the rulebook), in the same pass.
    masks (3x3x3 only): also write the per-row neighbour masks (`rb._rowmask`) the class sort of the 128-channel layers starts from.
    lean_table (with tile_channels): the int32 table gets only the rows the tiled convolutions can ask it for (tiles with escape
    entries) — for a caller that runs nothing but conv_forward's tiled path on this rulebook (`rb._lean` is set: conv_forward
    refuses any other kernel on it).
    esc_counter (with lean_table): a (1,) int32 device counter that receives += the number of 32-row groups with an escape entry.
    mark_next (with lean_table or masks): (out_grid, ksize, stride, padding) of the strided convolution that consumes these rows —
    the kernel marks its output sites in out_grid (all zero) on the way; rulebook_strided(..., premarked=True) then skips its
    own marking launch.  `rb._marked_next` says whether it was done."""
    L = _l.load()
    cap = max(indices.shape[0], 1)
    geom, _ = make_geom(ksize, 1, [k // 2 for k in _triple(ksize)], grid.shape, grid.shape)
    K = geom.ksize[0] * geom.ksize[1] * geom.ksize[2]
    nbr = torch.empty((K, cap), dtype=torch.int32, device=indices.device)
    mg, mgeom = None, None
    if mark_next is not None:
        out_grid, mk, ms, mp = mark_next
        mgeom, _ = make_geom(mk, ms, mp, grid.shape)
        if all((mgeom.ksize[d] + mgeom.stride[d] - 1) // mgeom.stride[d] <= 2 for d in range(3)):
            mg = out_grid.c(with_perm=False)
        else:
            mgeom = None
    if tile_channels and K == 27 and os.environ.get("FNP_TILE_FUSED", "1") != "0":   # (0: development A/B — the stand-alone build on first use)
        t = torch.empty((L.fnp_tile_rulebook_bytes(cap, tile_channels),), dtype=torch.uint8, device=indices.device)
        if lean_table:
            rc = L.fnp_rulebook_subm_tiled_lean(_l.ptr(indices), _l.ptr(n_dev), cap, geom, grid.c(), _l.ptr(nbr), tile_channels, _l.ptr(t), mg, mgeom,
                                                _l.ptr(esc_counter), _l.stream())
        else:
            rc = L.fnp_rulebook_subm_tiled(_l.ptr(indices), _l.ptr(n_dev), cap, geom, grid.c(), _l.ptr(nbr), tile_channels, _l.ptr(t), _l.stream())
        _l.check(rc, "fnp_rulebook_subm_tiled")
        rb = Rulebook(nbr=nbr, K=K, cap_out=cap, geom=geom)
        rb._tile_rb = {tile_channels: t}
        rb._lean = bool(lean_table)
        rb._marked_next = bool(lean_table and mg is not None)
        return rb
    if masks and K == 27:
        rowmask = torch.empty((cap,), dtype=torch.int32, device=indices.device)
        rc = L.fnp_rulebook_subm_masked(_l.ptr(indices), _l.ptr(n_dev), cap, geom, grid.c(), _l.ptr(nbr), _l.ptr(rowmask), mg, mgeom, _l.stream())
        _l.check(rc, "fnp_rulebook_subm_masked")
        rb = Rulebook(nbr=nbr, K=K, cap_out=cap, geom=geom)
        rb._rowmask = rowmask
        rb._marked_next = mg is not None
        return rb
    rc = L.fnp_rulebook_subm(_l.ptr(indices), _l.ptr(n_dev), cap, geom, grid.c(), _l.ptr(nbr), _l.stream())
    _l.check(rc, "fnp_rulebook_subm")
    return Rulebook(nbr=nbr, K=K, cap_out=cap, geom=geom)


def tiled_by_default(channels, dtype, cap):
    """Does conv_forward take the tile-rulebook kernel by itself for a ranked `channels` -> `channels` 3x3x3 layer of this
    dtype and row capacity?  (What a caller that builds the rulebook asks, to have the tile rulebook written with it.)"""
    if TILE_MODE is not None:
        return bool(TILE_MODE) and channels in TILED_CHANNELS and dtype in (torch.bfloat16, torch.float16)
    return channels in TILED_AUTO and dtype in (torch.bfloat16, torch.float16)   # (measured ahead from 1 to 64 scenes per step)


def rulebook_strided(indices, n_dev, grid, ksize, stride, padding, cap_out, out_grid=None, want_nbr=True, premarked=False):
    """Builds out grid + out indices (rank order) + nbr.  out_n is the TRUE count (may exceed
    cap_out: the caller checks it when it synchronises).  want_nbr=False: grid and coordinates only, for a layer
    that resolves its neighbours inside the convolution (conv_forward_strided); Rulebook.nbr is then None and
    Rulebook.in_grid the input grid.  premarked: out_grid already holds the output sites (rulebook_subm(mark_next=...) of the
    input rows): the marking launch is skipped."""
    L = _l.load()
    dev = indices.device
    cap_in = max(indices.shape[0], 1)
    geom, out_shape = make_geom(ksize, stride, padding, grid.shape)
    K = geom.ksize[0] * geom.ksize[1] * geom.ksize[2]
    if out_grid is None:
        out_grid = alloc_grid(grid.batch_size, out_shape, dev)
    cap_out = max(int(cap_out), 1)
    out_idx = torch.empty((cap_out, 4), dtype=torch.int32, device=dev)
    out_n = torch.empty((1,), dtype=torch.int32, device=dev)    # always written by fnp_rulebook_strided
    nbr = torch.empty((K, cap_out), dtype=torch.int32, device=dev) if want_nbr else None
    ws = torch.empty((int(L.fnp_rankgrid_workspace_bytes(grid.batch_size, *out_shape)),), dtype=torch.uint8, device=dev)
    fn = L.fnp_rulebook_strided_premarked if premarked else L.fnp_rulebook_strided
    rc = fn(_l.ptr(indices), _l.ptr(n_dev), cap_in, geom, grid.c(), out_grid.c(with_perm=False),
            _l.ptr(out_idx), _l.ptr(out_n), cap_out, _l.ptr(nbr), _l.ptr(ws), ws.numel(), _l.stream())
    _l.check(rc, "fnp_rulebook_strided")
    return Rulebook(nbr=nbr, K=K, cap_out=cap_out, geom=geom, out_indices=out_idx, out_n=out_n, out_grid=out_grid,
                    out_shape=out_shape, in_grid=None if want_nbr else grid)


# --------------------------------------------------------------------------------- convolution
F32_MFMA_SHAPES = {(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (32, 16), (64, 32), (128, 64)}


class PermutedWeight(torch.Tensor):
    """Marker type: a packed f32 weight whose 16-channel groups are stored 4 x 4 transposed (fnp.h FNP_HINT_W_PERMUTED)."""


def pack_weight(weight, dtype, mfma_f32=False):
    """spconv 2.x layout (Cout,kD,kH,kW,Cin) -> packed (K, Cout, Cin) contiguous in `dtype`.
    (spconv 1.x (kD,kH,kW,Cin,Cout) is converted when a checkpoint is loaded: SparseConvolution._load_from_state_dict, spconv/conv.py.)
    mfma_f32: f32 weights of a shape the f32 MFMA kernel covers are stored with every group of 16 input channels
    transposed 4 x 4 (position 4q + r <- channel 4r + q): the layout that kernel reads without lane exchanges; the result
    is tagged (PermutedWeight) so that conv_forward passes FNP_HINT_W_PERMUTED."""
    Cout, Cin = weight.shape[0], weight.shape[-1]
    K = weight.shape[1] * weight.shape[2] * weight.shape[3]
    w = weight.detach().reshape(Cout, K, Cin).permute(1, 0, 2).contiguous().to(dtype)
    if mfma_f32 and dtype == torch.float32 and (Cin, Cout) in F32_MFMA_SHAPES:
        w = w.view(K, Cout, Cin // 16, 4, 4).transpose(3, 4).contiguous().view(K, Cout, Cin).as_subclass(PermutedWeight)
    return w


def pack_weight_train(weight, dtype, mirror=0):
    """pack_weight for the training path in ONE launch (fnp_pack_weight): returns (packed (K, Cout, Cin), mirror slabs or None).
    mirror 1: (K, Cout, Cin) with the offsets mirrored — what conv_dgrad of a SubM layer reads on the forward's table;
    mirror 2: mirrored and transposed (K, Cin, Cout) — the forward kernel run on the gradient of a SubM layer; mirror 3:
    transposed only — the same for a strided layer on its transposed table.  f32 parameters only (else the torch ops)."""
    Cout, Cin = weight.shape[0], weight.shape[-1]
    K = weight.shape[1] * weight.shape[2] * weight.shape[3]
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous() or not w.is_cuda:
        p = pack_weight(weight, dtype)
        m = (None if mirror == 0 else p.flip(0) if mirror == 1 else p.flip(0).transpose(1, 2).contiguous() if mirror == 2
             else p.transpose(1, 2).contiguous())
        return p, m
    L = _l.load()
    p = torch.empty((K, Cout, Cin), dtype=dtype, device=w.device)
    m = None if mirror == 0 else torch.empty((K, Cout, Cin) if mirror == 1 else (K, Cin, Cout), dtype=dtype, device=w.device)   # (2, 3: transposed)
    rc = L.fnp_pack_weight(_l.ptr(w), Cout, K, Cin, _l.dtype_code(p), _l.ptr(p), _l.ptr(m), mirror, _l.stream())
    _l.check(rc, "fnp_pack_weight")
    return p, m


def pack_weights_train(jobs, dtype):
    """pack_weight_train for several layers in ONE launch (fnp_pack_weight_multi).  jobs: list of (weight, mirror mode), f32
    contiguous CUDA parameters, at most 32; returns a list of (packed, mirror) in the same order."""
    L = _l.load()
    k = len(jobs)
    assert 0 < k <= 32
    outs, arrs = [], {n: [] for n in ("w", "p", "m", "co", "kk", "ci", "mo")}
    for weight, mirror in jobs:
        w = weight.detach()
        assert w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda
        Cout, Cin = w.shape[0], w.shape[-1]
        K = w.shape[1] * w.shape[2] * w.shape[3]
        p = torch.empty((K, Cout, Cin), dtype=dtype, device=w.device)
        m = None if mirror == 0 else torch.empty((K, Cout, Cin) if mirror == 1 else (K, Cin, Cout), dtype=dtype, device=w.device)
        outs.append((p, m))
        for n, v in (("w", _l.ptr(w)), ("p", _l.ptr(p)), ("m", _l.ptr(m)), ("co", Cout), ("kk", K), ("ci", Cin), ("mo", mirror)):
            arrs[n].append(v)
    P, I = ctypes.c_void_p * k, ctypes.c_int * k
    keep = [P(*arrs["w"]), I(*arrs["co"]), I(*arrs["kk"]), I(*arrs["ci"]), P(*arrs["p"]), P(*arrs["m"]), I(*arrs["mo"])]
    rc = L.fnp_pack_weight_multi(k, ctypes.cast(keep[0], ctypes.c_void_p), ctypes.cast(keep[1], ctypes.c_void_p), ctypes.cast(keep[2], ctypes.c_void_p),
                                 ctypes.cast(keep[3], ctypes.c_void_p), _l.dtype_code(outs[0][0]), ctypes.cast(keep[4], ctypes.c_void_p),
                                 ctypes.cast(keep[5], ctypes.c_void_p), ctypes.cast(keep[6], ctypes.c_void_p), _l.stream())
    _l.check(rc, "fnp_pack_weight_multi")
    return outs


HINT_ROWS_RANKED = 1   # fnp.h FNP_HINT_ROWS_RANKED
HINT_VALU = 2          # fnp.h FNP_HINT_VALU
HINT_W_PERMUTED = 4    # fnp.h FNP_HINT_W_PERMUTED
# development / tests: force (True) or forbid (False) the tile-rulebook kernel wherever a caller leaves `tile` unset (FNP_TILE=1 / 0)
TILE_MODE = {"0": False, "1": True}.get(os.environ.get("FNP_TILE", ""))


def tile_rulebook(rb, n_out_dev, channels):
    """The tile rulebook of a 3x3x3 rulebook for `channels`-channel layers (fnp_tile_rulebook_build),
    built on first use and kept with it: valid as long as rb.nbr and the row count are (a Rulebook object is never rewritten in place)."""
    cache = rb.__dict__.setdefault("_tile_rb", {})
    key = channels
    t = cache.get(key)
    if t is None:
        L = _l.load()
        if getattr(rb, "_lean", False):
            raise _l.FnpError("this rulebook's int32 table holds the rows of escape tiles only: no other tile rulebook can be made from it")
        t = torch.empty((L.fnp_tile_rulebook_bytes(rb.cap_out, channels),), dtype=torch.uint8, device=rb.nbr.device)
        rc = L.fnp_tile_rulebook_build(_l.ptr(rb.nbr), rb.nbr.shape[1], rb.K, _l.ptr(n_out_dev), rb.cap_out, channels, _l.ptr(t), _l.stream())
        _l.check(rc, "fnp_tile_rulebook_build")
        cache[key] = t
    return t


TILED_CHANNELS = (32, 64)   # channel counts fnp_spconv_forward_tiled covers


def tiled_fits(n_in_rows, channels, nbr_stride, cap_out):
    """fnp_spconv_forward_tiled's documented size limit (fnp.h): the feature tensor, the int32 table and the tile rulebook are
    addressed with 32-bit byte offsets."""
    rb_bytes = int(_l.load().fnp_tile_rulebook_bytes(cap_out, channels))
    return n_in_rows * channels * 2 < 0x7fffffff and 27 * nbr_stride * 4 < 0x7fffffff and rb_bytes < 0x7fffffff

# ... and the ones that take it by themselves (measured at 64 scenes, per layer: 32 channels 0.305 -> 0.18 ms; 64 channels
# 0.43 -> 0.385 ms, +1 % end to end)
TILED_AUTO = tuple(int(v) for v in os.environ.get("FNP_TILED_AUTO", "32,64").split(",") if v)   # (development A/B: e.g. FNP_TILED_AUTO=32)


# --------------------------------------------------------------------------------- compact rulebook (sparse-neighbourhood layers)
ELL_SHAPES = {(4, 16), (5, 16), (16, 16), (16, 32)}   # (Cin, Cout) fnp_spconv_forward_ell covers (the first: f32 point features in)
ELL_MODE = {"0": False, "1": True}.get(os.environ.get("FNP_ELL", ""))   # (development / tests: forbid or force)


def ell_rulebook(coords, n_dev, cap, geom, in_grid, pool_records, nbr=None, used=None):
    """Compact rulebook (fnp_rulebook_ell) of the rows whose cells are `coords`: returns (records, pool_records, pool_used).
    nbr: optional (27, cap) int32 tensor that receives the table of the same rows in the same pass.
    used: a (1,) int32 counter the caller keeps, holding ZERO (reset when it is read, fnp_gather_counts): no launch to clear one.
    No host sync: the caller reads pool_used with its other counts and discards the result when it exceeds the pool."""
    L = _l.load()
    dev = coords.device
    pool_records = max(int(pool_records), 0)
    rec = torch.empty((int(L.fnp_ell_bytes(cap, pool_records)) // 4,), dtype=torch.int32, device=dev)
    kept = used is not None
    if not kept:
        used = torch.empty((1,), dtype=torch.int32, device=dev)
    rc = L.fnp_rulebook_ell(_l.ptr(coords), _l.ptr(n_dev), cap, geom, in_grid.c(), _l.ptr(rec), pool_records, _l.ptr(used), int(kept), _l.ptr(nbr),
                            _l.stream())
    _l.check(rc, "fnp_rulebook_ell")
    return rec, pool_records, used


def rulebook_subm_ell(indices, n_dev, grid, pool_records, with_table=False, used=None):
    """SubM 3x3x3 rulebook in the compact form (`_ell` set); with_table: the (27, cap) table too, from the same pass (else
    Rulebook.nbr is None)."""
    cap = max(indices.shape[0], 1)
    geom, _ = make_geom(3, 1, 1, grid.shape, grid.shape)
    nbr = torch.empty((27, cap), dtype=torch.int32, device=indices.device) if with_table else None
    rb = Rulebook(nbr=nbr, K=27, cap_out=cap, geom=geom)
    rb._ell = ell_rulebook(indices, n_dev, cap, geom, grid, pool_records, nbr=nbr, used=used)
    return rb


def ell_for_strided(rb, pool_records, used=None):
    """Compact rulebook of a strided 3x3x3 layer built with want_nbr=False (its output coordinates + the input grid)."""
    assert rb.nbr is None and rb.in_grid is not None and rb.K == 27
    rb._ell = ell_rulebook(rb.out_indices, rb.out_n, rb.cap_out, rb.geom, rb.in_grid, pool_records, used=used)
    return rb


ELL_MFMA = os.environ.get("FNP_ELL_MFMA", "1") == "1"   # 16-channel rows on records: the matrix kernel (else the VALU kernel)


def conv_forward_ell(feat_in, w_packed, rb, n_out_dev, out_dtype=None, scale=None, shift=None, residual=None, relu=False, mfma=None):
    """conv_forward on the compact rulebook (`rb._ell`): the sparse-neighbourhood layers.  No host sync.
    mfma: 16-channel rows only — True: the MFMA kernel with the records expanded per tile (fnp_spconv_forward_ell_mfma: the
    table kernel's values bit for bit), False: the VALU kernel (another summation order); None: ELL_MFMA."""
    L = _l.load()
    rec, pool, _ = rb._ell
    K, Cout, Cin = w_packed.shape
    assert K == 27 and (Cin, Cout) in ELL_SHAPES and feat_in.shape[1] == Cin and feat_in.dtype == w_packed.dtype
    assert feat_in.is_contiguous() and w_packed.is_contiguous() and not isinstance(w_packed, PermutedWeight)
    out_dtype = out_dtype or feat_in.dtype
    out = torch.empty((rb.cap_out, Cout), dtype=out_dtype, device=feat_in.device)
    if residual is not None:
        assert residual.dtype == out.dtype and residual.shape[1] == Cout and residual.is_contiguous()
    use_mfma = (ELL_MFMA if mfma is None else mfma) and Cin == 16 and feat_in.dtype in (torch.bfloat16, torch.float16) and out.dtype == feat_in.dtype
    fn = L.fnp_spconv_forward_ell_mfma if use_mfma else L.fnp_spconv_forward_ell
    rc = fn(_l.ptr(feat_in), _l.dtype_code(feat_in), feat_in.shape[0], _l.ptr(w_packed), _l.ptr(rec), rb.cap_out, pool,
            _l.ptr(n_out_dev), _l.ptr(out), _l.dtype_code(out), _l.ptr(scale), _l.ptr(shift), _l.ptr(residual),
            int(bool(relu)), Cin, Cout, _l.stream())
    _l.check(rc, "fnp_spconv_forward_ell_mfma" if use_mfma else "fnp_spconv_forward_ell")
    return out


# the next stage's output sites marked by this stage's rulebook kernel (rulebook_subm(mark_next=...)): built and tested, OFF by
# default — measured -1.1 % end to end at 64 scenes (the atomics' dependent chain lengthens every pass of the rulebook kernels,
# which the stand-alone marking launch hides behind a grid of its own with the next row's coordinates prefetched)
MARK_FUSED = os.environ.get("FNP_MARK_FUSED", "0") == "1"
SORTED_SHAPES = {(128, 128)}   # (Cin, Cout) fnp_spconv_forward_sorted covers
# the class-sorted sweep pays from a few scenes on (one more small kernel per forward against ~20 % of four sweeps); 0 / 1 force it
SORT_MODE = {"0": False, "1": True}.get(os.environ.get("FNP_SORT", ""))
SORT_MIN_ROWS = 131072   # (row CAPACITY of the stage: a one-scene hipGraph has 65,536; from ~4 scenes on the sort pays)


def sorted_by_default(cin, cout, dtype, cap):
    if (cin, cout) not in SORTED_SHAPES or dtype not in (torch.bfloat16, torch.float16):
        return False
    return bool(SORT_MODE) if SORT_MODE is not None else cap >= SORT_MIN_ROWS


def classsort(rb, n_out_dev, channels=128):
    """Processing order of a 3x3x3 SubM rulebook for the class-sorted 128-channel sweep (fnp_rulebook_classsort): kept with
    the rulebook (`rb._sorted` = (perm, blockmask)), valid as long as rb.nbr and the row count are.  Starts from the row masks
    the rulebook kernel wrote (rulebook_subm(masks=True)) or derives them from the table.  No host sync."""
    L = _l.load()
    assert rb.K == 27
    dev = rb.nbr.device
    perm = torch.empty((rb.cap_out,), dtype=torch.int32, device=dev)
    blockmask = torch.empty((rb.cap_out // 16 + 1,), dtype=torch.int32, device=dev)
    rowmask = getattr(rb, "_rowmask", None)
    ws = None
    if rowmask is None:
        ws = torch.empty((int(L.fnp_classsort_workspace_bytes(rb.cap_out)),), dtype=torch.uint8, device=dev)
    rc = L.fnp_rulebook_classsort(_l.ptr(rb.nbr), rb.nbr.shape[1], rb.K, _l.ptr(rowmask), _l.ptr(n_out_dev), rb.cap_out, channels, channels,
                                  _l.ptr(perm), _l.ptr(blockmask), _l.ptr(ws), 0 if ws is None else ws.numel(), _l.stream())
    _l.check(rc, "fnp_rulebook_classsort")
    rb._sorted = (perm, blockmask)
    if rowmask is None:
        rb._rowmask = ws.view(torch.int32)
    return rb


# f32 engine: the 3x3x3 SubM layers sweep every workgroup range class by class (the f32 kernel is bound by the matrix pipe and
# skips the MFMAs of a 16-row block at offsets none of its rows has a neighbour at).  FNP_F32_SORT=0: row order (development A/B)
F32_SORT = os.environ.get("FNP_F32_SORT", "1") != "0"
F32_SORT_CHANNELS = {16, 32, 64, 128}
F32_SORT_MIN_ROWS = int(os.environ.get("FNP_F32_SORT_MIN_ROWS", "32768"))


def f32_sorted_by_default(channels, dtype, cap):
    return F32_SORT and dtype == torch.float32 and channels in F32_SORT_CHANNELS and cap >= F32_SORT_MIN_ROWS


def classsort_f32(rb, n_out_dev, channels):
    """Processing order of a 3x3x3 SubM rulebook for the f32 `channels` -> `channels` layers (fnp_rulebook_classsort_f32), kept
    with the rulebook (`rb._perm_f32[channels]`); needs the row masks of rulebook_subm(masks=True).  No host sync."""
    L = _l.load()
    assert rb.K == 27 and getattr(rb, "_rowmask", None) is not None
    perm = torch.empty((rb.cap_out,), dtype=torch.int32, device=rb.nbr.device)
    rc = L.fnp_rulebook_classsort_f32(_l.ptr(rb._rowmask), _l.ptr(n_out_dev), rb.cap_out, channels, channels, _l.ptr(perm), _l.stream())
    _l.check(rc, "fnp_rulebook_classsort_f32")
    if getattr(rb, "_perm_f32", None) is None:
        rb._perm_f32 = {}
    rb._perm_f32[channels] = perm
    return rb


def conv_forward(feat_in, w_packed, rb, n_out_dev, out_dtype=None, scale=None, shift=None, residual=None, relu=False,
                 out=None, ranked=False, valu=False, tile=None):
    """feat_out (cap_out, Cout) = act(conv * scale + shift + residual).  No host sync.
    ranked: input and output rows are both in rank-grid order (performance hint only).
    valu: f32 only — the thread-per-element chain instead of the f32 MFMA kernel (same bits).
    tile: 16-bit 32 -> 32 and 64 -> 64 layers of 3x3x3 kernels — True / False forces / forbids the tile-rulebook kernel
    (None: ranked tensors take it; same bits either way)."""
    L = _l.load()
    _l.require_device(feat_in, w_packed, rb.nbr, n_out_dev)
    if tile is None:
        tile = TILE_MODE
    if tile is None and getattr(rb, "_no_tile", False):   # (the fused engine's tile gate: this rulebook's stage runs on the gather kernels)
        tile = False
    K, Cout, Cin = w_packed.shape
    assert K == rb.K and feat_in.shape[1] == Cin and feat_in.dtype == w_packed.dtype
    assert feat_in.is_contiguous() and w_packed.is_contiguous()
    out_dtype = out_dtype or feat_in.dtype
    cap_out = rb.cap_out
    if out is None:
        out = torch.empty((cap_out, Cout), dtype=out_dtype, device=feat_in.device)
    if residual is not None:
        assert residual.dtype == out.dtype and residual.shape[1] == Cout and residual.is_contiguous()
    if scale is not None:
        assert scale.dtype == torch.float32 and shift.dtype == torch.float32
    if (K == 27 and Cin == Cout and Cin in TILED_CHANNELS and feat_in.dtype in (torch.bfloat16, torch.float16) and out.dtype == feat_in.dtype
            and (tile or (tile is None and ranked and tiled_by_default(Cin, feat_in.dtype, cap_out)))
            and tiled_fits(feat_in.shape[0], Cin, rb.nbr.shape[1], cap_out)):
        # (a tensor beyond the tiled kernels' 32-bit offsets takes the gather kernel below: decided HERE from the documented
        #  condition, so that any error code of the call is an error and not a silent change of kernel)
        rc = L.fnp_spconv_forward_tiled(_l.ptr(feat_in), _l.dtype_code(feat_in), feat_in.shape[0], _l.ptr(w_packed),
                                        _l.ptr(tile_rulebook(rb, n_out_dev, Cin)), _l.ptr(rb.nbr), rb.nbr.shape[1], _l.ptr(n_out_dev), cap_out,
                                        _l.ptr(out), _l.ptr(scale), _l.ptr(shift), _l.ptr(residual), int(bool(relu)), Cin, Cout, _l.stream())
        _l.check(rc, "fnp_spconv_forward_tiled")
        return out
    if getattr(rb, "_lean", False):
        raise _l.FnpError("this rulebook's int32 table holds the rows of escape tiles only (rulebook_subm(lean_table=True)): "
                          "only the tiled convolution of its channel count may run on it")
    srt = getattr(rb, "_sorted", None)
    if (srt is not None and K == 27 and (Cin, Cout) in SORTED_SHAPES and feat_in.dtype in (torch.bfloat16, torch.float16)
            and out.dtype == feat_in.dtype and feat_in.shape[0] * Cin * 2 < 0x7fffffff):
        rc = L.fnp_spconv_forward_sorted(_l.ptr(feat_in), _l.dtype_code(feat_in), feat_in.shape[0], _l.ptr(w_packed), _l.ptr(rb.nbr),
                                         rb.nbr.shape[1], _l.ptr(srt[0]), _l.ptr(srt[1]), _l.ptr(n_out_dev), cap_out, _l.ptr(out),
                                         _l.ptr(scale), _l.ptr(shift), _l.ptr(residual), int(bool(relu)), Cin, Cout, _l.stream())
        _l.check(rc, "fnp_spconv_forward_sorted")
        return out
    pf = (getattr(rb, "_perm_f32", None) or {}).get(Cin)
    if (pf is not None and K == 27 and Cin == Cout and feat_in.dtype == torch.float32 and out.dtype == torch.float32 and not valu
            and feat_in.shape[0] * Cin * 4 < 0x7fffffff):
        rc = L.fnp_spconv_forward_f32_sorted(_l.ptr(feat_in), feat_in.shape[0], _l.ptr(w_packed), _l.ptr(rb.nbr), rb.nbr.shape[1], _l.ptr(pf),
                                             _l.ptr(n_out_dev), cap_out, _l.ptr(out), _l.ptr(scale), _l.ptr(shift), _l.ptr(residual),
                                             int(bool(relu)), int(isinstance(w_packed, PermutedWeight)), Cin, Cout, _l.stream())
        _l.check(rc, "fnp_spconv_forward_f32_sorted")
        return out
    if isinstance(w_packed, PermutedWeight) and feat_in.shape[0] * Cin * 4 >= 0x7fffffff:
        # the f32 MFMA kernel addresses the features with 32-bit offsets; beyond them the thread-per-element chain runs,
        # and it reads the plain (K, Cout, Cin) layout: undo the 4 x 4 transposition of the 16-channel groups
        w_packed = w_packed.as_subclass(torch.Tensor).view(K, Cout, Cin // 16, 4, 4).transpose(3, 4).contiguous().view(K, Cout, Cin)
    rc = L.fnp_spconv_forward(_l.ptr(feat_in), _l.dtype_code(feat_in), feat_in.shape[0], _l.ptr(w_packed),
                              _l.ptr(rb.nbr), rb.nbr.shape[1], K, _l.ptr(n_out_dev), cap_out,
                              _l.ptr(out), _l.dtype_code(out), _l.ptr(scale), _l.ptr(shift), _l.ptr(residual),
                              int(bool(relu)), (HINT_ROWS_RANKED if ranked else 0) | (HINT_VALU if valu else 0) |
                              (HINT_W_PERMUTED if isinstance(w_packed, PermutedWeight) else 0), Cin, Cout, _l.stream())
    _l.check(rc, "fnp_spconv_forward")
    return out


# (Cin, Cout) the matrix kernel behind fnp_spconv_forward_split covers
SPLIT_SHAPES = {(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (32, 16), (64, 32), (128, 64)}


def conv_forward_split(feat_in, w_packed, rb, n_out_dev, scale=None, shift=None, residual=None, addend=None, relu=True, ranked=False, tile=None,
                       want_f32=True):
    """The bf16x3 engine's main product (fnp_spconv_forward_split / _tiled_split): 16-bit features and weights,
    y = act(conv * scale + shift + residual (f32) + float(addend) (16-bit)), returned as (y f32, hi, lo) with hi = 16bit(y),
    lo = 16bit(y - hi) — conv_forward(out_dtype=float32, relu=False) followed by split_bf16_add, bit for bit, in one launch.
    ranked / tile: as conv_forward (the 32 -> 32 and 64 -> 64 layers of ranked tensors run on the tile rulebook).
    want_f32=False: y is not written (returned as None) — a layer of which only the split is read.  No host sync."""
    L = _l.load()
    _l.require_device(feat_in, w_packed, rb.nbr, n_out_dev)
    K, Cout, Cin = w_packed.shape
    assert K == rb.K and feat_in.shape[1] == Cin and feat_in.dtype == w_packed.dtype and feat_in.dtype in (torch.bfloat16, torch.float16)
    assert feat_in.is_contiguous() and w_packed.is_contiguous() and not isinstance(w_packed, PermutedWeight)
    cap_out, dev = rb.cap_out, feat_in.device
    y = torch.empty((cap_out, Cout), dtype=torch.float32, device=dev) if want_f32 else None
    hi = torch.empty((cap_out, Cout), dtype=feat_in.dtype, device=dev)
    lo = torch.empty((cap_out, Cout), dtype=feat_in.dtype, device=dev)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.shape == hi.shape and residual.is_contiguous()
    if addend is not None:
        assert addend.dtype == feat_in.dtype and addend.shape == hi.shape and addend.is_contiguous()
    if scale is not None:
        assert scale.dtype == torch.float32 and shift.dtype == torch.float32
    if tile is None:
        tile = TILE_MODE
    if tile is None and getattr(rb, "_no_tile", False):
        tile = False
    if (K == 27 and Cin == Cout and Cin in TILED_CHANNELS and (tile or (tile is None and ranked and tiled_by_default(Cin, feat_in.dtype, cap_out)))
            and tiled_fits(feat_in.shape[0], Cin, rb.nbr.shape[1], cap_out)):
        rc = L.fnp_spconv_forward_tiled_split(_l.ptr(feat_in), _l.dtype_code(feat_in), feat_in.shape[0], _l.ptr(w_packed),
                                              _l.ptr(tile_rulebook(rb, n_out_dev, Cin)), _l.ptr(rb.nbr), rb.nbr.shape[1], _l.ptr(n_out_dev), cap_out,
                                              _l.ptr(y), _l.ptr(scale), _l.ptr(shift), _l.ptr(residual), _l.ptr(addend), int(bool(relu)), Cin, Cout,
                                              _l.ptr(hi), _l.ptr(lo), _l.stream())
        _l.check(rc, "fnp_spconv_forward_tiled_split")
        return y, hi, lo
    if getattr(rb, "_lean", False):
        raise _l.FnpError("this rulebook's int32 table holds the rows of escape tiles only (rulebook_subm(lean_table=True)): "
                          "only the tiled convolution of its channel count may run on it")
    srt = getattr(rb, "_sorted", None)
    if srt is not None and ranked and K == 27 and (Cin, Cout) in SORTED_SHAPES and feat_in.shape[0] * Cin * 2 < 0x7fffffff:
        rc = L.fnp_spconv_forward_sorted_split(_l.ptr(feat_in), _l.dtype_code(feat_in), feat_in.shape[0], _l.ptr(w_packed), _l.ptr(rb.nbr), rb.nbr.shape[1],
                                               _l.ptr(srt[0]), _l.ptr(srt[1]), _l.ptr(n_out_dev), cap_out, _l.ptr(y), _l.ptr(scale), _l.ptr(shift),
                                               _l.ptr(residual), _l.ptr(addend), int(bool(relu)), Cin, Cout, _l.ptr(hi), _l.ptr(lo), _l.stream())
        _l.check(rc, "fnp_spconv_forward_sorted_split")
        return y, hi, lo
    rc = L.fnp_spconv_forward_split(_l.ptr(feat_in), _l.dtype_code(feat_in), feat_in.shape[0], _l.ptr(w_packed), _l.ptr(rb.nbr), rb.nbr.shape[1], K,
                                    _l.ptr(n_out_dev), cap_out, _l.ptr(y), _l.ptr(scale), _l.ptr(shift), _l.ptr(residual), _l.ptr(addend),
                                    int(bool(relu)), HINT_ROWS_RANKED if ranked else 0, Cin, Cout, _l.ptr(hi), _l.ptr(lo), _l.stream())
    _l.check(rc, "fnp_spconv_forward_split")
    return y, hi, lo


def split_bf16(x, n_dev):
    """f32 rows (cap, C) -> (hi, lo) bf16 with hi + lo == x to 2^-17 of |x| (fnp_split_bf16); rows >= n are left undefined."""
    L = _l.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _l.check(L.fnp_split_bf16(_l.ptr(x), _l.ptr(n_dev), x.shape[0], x.shape[1], _l.ptr(hi), _l.ptr(lo), _l.stream()), "fnp_split_bf16")
    return hi, lo


def split_bf16_add(x, t, n_dev, relu=True):
    """y = relu(x + float(t)) written over x (f32), and its (hi, lo) bf16 split (fnp_split_bf16_add)."""
    L = _l.load()
    assert x.dtype == torch.float32 and t.dtype == torch.bfloat16 and x.shape == t.shape and x.is_contiguous() and t.is_contiguous()
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _l.check(L.fnp_split_bf16_add(_l.ptr(x), _l.ptr(t), int(bool(relu)), _l.ptr(n_dev), x.shape[0], x.shape[1], _l.ptr(x), _l.ptr(hi), _l.ptr(lo),
                                  _l.stream()), "fnp_split_bf16_add")
    return hi, lo


def conv_forward_strided(feat_in, w_packed, rb, scale=None, shift=None, relu=False, out=None):
    """A strided 3x3x3 convolution whose rulebook rows are computed inside the kernel (rb from
    rulebook_strided(..., want_nbr=False)).  bf16 or fp16, (Cin, Cout) in {(16,32), (32,64), (64,128)}.  Same result as
    conv_forward on the table.  No host sync."""
    L = _l.load()
    _l.require_device(feat_in, w_packed, rb.out_indices, rb.out_n)
    K, Cout, Cin = w_packed.shape
    assert rb.nbr is None and rb.in_grid is not None and K == 27 and feat_in.shape[1] == Cin
    assert feat_in.dtype in (torch.bfloat16, torch.float16) and w_packed.dtype == feat_in.dtype and feat_in.is_contiguous()
    if out is None:
        out = torch.empty((rb.cap_out, Cout), dtype=feat_in.dtype, device=feat_in.device)
    rc = L.fnp_spconv_forward_strided(_l.ptr(feat_in), _l.dtype_code(feat_in), feat_in.shape[0], _l.ptr(w_packed),
                                      rb.in_grid.c(), rb.geom, _l.ptr(rb.out_indices), _l.ptr(rb.out_n), rb.cap_out,
                                      _l.ptr(out), _l.dtype_code(out), _l.ptr(scale), _l.ptr(shift), int(bool(relu)),
                                      Cin, Cout, _l.stream())
    _l.check(rc, "fnp_spconv_forward_strided")
    return out


# layers the fused backbone runs this way (16 -> 32 is built and tested too, but measured 2.5 % slower end to end than its table
# path: its input grid carries the voxeliser's permutation and its LDS strip costs a resident workgroup)
FUSED_STRIDED_SHAPES = {(32, 64), (64, 128)} | ({(16, 32)} if os.environ.get("FNP_FUSED1632") == "1" else set())   # (development switch)


# --------------------------------------------------------------------------------- backward
def rulebook_transpose(rb, n_out_dev, cap_in):
    """nbr (K, cap_out) over output rows -> (K, cap_in) over input rows (for the data gradient)."""
    L = _l.load()
    nbr_t = torch.empty((rb.K, cap_in), dtype=torch.int32, device=rb.nbr.device)
    rc = L.fnp_rulebook_transpose(_l.ptr(rb.nbr), rb.nbr.shape[1], rb.K, _l.ptr(n_out_dev), rb.cap_out, _l.ptr(nbr_t), cap_in,
                                  _l.stream())
    _l.check(rc, "fnp_rulebook_transpose")
    return nbr_t


def conv_dgrad(grad_out, w_packed, nbr_t, n_in_dev, cap_in, pretransposed=False):
    """dx (cap_in, Cin) = sum_k W_k dy[nbr_t[k]]: the forward kernel on the transposed rulebook and slabs.
    w_packed (K, Cout, Cin) in grad_out's dtype; pretransposed: it is (K, Cin, Cout) already (pack_weight_train mirror 2 / 3)."""
    L = _l.load()
    if pretransposed:
        K, Cin, Cout = w_packed.shape
        w_t = w_packed
        assert w_t.is_contiguous()
    else:
        K, Cout, Cin = w_packed.shape
        w_t = w_packed.transpose(1, 2).contiguous()            # (K, Cin, Cout): "Cout" = Cin, "Cin" = Cout
    assert grad_out.is_contiguous() and grad_out.dtype == w_t.dtype and grad_out.shape[1] == Cout
    dx = torch.empty((cap_in, Cin), dtype=grad_out.dtype, device=grad_out.device)
    rc = L.fnp_spconv_forward(_l.ptr(grad_out), _l.dtype_code(grad_out), grad_out.shape[0], _l.ptr(w_t),
                              _l.ptr(nbr_t), nbr_t.shape[1], K, _l.ptr(n_in_dev), cap_in,
                              _l.ptr(dx), _l.dtype_code(dx), None, None, None, 0, 0, Cout, Cin, _l.stream())
    _l.check(rc, "fnp_spconv_forward (dgrad)")
    return dx


WGRAD_PAIR_SHAPES = {(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)}   # (Cin, Cout) of the MFMA weight gradient
WGRAD_PAIRS = os.environ.get("FNP_WGRAD_PAIRS", "1") == "1"


def mfma_pad_channels(cin, cout):
    """Input channels a 16-bit training layer of `cin` < 16 channels is zero-padded to so that its forward AND its weight gradient
    run on the matrix kernels (SPLIT_SHAPES = the shapes the 16-bit matrix kernel covers, WGRAD_PAIR_SHAPES those of the MFMA
    weight gradient), or 0 when no such shape exists / none is needed: conv_input (4 / 5 -> 16, and 4 / 5 -> 32 of a wider first
    layer) -> 16."""
    if (cin, cout) in SPLIT_SHAPES and (cin, cout) in WGRAD_PAIR_SHAPES:
        return 0
    fits = sorted(ci for (ci, co) in SPLIT_SHAPES & WGRAD_PAIR_SHAPES if co == cout and ci >= cin)
    return fits[0] if fits else 0


def rulebook_pairs(rb, n_out_dev, rows=None):
    """Pair lists of a rulebook (fnp_rulebook_pairs), kept with it (`rb._pairs` = (pair_o, pair_i, pair_count)): per kernel
    offset the output rows that have a neighbour there, ascending, and those neighbours.  Built once per rulebook and
    backward (the two to four convolutions of a stage share it).  No host sync.  rows: an upper bound of the row count the
    caller knows on the host (a strided layer's table is sized for the worst case: 27 outputs per input)."""
    L = _l.load()
    K, stride = rb.K, rb.nbr.shape[1]
    cap = min(rb.cap_out, int(rows)) if rows else rb.cap_out
    dev = rb.nbr.device
    po = torch.empty((K, cap), dtype=torch.int32, device=dev)     # (cap, not the table's stride: see `rows`)
    pi = torch.empty((K, cap), dtype=torch.int32, device=dev)
    cnt = torch.empty((K,), dtype=torch.int32, device=dev)
    ws = torch.empty((int(L.fnp_rulebook_pairs_workspace_bytes(K, cap)),), dtype=torch.uint8, device=dev)
    rc = L.fnp_rulebook_pairs(_l.ptr(rb.nbr), stride, K, _l.ptr(n_out_dev), cap, _l.ptr(po), _l.ptr(pi), cap, _l.ptr(cnt), _l.ptr(ws), ws.numel(),
                              _l.stream())
    _l.check(rc, "fnp_rulebook_pairs")
    rb._pairs = (po, pi, cnt)
    return rb._pairs


def conv_wgrad(feat_in, grad_out, rb, n_out_dev, Cin, Cout, pairs=None, module_shape=None):
    """dW (K, Cout, Cin) f32 = sum_o dy[o] (x) x[nbr[k][o]] (deterministic two-stage reduction).
    pairs: None = on the rulebook's pair lists where they apply (16-bit tensors, MFMA channel pairs; built on first use and
    kept with the rulebook), False = the sweep over the table.
    module_shape: (Cout, kD, kH, kW, Cin) — the result in the module parameter's layout instead (same values; the final
    reduction writes it: no permute-copy)."""
    L = _l.load()
    assert feat_in.is_contiguous() and grad_out.is_contiguous()
    assert feat_in.shape[1] == Cin and grad_out.shape[1] == Cout
    layout = 0 if module_shape is None else 1
    dw = torch.empty((rb.K, Cout, Cin) if module_shape is None else tuple(module_shape), dtype=torch.float32, device=feat_in.device)
    ws = torch.empty((int(L.fnp_spconv_wgrad_workspace_bytes(rb.K, Cin, Cout)),), dtype=torch.uint8, device=feat_in.device)
    cap = min(rb.cap_out, grad_out.shape[0])
    use_pairs = WGRAD_PAIRS if pairs is None else pairs
    mfma_pairs = (Cin, Cout) in WGRAD_PAIR_SHAPES and feat_in.dtype in (torch.bfloat16, torch.float16) and grad_out.dtype == feat_in.dtype
    small_pairs = Cin * Cout <= 128 and feat_in.dtype == torch.float32 and grad_out.dtype == torch.float32   # (conv_input: 3.6 of 27 neighbours per row)
    if use_pairs and (mfma_pairs or small_pairs) and rb.nbr is not None and not getattr(rb, "_lean", False):
        pr = getattr(rb, "_pairs", None) or rulebook_pairs(rb, n_out_dev, rows=cap)
        rc = L.fnp_spconv_wgrad_pairs(_l.ptr(feat_in), _l.dtype_code(feat_in), _l.ptr(grad_out), _l.dtype_code(grad_out), _l.ptr(pr[0]), _l.ptr(pr[1]),
                                      _l.ptr(pr[2]), pr[0].shape[1], rb.K, _l.ptr(n_out_dev), min(cap, pr[0].shape[1]), _l.ptr(dw), layout, Cin, Cout, _l.ptr(ws), ws.numel(),
                                      _l.stream())
        _l.check(rc, "fnp_spconv_wgrad_pairs")
        return dw
    rc = L.fnp_spconv_wgrad(_l.ptr(feat_in), _l.dtype_code(feat_in), _l.ptr(grad_out), _l.dtype_code(grad_out),
                            _l.ptr(rb.nbr), rb.nbr.shape[1], rb.K, _l.ptr(n_out_dev), cap, _l.ptr(dw), layout, Cin, Cout,
                            _l.ptr(ws), ws.numel(), _l.stream())
    _l.check(rc, "fnp_spconv_wgrad")
    return dw


def to_dense(features, indices, n_dev, batch_size, shape, workspace=None, out=None, fill=None):
    """SparseConvTensor.dense(): (B, C, D, H, W), written once (zeros included) through a cell -> row map.
    workspace / out: optional persistent buffers of a caller that densifies every step.
    fill: optional (C,) f32 device tensor — the value of cells without a row, per channel, instead of 0."""
    L = _l.load()
    C = features.shape[1]
    need = int(L.fnp_sparse_to_dense_workspace_bytes(batch_size, *shape))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=features.device)
    if out is None:
        out = torch.empty((batch_size, C, *shape), dtype=features.dtype, device=features.device)
    assert out.is_contiguous() and out.dtype == features.dtype and tuple(out.shape) == (batch_size, C, *shape)
    if fill is not None:
        assert fill.dtype == torch.float32 and fill.numel() == C and fill.is_contiguous()
        rc = L.fnp_sparse_to_dense_fill(_l.ptr(features), _l.dtype_code(features), _l.ptr(indices), _l.ptr(n_dev),
                                        max(indices.shape[0], 1), C, batch_size, *shape, _l.ptr(out), _l.ptr(fill), _l.ptr(workspace),
                                        workspace.numel(), _l.stream())
        _l.check(rc, "fnp_sparse_to_dense_fill")
        return out
    rc = L.fnp_sparse_to_dense(_l.ptr(features), _l.dtype_code(features), _l.ptr(indices), _l.ptr(n_dev),
                               max(indices.shape[0], 1), C, batch_size, *shape, _l.ptr(out), _l.ptr(workspace),
                               workspace.numel(), _l.stream())
    _l.check(rc, "fnp_sparse_to_dense")
    return out
