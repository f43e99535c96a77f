"""SubMConv3d / SparseConv3d modules (spconv_backbone.py:12-17,39-46,193-234).

Parameters keep spconv 2.x names and layout so the released checkpoint loads through the
reference's own loader (detector3d_template.py:401-433): `weight` (Cout,kD,kH,kW,Cin), `bias`.
With grad enabled the convolution runs through `SparseConvFunction`: data gradient
= the forward kernel on the transposed rulebook, weight gradient = deterministic two-stage reduction
(`csrc/spconv_bwd.hip`); BatchNorm / ReLU / residual adds stay ordinary torch ops on `.features`.
"""
import math

import threading

import torch
import torch.nn as nn

from .. import sparse as S
from .core import SparseConvTensor
from .modules import SparseModule


def kvol_is_27(ks):
    return tuple(int(v) for v in ks) == (3, 3, 3)


def _triple(v):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


class SparseConvFunction(torch.autograd.Function):
    """out (cap_out, Cout) = sum_k W_k^T x[nbr[k]]  with autograd (features and weight).

    weight is the module parameter (Cout, kD, kH, kW, Cin); computation dtype = features dtype (f32, bf16 or
    fp16), weight gradient accumulated in f32 and returned in the parameter's dtype."""

    @staticmethod
    def forward(ctx, feats, weight, rb, n_out_dev, n_in_dev, ranked=False, rows=None, prepacked=None):
        # rows: the output row count when the caller knows it on the host — the output then has exactly that many rows and
        # nobody slices it afterwards (the backward of a slice zero-fills a tensor of the full capacity: 4 x 0.1 ms per
        # training step at 16 scenes)
        # the slabs the backward's data gradient reads are written by the same launch as the forward's (one kernel instead of a
        # permute-copy and a cast here, a flip and a transpose-copy there: the training step is bound by the host's launches)
        Cout, Cin = weight.shape[0], weight.shape[-1]
        odd = all(int(v) & 1 for v in rb.geom.ksize)
        subm_self = rb.out_indices is None and rb.cap_out == feats.shape[0] and odd
        # (2: mirrored + transposed, a SubM layer's data gradient on the forward's own table; 3: transposed, a strided layer's on
        #  its transposed table — either way the slabs the forward KERNEL reads when it runs on the gradient)
        mirror = 0
        if ctx.needs_input_grad[0]:
            mirror = 2 if subm_self else 3
        # prepacked: (dtype, mirror mode, packed, mirror slabs) from the backbone's one launch for all its layers
        if prepacked is not None and prepacked[0] == feats.dtype and (mirror == 0 or prepacked[1] == mirror):
            w, wm = prepacked[2], prepacked[3]
        else:
            w, wm = S.pack_weight_train(weight, feats.dtype, mirror)
        out = S.conv_forward(feats, w, rb, n_out_dev, ranked=ranked,
                             out=None if rows is None else torch.empty((rows, weight.shape[0]), dtype=feats.dtype, device=feats.device))
        ctx.save_for_backward(feats, weight)
        ctx.rb, ctx.n_out_dev, ctx.n_in_dev, ctx.ranked = rb, n_out_dev, n_in_dev, ranked
        ctx.w_packed, ctx.w_mirror, ctx.mirror = w, wm, mirror
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight = ctx.saved_tensors
        rb, n_out_dev, n_in_dev = ctx.rb, ctx.n_out_dev, ctx.n_in_dev
        grad_out = grad_out.contiguous().to(feats.dtype)
        K, Cout, Cin = rb.K, weight.shape[0], weight.shape[-1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wp = ctx.w_packed
            odd = all(int(v) & 1 for v in rb.geom.ksize)
            if rb.out_indices is None and rb.cap_out == feats.shape[0] and odd:
                # SubM: the rulebook is its own transpose up to the mirror of the offsets (input i feeds output o through
                # offset k  <=>  o is the neighbour of i at offset K-1-k): same table, weight slabs in mirrored order.
                # That identity needs a centred kernel (every size odd); an even size takes the transposed table below.
                wt = ctx.w_mirror if ctx.mirror == 2 else wp.flip(0).transpose(1, 2).contiguous()
                if ctx.ranked and Cin == Cout and (S.tiled_by_default(Cin, feats.dtype, rb.cap_out) or getattr(rb, "_sorted", None) is not None):
                    # ... and on the tile rulebook (or the class-sorted sweep) where the forward ran on it
                    dx = S.conv_forward(grad_out, wt, rb, n_in_dev, ranked=True)
                else:
                    dx = S.conv_dgrad(grad_out, wt, rb.nbr, n_in_dev, feats.shape[0], pretransposed=True)
            else:
                if getattr(rb, "_nbr_t", None) is None or rb._nbr_t.shape[1] != feats.shape[0]:
                    rb._nbr_t = S.rulebook_transpose(rb, n_out_dev, feats.shape[0])   # (kept with the rulebook: one per layer)
                if ctx.mirror == 3:
                    dx = S.conv_dgrad(grad_out, ctx.w_mirror, rb._nbr_t, n_in_dev, feats.shape[0], pretransposed=True)
                else:
                    dx = S.conv_dgrad(grad_out, wp, rb._nbr_t, n_in_dev, feats.shape[0])
        if ctx.needs_input_grad[1]:
            dw = S.conv_wgrad(feats, grad_out, rb, n_out_dev, Cin, Cout, module_shape=weight.shape).to(weight.dtype)   # (f32: no copy)
        return dx, dw, None, None, None, None, None, None


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 algo=None, fp32_accum=None, name=None):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed and not inverse
        assert _triple(dilation) == [1, 1, 1], "dilation is not used on this path"
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.dilation = _triple(dilation)
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._packed = {}
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, subm={self.subm}, indice_key={self.indice_key}")

    def packed_weight(self, dtype):
        """(K, Cout, Cin) contiguous copy in `dtype`, refreshed when the parameter changes (f32 copies of the shapes the
        f32 MFMA kernel covers are stored in that kernel's channel order: sparse.pack_weight(mfma_f32=True))."""
        key = (dtype, self.weight.device)
        ver = (self.weight._version, self.weight.data_ptr())
        hit = self._packed.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, S.pack_weight(self.weight, dtype, mfma_f32=True))
            self._packed[key] = hit
        return hit[1]

    def pad_channels(self, dtype):
        """zero channels appended to the input of a 16-bit TRAINING forward so that the layer runs on the MFMA kernels (0: none)"""
        if dtype not in (torch.float16, torch.bfloat16) or self.in_channels >= 16 or not kvol_is_27(self.kernel_size):
            return 0
        to = S.mfma_pad_channels(self.in_channels, self.out_channels)
        return to - self.in_channels if to else 0

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Checkpoints written with spconv 1.x hold the weight as (kD, kH, kW, Cin, Cout); the reference's loader adapts them
        (detector3d_template.py:401-433: permute(4, 0, 1, 2, 3) gives this module's (Cout, kD, kH, kW, Cin)).  A plain
        load_state_dict() of such a checkpoint does the same here instead of failing on the shape."""
        key = prefix + "weight"
        w = state_dict.get(key)
        if isinstance(w, torch.Tensor) and w.dim() == 5 and tuple(w.shape) != tuple(self.weight.shape):
            if tuple(w.shape) == (*self.weight.shape[1:4], self.weight.shape[4], self.weight.shape[0]):
                state_dict[key] = w.permute(4, 0, 1, 2, 3).contiguous()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def forward(self, input: SparseConvTensor):
        assert isinstance(input, SparseConvTensor)
        with_grad = torch.is_grad_enabled() and (self.weight.requires_grad or input.features.requires_grad)
        feats = input.features if with_grad else input.features.detach()
        if torch.is_autocast_enabled():
            # the reference trains under torch.cuda.amp (tools/train_utils/train_utils.py:172): spconv then runs its
            # features and weights in the autocast dtype (fp16) with fp32 accumulation — same here
            feats = feats.to(torch.get_autocast_gpu_dtype())
        if feats.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            feats = feats.to(torch.bfloat16)
        feats = feats.contiguous()
        w = None if with_grad else self.packed_weight(feats.dtype)
        n_dev = input.n_dev()

        # rows in rank-grid order on both sides of a SubM layer behind a strided one: the window / tile-rulebook kernels
        ranked = bool(self.subm and input.rows_ranked)

        # a 16-bit layer of fewer than 16 input channels in a training forward (conv_input under AMP: 5 -> 16 in fp16): the features
        # and the weight are zero-padded to 16 channels and the layer runs on the MFMA kernels, forward and weight gradient (the
        # padding's own backward slices the gradient back) — the thread-per-element kernels these shapes otherwise take cost
        # 0.34 + 0.39 ms per step at the shipped training configuration, the 16 -> 16 MFMA ones 0.03 + 0.05
        # (which layers: derived from the channel pairs the matrix kernels cover — sparse.mfma_pad_channels —, not a literal 16 -> 16:
        #  a 4 / 5 -> 32 first layer pads to 16 -> 32 the same way)
        pad_c = self.pad_channels(feats.dtype) if (with_grad and feats.is_cuda) else 0
        weight = self.weight
        if pad_c:
            feats = torch.nn.functional.pad(feats, (0, pad_c))
            weight = torch.nn.functional.pad(weight, (0, pad_c))

        def run(rb, n_out_dev, rows=None):
            if with_grad and pad_c:
                return SparseConvFunction.apply(feats, weight, rb, n_out_dev, n_dev, ranked, rows, None)
            if with_grad:
                pre = self.__dict__.get("_fnp_prepack")   # (version, data_ptr, dtype, mode, packed, mirror): see prepack_weights
                if pre is not None and (pre[0] != self.weight._version or pre[1] != self.weight.data_ptr()):
                    pre = None
                return SparseConvFunction.apply(feats, self.weight, rb, n_out_dev, n_dev, ranked, rows, None if pre is None else pre[2:])
            return S.conv_forward(feats, w, rb, n_out_dev, ranked=ranked,
                                  out=None if rows is None else torch.empty((rows, self.out_channels), dtype=feats.dtype, device=feats.device))

        if self.subm:
            rb = input.find_indice_pair(self.indice_key)
            if rb is None or rb.K != self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]:
                ch = self.in_channels if self.in_channels == self.out_channels else 0
                # the 128 -> 128 layers of a large enough stage sweep their rows class by class, forward and data gradient
                # (the fused engine's kernel: same values as the plain sweep)
                srt = bool(ranked and ch and kvol_is_27(self.kernel_size) and S.sorted_by_default(ch, ch, feats.dtype, feats.shape[0]))
                rb = S.rulebook_subm(input.indices, n_dev, input.rank_grid(), self.kernel_size,
                                     tile_channels=ch if ranked and ch and S.tiled_by_default(ch, feats.dtype, feats.shape[0]) else None,
                                     masks=srt)
                if srt:
                    S.classsort(rb, n_dev, ch)
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = rb
            out_feats = run(rb, n_dev)
            if out_feats.shape[0] != feats.shape[0]:
                out_feats = out_feats[: feats.shape[0]]
            if self.bias is not None:
                out_feats = out_feats + self.bias.to(out_feats.dtype)
            out = SparseConvTensor(out_feats, input.indices, input.spatial_shape, input.batch_size, input.grid,
                                   input.voxel_num, input.indice_dict, input.benchmark, n_dev, input._rank_grid)
            return out
        # strided: the output site count is data dependent -> one host sync, like spconv itself — unless the rulebook was
        # asked for ahead (prefetch(): its count came back on a side stream while the layers before this one were enqueued)
        n_in = feats.shape[0]
        pf = input.indice_dict.pop(("prefetch", id(self)), None)
        if pf is not None and pf[3] == input.indices.data_ptr() and pf[4] == n_in:
            rb, cap_out = pf[0], pf[0].cap_out
            pf[1].synchronize()
            n_out = int(pf[2][0])
        else:
            rb, cap_out = self._strided_rulebook(input.indices, n_dev, input.rank_grid(), input.spatial_shape, input.batch_size, n_in)
            n_out = int(rb.out_n.item())
        assert n_out <= cap_out
        nxt = getattr(self, "_fnp_next", None)
        if nxt is not None and n_out > 0 and with_grad:
            # the NEXT strided layer's rulebook needs this layer's output coordinates only: enqueue it before this layer's own
            # convolution, so that its count is on the host long before that layer's forward asks for it
            nxt.prefetch(rb.out_indices[:n_out], rb.out_n, rb.out_grid, rb.out_shape, input.batch_size, input.indice_dict)
        out_feats = run(rb, rb.out_n, rows=max(n_out, 1))
        if n_out == 0:
            out_feats = out_feats[:0]
        if self.bias is not None:
            out_feats = out_feats + self.bias.to(out_feats.dtype)
        if self.indice_key is not None:
            # (with the input side: SparseInverseConv3d of the same indice_key maps back onto exactly these sites)
            rb._in_side = (input.indices, list(input.spatial_shape), n_dev, input._rank_grid)
            input.indice_dict[self.indice_key] = rb
        return SparseConvTensor(out_feats, rb.out_indices[:n_out], rb.out_shape, input.batch_size, input.grid,
                                input.voxel_num, input.indice_dict, input.benchmark, rb.out_n, rb.out_grid)


    def _strided_rulebook(self, indices, n_dev, rank_grid, spatial_shape, batch_size, n_in):
        kvol = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        out_shape = [(spatial_shape[d] + 2 * self.padding[d] - self.kernel_size[d]) // self.stride[d] + 1 for d in range(3)]
        cells = batch_size * out_shape[0] * out_shape[1] * out_shape[2]
        cap_out = max(1, min(n_in * kvol, cells))
        return S.rulebook_strided(indices, n_dev, rank_grid, self.kernel_size, self.stride, self.padding, cap_out), cap_out

    def prefetch(self, indices, n_dev, rank_grid, spatial_shape, batch_size, indice_dict):
        """Build this strided layer's rulebook NOW for the input whose coordinates are `indices` (it depends on nothing else) and
        bring its output count to the host on a side stream; forward() of that input then finds it in `indice_dict` and does
        not drain the main stream for the count.  (The training step is bound by the host: each drained sync left the GPU
        idle for ~0.17 ms while the next layers were being enqueued.)"""
        if self.subm or not indices.is_cuda or not PREFETCH:
            return
        dev = indices.device
        rb, _ = self._strided_rulebook(indices, n_dev, rank_grid, spatial_shape, batch_size, indices.shape[0])
        side = _side_stream(dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        side.wait_event(ready)
        slot, host = _pinned_word()     # (a word of this prefetch's own: module replicas and forwards in flight never share one)
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            host.copy_(rb.out_n, non_blocking=True)
            done.record(side)
        with _PIN_LOCK:
            _PIN_BUSY[slot] = done
        rb.out_n.record_stream(side)
        indice_dict[("prefetch", id(self))] = (rb, done, host, indices.data_ptr(), indices.shape[0])


def prepack_weights(convs_and_dtypes):
    """Pack the weights of several SparseConvolution modules (and the slabs their data gradients read) in ONE launch per dtype,
    ahead of their forwards: [(module, dtype)] -> each module remembers the result until its weight changes.  The training
    forward is bound by the host: a pack launch per layer was 14 us of it, 21 times."""
    groups = {}
    for conv, dtype in convs_and_dtypes:
        w = conv.weight
        if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            continue
        if conv.pad_channels(dtype):
            continue    # (its forward pads features and weight and packs the padded weight itself: a pack of the bare one is dead work)
        odd = all(int(v) & 1 for v in conv.kernel_size)
        groups.setdefault(dtype, []).append((conv, 2 if (conv.subm and odd) else 3))
    for dtype, items in groups.items():
        for i in range(0, len(items), 32):
            part = items[i:i + 32]
            outs = S.pack_weights_train([(c.weight, mode) for c, mode in part], dtype)
            for (c, mode), (p, m) in zip(part, outs):
                c.__dict__["_fnp_prepack"] = (c.weight._version, c.weight.data_ptr(), dtype, mode, p, m)


PREFETCH = True   # (tests switch it off to compare with the synchronous path)
_SIDE = {}
_PIN_RING, _PIN_NEXT, _PIN_LOCK = None, [0], threading.Lock()
_PIN_BUSY = {}      # slot -> the event recorded behind the copy that last wrote it


def _pinned_word():
    """(slot, one int32 of pinned host memory) out of a ring of 256 (a forward has at most four prefetches in flight).  The
    ring index is taken under a lock (DataParallel replicas and loader threads prefetch concurrently), and a slot that comes
    round again waits for the copy that last wrote it — consumed long ago in any sane schedule; the wait makes it certain."""
    global _PIN_RING
    with _PIN_LOCK:
        if _PIN_RING is None:
            _PIN_RING = torch.empty((256,), dtype=torch.int32, pin_memory=True)
        i = _PIN_NEXT[0]
        _PIN_NEXT[0] = (i + 1) % 256
        old = _PIN_BUSY.pop(i, None)
    if old is not None:
        old.synchronize()
    return i, _PIN_RING[i:i + 1]


def end_of_backbone_forward(convs, indice_dict=None):
    """What prepack_weights / prefetch left on the modules is state of ONE backbone forward: the packed and mirrored slabs
    (a later call of a layer outside the backbone, after an in-place edit through .data that does not bump the parameter's
    version, must repack), the link to the next strided layer, and rulebooks prefetched for a layer that never ran."""
    for c in convs:
        c.__dict__.pop("_fnp_prepack", None)
        c.__dict__.pop("_fnp_next", None)
    if indice_dict is not None:
        for k in [k for k in indice_dict if isinstance(k, tuple) and k and k[0] == "prefetch"]:
            del indice_dict[k]


def _side_stream(device):
    key = str(device)
    if key not in _SIDE:
        # (a stream SEEN to run beside the caller's: torch's first pool streams share the default stream's hardware queue in a fresh
        #  process, and the prefetched rulebooks would be built behind the convolutions instead of beside them; sparse.concurrent_streams.
        #  Not while a stream is capturing: the test synchronises.)
        from .. import sparse as _S
        if torch.cuda.is_current_stream_capturing():
            return torch.cuda.Stream(device)
        _SIDE[key] = _S.concurrent_streams(device, 1, beside=[torch.cuda.current_stream(device)])[0]
    return _SIDE[key]


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key, algo=algo, fp32_accum=fp32_accum, name=name)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, False,
                         indice_key=indice_key, algo=algo, fp32_accum=fp32_accum, name=name)


class SparseInverseConv3d(SparseConvolution):
    """spconv.SparseInverseConv3d (post_act_block conv_type 'inverseconv', spconv_backbone.py:16-17; used by
    pcdet/models/backbones_3d/spconv_unet.py's up-sampling path only): the inverse of the SparseConv3d that shares its
    indice_key.  It runs on THAT layer's indice pairs with the two sides swapped — out[i] = sum over the pairs (k, i, o) of
    in[o] W_k — so its output sites are exactly the input sites of the paired layer (same order, same spatial shape), and it
    has its own weight (out_channels, kD, kH, kW, in_channels).  Inference only (the UNet's training path is outside SURVEY
    section 8): the transposed rulebook is the one the paired layer's data gradient uses (fnp_rulebook_transpose), the
    convolution the ordinary forward kernel on it."""

    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, 1, 0, 1, 1, bias, False, indice_key=indice_key, algo=algo,
                         fp32_accum=fp32_accum, name=name)
        self.inverse = True

    def forward(self, input: SparseConvTensor):
        assert isinstance(input, SparseConvTensor)
        rb = input.find_indice_pair(self.indice_key)
        if rb is None or getattr(rb, "_in_side", None) is None or rb.nbr is None:
            raise ValueError(f"SparseInverseConv3d(indice_key={self.indice_key!r}): no SparseConv3d with this indice_key has run on the way here")
        if torch.is_grad_enabled() and (self.weight.requires_grad or input.features.requires_grad):
            raise NotImplementedError("SparseInverseConv3d: inference only (training the UNet decoder is outside the hot path)")
        kvol = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        assert rb.K == kvol, "kernel size differs from the paired convolution's"
        in_idx, in_shape, n_in_dev, in_grid = rb._in_side
        feats = input.features.detach()
        if feats.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            feats = feats.to(torch.bfloat16)
        feats = feats.contiguous()
        cap_in = max(in_idx.shape[0], 1)
        nbr_t = getattr(rb, "_nbr_t", None)
        if nbr_t is None or nbr_t.shape[1] != cap_in:
            nbr_t = rb._nbr_t = S.rulebook_transpose(rb, rb.out_n, cap_in)
        rbt = S.Rulebook(nbr=nbr_t, K=rb.K, cap_out=cap_in, geom=rb.geom)
        out_feats = S.conv_forward(feats, self.packed_weight(feats.dtype), rbt, n_in_dev, tile=False)[: in_idx.shape[0]]
        if self.bias is not None:
            out_feats = out_feats + self.bias.to(out_feats.dtype)
        return SparseConvTensor(out_feats, in_idx, in_shape, input.batch_size, input.grid, input.voxel_num, input.indice_dict,
                                input.benchmark, n_in_dev, in_grid)
