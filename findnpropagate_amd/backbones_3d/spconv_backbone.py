"""VoxelResBackBone8x / VoxelBackBone8x on the MI355X sparse-conv kernels.

Mirrors pcdet/models/backbones_3d/spconv_backbone.py: same constructor signature, module tree
(hence the same state_dict keys: 'conv_input.0.weight', 'conv1.0.conv1.weight',
'conv2.0.0.weight', '...bn1.running_mean', ...), same `forward(batch_dict)` contract
(:243-295).  Two execution paths:

  * module path  — module by module through findnpropagate_amd.spconv (unfused BN/ReLU in torch,
    one host sync per strided conv, like spconv itself).  Used when `self.training`.
  * fused path   — eval mode.  One sync-free stream of HIP launches for the whole backbone:
    rank-grid rulebooks, implicit-GEMM convs with BatchNorm(eval)+residual+ReLU folded into
    the epilogue, bf16 storage / fp32 accumulate on MFMA (or all-f32 validation mode), a
    single host sync at the end to size the returned tensors.  `forward_points` additionally
    fuses voxelisation + MeanVFE in front (points never leave the device).
"""
from functools import partial

import os
import torch
import torch.nn as nn

from .. import sparse as S
from .. import spconv


def replace_feature(out, new_features):
    """pcdet/utils/spconv_utils.py:32-38."""
    return out.replace_feature(new_features)


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0,
                   conv_type='subm', norm_fn=None):
    """spconv_backbone.py:8-27."""
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   bias=False, indice_key=indice_key)
    elif conv_type == 'inverseconv':
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    """spconv_backbone.py:30-67."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, bias=None, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        if bias is None:
            bias = norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out = replace_feature(out, self.bn1(out.features))
        out = replace_feature(out, self.relu(out.features))
        out = self.conv2(out)
        out = replace_feature(out, self.bn2(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = replace_feature(out, out.features + identity.features)
        out = replace_feature(out, self.relu(out.features))
        return out


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class _BackboneBase(nn.Module):
    def _common_init(self, model_cfg, grid_size):
        self.model_cfg = model_cfg
        self.sparse_shape = [int(v) for v in (list(grid_size[::-1]))]
        self.sparse_shape[0] += 1  # grid_size[::-1] + [1, 0, 0], spconv_backbone.py:191
        # storage / matrix-instruction dtype of the engine: 'bf16' (default), 'fp16' (the reference's AMP mode) — both
        # with fp32 accumulation on v_mfma_f32_16x16x32 — or 'fp32' (v_mfma_f32_16x16x4_f32: the reference's default
        # precision, bit-comparable with the CPU oracle, BASELINE.json's 1e-4 mode)
        self.fnp_dtype = str(_cfg_get(model_cfg, 'FNP_DTYPE', 'bf16')).lower()
        assert self.fnp_dtype in ('bf16', 'fp16', 'fp32', 'bf16x3'), self.fnp_dtype
        # dtype of what crosses the reference boundary (encoded_spconv_tensor, multi_scale_3d_features): the reference
        # contract is float32 (its BaseBEVBackbone / heads are f32 modules), so that is the default whatever the
        # engine computes in; 'native' hands out the engine's own storage dtype (bf16 / fp16) without a cast
        self.fnp_out_dtype = str(_cfg_get(model_cfg, 'FNP_OUT_DTYPE', 'fp32')).lower()

    def _act_dtype(self):
        # (bf16x3: f32 tensors between the layers and at the boundary; the engine keeps their (hi, lo) bf16 split beside them)
        return {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16, 'bf16x3': torch.float32}[self.fnp_dtype]

    def _boundary_dtype(self):
        return {'fp32': torch.float32, 'float32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16,
                'native': None}[self.fnp_out_dtype]

    def _pack_outputs(self, batch_dict, out, x1, x2, x3, x4):
        dt = self._boundary_dtype()
        if dt is not None:
            out, x1, x2, x3, x4 = [t if t.features.dtype == dt else t.replace_feature(t.features.to(dt)) for t in (out, x1, x2, x3, x4)]
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {'x_conv1': x1, 'x_conv2': x2, 'x_conv3': x3, 'x_conv4': x4}})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict


class VoxelBackBone8x(_BackboneBase):
    """spconv_backbone.py:70-181 (plain variant; module path only)."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self._common_init(model_cfg, grid_size)
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key='subm1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'))
        last_pad = _cfg_get(model_cfg, 'last_pad', 0)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'), norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}

    def forward(self, batch_dict):
        return _module_forward(self, batch_dict)


def _bn_relu(bn, relu, x, act, residual=None):
    """features of `x` through BatchNorm1d -> (+ residual) -> ReLU, stored in `act`.  Training-mode BatchNorm takes the
    fused kernels (spconv/norm.py: one autograd node, two passes per direction); a frozen (eval) one inside a training
    run, or a shape the kernels do not take, goes through the torch modules in f32 like the reference."""
    from ..spconv import norm as N
    f = x.features
    if N.fusable(bn) and relu is not None and f.is_cuda and f.dtype in (act, torch.float32):
        res = None if residual is None else residual.to(f.dtype)
        return N.bn_act(f, x.n_dev(), bn, residual=res, relu=True).to(act)   # (f32 in: the first layer's conv output)
    y = bn(f.float())
    if residual is not None:
        y = y.to(act).float() + residual.float()
    return (relu(y) if relu is not None else y).to(act)


def _module_forward(self, batch_dict):
    voxel_features, voxel_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
    batch_size = batch_dict['batch_size']
    act = self._act_dtype()
    if torch.is_autocast_enabled() and voxel_features.is_cuda:
        # under torch.cuda.amp (tools/train_utils/train_utils.py:172) the reference's activations ARE the autocast dtype from
        # layer to layer: spconv's convolutions write fp16, nn.BatchNorm1d / ReLU keep the dtype they are given.  Same here —
        # and the fused BatchNorm kernels then run (round 5: with `act` left at FNP_DTYPE every layer went conv fp16 -> torch
        # batch_norm in f32 -> bf16 -> fp16 again, 8 of the 19.5 ms of the step at the shipped configuration)
        act = torch.get_autocast_dtype('cuda')
    x_in = spconv.SparseConvTensor(features=voxel_features.float().contiguous(), indices=voxel_coords.int().contiguous(),
                                   spatial_shape=self.sparse_shape, batch_size=batch_size)
    # the strided layers' rulebooks are asked for ahead of the layers before them (spconv/conv.py prefetch): the first one from
    # the input coordinates right here, each further one by the strided layer before it
    prepacked = x_in.features.is_cuda and torch.is_grad_enabled()
    if prepacked:
        # every layer's packed weights (and its data gradient's slabs) in one launch per dtype
        from ..spconv import conv as _C
        conv0 = self.conv_input[0]
        auto = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else None
        convs = [m for m in self.modules() if isinstance(m, spconv.SparseConvolution) and m.weight.requires_grad]
        _C.prepack_weights([(m, auto if auto is not None else (torch.float32 if m is conv0 else act)) for m in convs])
        chain = _strided_chain(self)
        for a, b in zip(chain, chain[1:] + [None]):
            object.__setattr__(a, "_fnp_next", b)   # (not a registered submodule: parameter names and state_dict stay the reference's)
        if chain and x_in.indices.shape[0] > 0:
            chain[0].prefetch(x_in.indices, x_in.n_dev(), x_in.rank_grid(), x_in.spatial_shape, batch_size, x_in.indice_dict)
    # first conv consumes f32 point features; activations then live in `act`
    try:
        conv0 = self.conv_input[0]
        x = conv0(x_in)
        x = x.replace_feature(_bn_relu(self.conv_input[1], self.conv_input[2], x, act))
        x_conv1 = _seq_forward(self.conv1, x, act)
        x_conv2 = _seq_forward(self.conv2, x_conv1, act)
        x_conv3 = _seq_forward(self.conv3, x_conv2, act)
        x_conv4 = _seq_forward(self.conv4, x_conv3, act)
        out = _seq_forward(self.conv_out, x_conv4, act)
    finally:
        # (also when a layer raised: the autograd nodes hold their own slabs; nothing stale stays on the modules)
        if prepacked:
            _C.end_of_backbone_forward(convs + [c for c in chain if c not in convs], x_in.indice_dict)
    return self._pack_outputs(batch_dict, out, x_conv1, x_conv2, x_conv3, x_conv4)


def _strided_chain(self):
    """the strided convolutions of the backbone in forward order (the first sparse module of conv2, conv3, conv4, conv_out)"""
    out = []
    for name in ("conv2", "conv3", "conv4", "conv_out"):
        m = getattr(self, name, None)
        while isinstance(m, spconv.SparseSequential) and len(m._modules):
            m = next(iter(m._modules.values()))
        if isinstance(m, spconv.SparseConv3d) and not m.subm:
            out.append(m)
    return out


def _seq_forward(seq, x, act):
    """SparseSequential forward with dense modules evaluated in f32 and stored back in `act`."""
    mods = list(seq._modules.values())
    skip = False
    for i, m in enumerate(mods):
        if skip:                      # the ReLU folded into the BatchNorm before it
            skip = False
            continue
        if isinstance(m, nn.BatchNorm1d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
            x = x.replace_feature(_bn_relu(m, mods[i + 1], x, act))
            skip = True
            continue
        if isinstance(m, spconv.SparseSequential):
            x = _seq_forward(m, x, act)
        elif isinstance(m, SparseBasicBlock):
            identity = x
            o = m.conv1(x)
            o = o.replace_feature(_bn_relu(m.bn1, m.relu, o, act))
            o = m.conv2(o)
            x = o.replace_feature(_bn_relu(m.bn2, m.relu, o, act, residual=identity.features))
        elif isinstance(m, spconv.SparseModule):
            x = m(x)
        else:
            x = x.replace_feature(m(x.features.float()).to(act))
    return x


class VoxelResBackBone8x(_BackboneBase):
    """spconv_backbone.py:184-295."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self._common_init(model_cfg, grid_size)
        use_bias = _cfg_get(model_cfg, 'USE_BIAS', None)
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.input_channels = input_channels
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(
            SparseBasicBlock(16, 16, bias=use_bias, norm_fn=norm_fn, indice_key='res1'),
            SparseBasicBlock(16, 16, bias=use_bias, norm_fn=norm_fn, indice_key='res1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            SparseBasicBlock(32, 32, bias=use_bias, norm_fn=norm_fn, indice_key='res2'),
            SparseBasicBlock(32, 32, bias=use_bias, norm_fn=norm_fn, indice_key='res2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            SparseBasicBlock(64, 64, bias=use_bias, norm_fn=norm_fn, indice_key='res3'),
            SparseBasicBlock(64, 64, bias=use_bias, norm_fn=norm_fn, indice_key='res3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 128, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            SparseBasicBlock(128, 128, bias=use_bias, norm_fn=norm_fn, indice_key='res4'),
            SparseBasicBlock(128, 128, bias=use_bias, norm_fn=norm_fn, indice_key='res4'))
        last_pad = _cfg_get(model_cfg, 'last_pad', 0)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'), norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 128}
        self._engine = None

    # ---------------------------------------------------------------- reference contract
    def forward(self, batch_dict):
        """batch_dict: batch_size, voxel_features (M,C), voxel_coords (M,4) [b,z,y,x] ->
        encoded_spconv_tensor (+stride 8), multi_scale_3d_features, multi_scale_3d_strides."""
        if self.training:
            return _module_forward(self, batch_dict)
        feats = batch_dict['voxel_features']
        coords = batch_dict['voxel_coords']
        batch_size = int(batch_dict['batch_size'])
        dev = feats.device
        feats = feats.detach().float().contiguous()
        coords = coords.int().contiguous()
        n = S.device_scalar(feats.shape[0], dev)
        # (conv_out's epilogue writes the boundary dtype directly: no cast pass over the encoded tensor)
        res = self.engine().run(feats, coords, n, batch_size, grid1=None, final_dtype=self._boundary_dtype())
        return self._pack_outputs(batch_dict, res['out'], res['x_conv1'], res['x_conv2'], res['x_conv3'], res['x_conv4'])

    # ---------------------------------------------------------------- fused device path
    def engine(self):
        if self._engine is None:
            self._engine = FusedResBackbone(self)
        return self._engine

    def forward_points(self, points, batch_offsets, batch_size, voxel_cfg, sync=True):
        """points (N,C) f32 device (scenes concatenated), batch_offsets (B+1,) int32 device.
        Voxelise + MeanVFE + backbone without leaving the device.  Returns the same dict of
        SparseConvTensors as forward() produces plus 'voxel_coords', 'voxel_num_points',
        'voxel_features'."""
        return self.engine().run_points(points, batch_offsets, batch_size, voxel_cfg, sync=sync)

    def forward_points_graphed(self, points, batch_offsets, batch_size, voxel_cfg, capacity=None, probe=False):
        """forward_points replayed from a captured hipGraph (one graph launch instead of ~100 kernel
        launches: small batches are launch-bound).  Same results; the returned tensors are views of the
        graph's static buffers and are overwritten by the next call with the same (batch_size, capacity).
        capacity: point capacity of the graph (default: N rounded up to 64 Ki)."""
        return self.engine().run_points_graphed(points, batch_offsets, batch_size, voxel_cfg, capacity, probe=probe)

    def points_pipeline(self, batch_size, voxel_cfg, depth=2, capacity=65536, n_feat=5, probe=False, serial_convs=None):
        """forward_points_graphed for frames that arrive one at a time, `depth` of them in flight on their own HIP streams
        (PointsPipeline: submit / result / map).  Same results; the frame rate of a one-scene stream roughly doubles."""
        return PointsPipeline(self, batch_size, voxel_cfg, depth=depth, capacity=capacity, n_feat=n_feat, probe=probe, serial_convs=serial_convs)

    def forward_points_iter(self, batches, batch_size, voxel_cfg, depth=2, capacity=None):
        """THE BATCH PATH for a caller that has more than one batch to run (an extraction or evaluation loop, a server): an iterator
        over (points, batch_offsets) pairs -> the dicts of forward_points, in order, with `depth` batches in flight (default TWO,
        round 6: the latency-bound index kernels of one batch run under the convolutions of the other — +5 to +7 % at 128 scenes
        per batch, 2x at one scene; results identical to forward_points batch by batch).  A returned dict's tensors are views of
        a slot's static buffers: valid until `depth` more batches have been taken from the iterator — of this call or of the next one with
        the same (batch_size, depth, capacity, voxel configuration): the pipeline of the last call is kept (its slots are captured
        graphs and streams picked by test, ~0.5 s to make) and reused.
        capacity: point capacity of a slot (default: sized by the first batch, rounded up to 64 Ki points)."""
        pipe = None
        for pts, off in batches:
            if pipe is None:
                cap = int(capacity) if capacity else max(65536, (int(pts.shape[0]) + 65535) // 65536 * 65536)
                key = (int(batch_size), int(depth), cap, int(pts.shape[1]), str(pts.device), bytes(voxel_cfg))
                kept = self.__dict__.get("_iter_pipeline")
                if kept is not None and kept[0] == key and int(pts.shape[0]) <= cap:
                    pipe = kept[1]
                    while pipe.pending:          # (an iterator of the last call that was dropped half-way)
                        pipe.result()
                else:
                    pipe = PointsPipeline(self, batch_size, voxel_cfg, depth=depth, capacity=cap, n_feat=int(pts.shape[1]))
                    self.__dict__["_iter_pipeline"] = (key, pipe)
            if len(pipe.pending) == pipe.depth:
                yield pipe.result()
            pipe.submit(pts, off)
        while pipe is not None and pipe.pending:
            yield pipe.result()


X3_FUSED = os.environ.get("FNP_X3_FUSED", "1") != "0"   # bf16x3: the main product's epilogue adds the cross terms and writes the split
X3_FAST = os.environ.get("FNP_X3_FAST", "1") != "0"   # bf16x3: cross terms on the bf16-out fast kernels (0: three f32-out gather launches per layer)

# the fused engine's index chain on a side stream (FusedResBackbone._run_once): None = while a hipGraph is being captured (the
# replayed graph runs the two branches side by side, the ~4.5 us between its nodes overlap: +4 to +10 % from 1 to 64 scenes),
# not for stream launches (measured neutral within a box's +-1.5 %: back-to-back launches have no such gaps to hide);
# FNP_TWO_STREAMS=1 / 0 forces / forbids it everywhere
TWO_STREAMS = {"0": False, "1": True}.get(os.environ.get("FNP_TWO_STREAMS", ""))


class _PointsGraph:
    """Static inputs + captured forward of one (batch_size, point capacity) configuration."""
    FAR = 1.0e9   # padding points: outside every range, dropped by the voxeliser

    def __init__(self, engine, capacity, n_feat, batch_size, voxel_cfg, device, probe=False, split_index=False):
        # probe: the forward is captured as TWO graphs with the launches of the last SubM stage (the four 128 -> 128
        # convolutions of VoxelResBackBone8x: the step's dominant kernel) left out between them; replay() issues those as plain
        # launches, each bracketed with a pair of timing events when asked to.  A measurement device (bench.py: the step replayed
        # from graphs AND its dominant kernel timed with events inside the timed region, which nodes of a graph do not allow);
        # same kernels, same buffers, same values as the one-graph form.
        # A probe capture is TWO graphs, cut where the counts are final (in front of the last SubM stage): replay() copies them to
        # pinned memory between the two and records counts_event there, so the host sizes the outputs while the second graph
        # still runs (done_event marks its end).  Every other capture is one graph whose counts launch writes to the host itself
        # (host_counts, below).
        # split_index (round 6, PointsPipeline's serial-convolution mode): the voxeliser and the whole index chain — everything that
        # reads coordinates only — are a graph of their own (graph_i) in front of the other two, so that replay() can issue them on the
        # slot's stream while every slot's convolutions go, in submission order, through ONE stream shared by the pipeline: batch
        # i + 1's latency-bound index kernels run under batch i's convolutions, and two convolutions never share the CUs.
        # HOST COUNTS (round 6, every capture that is not a probe): the counts launch stores the counts into pinned host memory itself
        # and its sequence number behind them (fnp_gather_counts_host); counts() polls for that number.  No event, no copy — and no
        # cut: the convolutions are ONE graph, the last SubM stage follows the strided layer in front of it without the ~25 us a
        # one-scene forward spent between two graph launches (end of graph, counts copy, event, start of graph).  FNP_HOST_COUNTS=0:
        # the two-graph form with the copy between them.
        self.host_counts = (not probe) and os.environ.get("FNP_HOST_COUNTS", "1") != "0"
        self.counts_seq, self._replays = None, 0
        self.split_index, self.graph_i, self._index_cut = bool(split_index), None, False
        self.probe, self.deferred, self.graph_b = bool(probe), [], None
        self.counts_dev, self.counts_pin = None, None
        if self.host_counts:    # (allocated and zeroed HERE: a fill issued during the capture would be a node of the graph)
            self.counts_pin = torch.zeros((32,), dtype=torch.int32, pin_memory=True)
            self._pin_np = self.counts_pin.numpy()
            self.counts_seq = torch.zeros((1,), dtype=torch.int32, device=device)
        self.counts_event, self.done_event = torch.cuda.Event(), torch.cuda.Event()
        self._replayed = False
        self.engine, self.capacity, self.batch_size = engine, capacity, batch_size
        self.pts = torch.full((capacity, n_feat), self.FAR, dtype=torch.float32, device=device)
        self.off = torch.zeros((batch_size + 1,), dtype=torch.int32, device=device)
        self.n_prev = 0
        self.prep_key = engine._prep_key
        # warm-up (allocations of persistent grids/workspaces, lazy kernel attributes), then capture.  The warm-up is not a frame:
        # it must not count the tile gate's gather-kernel period down, and the key the graph is stored under is read AFTER it
        # (a key taken before an eager warm-up that ticked the gate went stale and forced one more recapture)
        engine._gate_hold = True
        try:
            self._body(voxel_cfg)
        finally:
            engine._gate_hold = False
        torch.cuda.synchronize(device)
        self.cap_factor = engine._graph_key()
        self.graph = torch.cuda.CUDAGraph()
        # No cyclic garbage collection while the stream is capturing: a collection that fires inside the capture and finalises
        # an OLDER graph or its pooled tensors (a replaced _PointsGraph, another test's engine) frees device memory in the
        # middle of it — an error inside a destructor, i.e. std::terminate.  Seen twice in round 3 as an intermittent
        # "Fatal Python error: Aborted" of a test run; the faulthandler trace of the second sighting ends in "Garbage-collecting"
        # under this very `with`.  (torch.cuda.graph collects BEFORE it starts capturing; nothing stops a collection during it.)
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            self.graph_b = torch.cuda.CUDAGraph()
            if self.split_index:
                self.graph_i = torch.cuda.CUDAGraph()
                self._ctx = torch.cuda.graph(self.graph_i)
            else:
                self._ctx = torch.cuda.graph(self.graph)
            self._ctx.__enter__()
            try:
                self.vox, self.res = self._body(voxel_cfg)   # (the engine calls counts_ready() where the first graph ends)
            finally:
                self._ctx.__exit__(None, None, None)
            assert self.counts_dev is not None and (self.deferred or not self.probe), "the engine did not reach its cut"
            if self.host_counts:
                self.graph_b = None     # (never begun: the capture was not cut)
            assert self._index_cut or not self.split_index, "the engine did not cut behind its index chain"
        finally:
            if gc_was_on:
                gc.enable()

    def stage(self, points, batch_offsets):
        """a frame into the static inputs (one launch on the current stream: the points, the padding behind them as far as the previous
        frame reached, the scene offsets)"""
        n = int(points.shape[0])
        if (points.is_cuda and points.dtype == torch.float32 and points.is_contiguous() and batch_offsets.is_cuda
                and batch_offsets.dtype == torch.int32 and batch_offsets.is_contiguous()):
            S.stage_points(points, batch_offsets, self.pts, self.off, self.n_prev, self.FAR)
        else:
            self.pts[:n].copy_(points, non_blocking=True)
            if self.n_prev > n:
                self.pts[n:self.n_prev].fill_(self.FAR)
            self.off.copy_(batch_offsets, non_blocking=True)
        self.n_prev = n

    def index_ready(self):
        """called by the engine between its index chain and its first convolution (split_index captures only; the chain must be on
        the capture stream itself: no second branch): ends graph_i, begins the first convolution graph in the same pool"""
        if not self.split_index or self._index_cut:
            return
        self._index_cut = True
        self._ctx.__exit__(None, None, None)
        self._ctx = torch.cuda.graph(self.graph, pool=self.graph_i.pool())
        self._ctx.__enter__()

    def counts_host(self):
        """(pinned tensor, device sequence word) the engine's counts launch writes to, or None (two-graph form: replay() copies)"""
        return (self.counts_pin, self.counts_seq) if self.host_counts else None

    def counts_ready(self, counts_dev):
        """called by the engine where the counts are final (every forked stream rejoined): ends the first graph's capture and
        begins the second's on the same capture stream and memory pool (host counts: nothing to cut)"""
        self.counts_dev = counts_dev
        if self.host_counts:
            return
        self._ctx.__exit__(None, None, None)
        self._ctx = torch.cuda.graph(self.graph_b, pool=(self.graph_i if self.split_index else self.graph).pool())
        self._ctx.__enter__()

    def replay(self, profile=None, conv_stream=None):
        """one forward.  probe graphs: first half, the deferred launches (each between two timing events appended to `profile`
        as (tag, start, end) when a list is given), second half.  conv_stream (split_index captures): the stream the convolution
        graphs and everything behind them are issued on; the index graph goes to the current stream."""
        cur = torch.cuda.current_stream(self.pts.device)
        if self._replayed:   # (the previous replay may have been issued on another stream: the buffers are shared)
            cur.wait_event(self.done_event)
        # ... and so may an EAGER forward of the same engine (it shares the persistent grids and workspaces and returns early)
        if self.engine._last_done is not None and self.engine._last_done is not self.done_event:
            cur.wait_event(self.engine._last_done)
        self._replayed = True
        if self.split_index:
            self.graph_i.replay()
            if conv_stream is not None and conv_stream != cur:
                ev = torch.cuda.Event()
                ev.record(cur)
                conv_stream.wait_event(ev)
                with torch.cuda.stream(conv_stream):
                    self._replay_convs(profile)
                return
        self._replay_convs(profile)

    def _replay_convs(self, profile):
        self.graph.replay()
        if self.host_counts:
            self._replays += 1
            self.done_event.record()
            self.engine._last_done = self.done_event
            return
        if self.counts_pin is None:
            self.counts_pin = torch.empty((16,), dtype=torch.int32, pin_memory=True)
        self.counts_pin[:self.counts_dev.numel()].copy_(self.counts_dev, non_blocking=True)
        self.counts_event.record()
        if self.probe:
            for tag, launch in self.deferred:
                if profile is None:
                    launch()
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch()
                    e1.record()
                    profile.append((tag, e0, e1))
        self.graph_b.replay()
        self.done_event.record()
        self.engine._last_done = self.done_event     # an eager forward issued next, on any stream, waits for this replay

    def counts(self):
        """the host's copy of the last replay's counts (waits for the counts only, not for the rest of the forward)"""
        if self.host_counts:
            # the counts launch of replay number r stores r behind the counts (after them, system scope): poll for it
            # (a one-scene forward has them after ~0.2 ms: a pure spin; a 128-scene batch behind another after ~9 ms: past the first
            #  ~0.3 ms the loop sleeps between looks, so that a long wait does not hold a core)
            import time
            want, word, t0 = self._replays & 0xffffffff, self._pin_np, None
            spins = 0
            while (int(word[16]) & 0xffffffff) != want:
                spins += 1
                if spins > 3000:
                    time.sleep(5.0e-5)
                    if spins & 0x3f == 0:      # (every few milliseconds: has the device stopped?)
                        t0 = t0 or time.monotonic()
                        if self.done_event.query() and (int(word[16]) & 0xffffffff) != want:
                            raise RuntimeError("the forward ended without its counts launch (sequence %d, expected %d)" % (int(word[16]), want))
                        if time.monotonic() - t0 > 60.0:
                            raise RuntimeError("no counts from the device after 60 s")
            return [int(v) for v in word[:self.counts_dev.numel()]]
        self.counts_event.synchronize()
        return self.counts_pin[:self.counts_dev.numel()].tolist()

    def _body(self, voxel_cfg):
        e = self.engine
        e._ensure_clean()
        e._dirty = True
        grids = e._get_grids(self.batch_size, self.pts.device)
        vox = S.voxelize(self.pts, self.off, self.batch_size, voxel_cfg, grid=grids[0], workspace=e._vox_ws)
        e._vox_ws = vox['workspace']
        res = e._run_once(vox['mean'], vox['coords'], vox['n'], self.batch_size, grids[0], sync=False, n_cells=vox['n_cells'],
                          probe=self if self.graph_b is not None else None)
        return vox, res


concurrent_streams = S.concurrent_streams     # (sparse.py: streams tested to run beside each other)


class PointsPipeline:
    """Frames that arrive ONE AT A TIME (the reference's extraction and evaluation loops run batch size 1) kept `depth` deep
    in flight: every slot owns an engine (its rank grids and workspaces), a captured hipGraph of forward_points and a HIP
    stream; submit() copies a frame into the slot's static inputs and replays its graph on the slot's stream, result() hands
    out the oldest frame's tensors.  A one-scene forward leaves most of the 256 CUs idle (it is a chain of ~65 short
    launches), so two or three frames overlap almost perfectly: the latency of a frame stays what it was, the frame rate
    multiplies.  Results are the graphed path's, bit for bit.  The returned tensors are views of the slot's static buffers:
    valid until `depth` more frames have been submitted."""

    def __init__(self, module, batch_size, voxel_cfg, depth=2, capacity=65536, n_feat=5, device=None, probe=False, serial_convs=None, streams=None):
        # probe: every slot's capture is the two-graph PROBE form of _PointsGraph (the launches of the last SubM stage issued as plain
        # launches between the two graphs); with `self.profile` a list each of them is bracketed by a pair of timing events on the
        # slot's stream, (tag, start, end) appended — how bench.py times the dominant kernel inside its pipelined timed region
        assert depth >= 1
        self.probe, self.profile = bool(probe), None
        self.module, self.batch_size, self.cfg, self.depth = module, int(batch_size), voxel_cfg, int(depth)
        self.capacity, self.n_feat = int(capacity), int(n_feat)
        self.device = device if device is not None else next(module.parameters()).device
        self.engines = [FusedResBackbone(module) for _ in range(self.depth)]
        for e in self.engines:
            # two frames in flight already put one frame's index kernels under another's convolutions; a second branch per
            # graph on top of that oversubscribes the hardware queues
            e.two_streams = self.depth == 1 or os.environ.get("FNP_PIPE_TWO", "0") == "1"

        # Which streams: ones that were SEEN to run beside each other and beside the caller's (concurrent_streams: the card's four
        # hardware queues are shared out in the order streams are first used, and two streams of one queue run one after the other).
        # tools/probe/queue_map.py, 128 scenes, one process: slots and convolutions on three queues, none the caller's 13.4 k scenes/s;
        # a slot on the caller's stream 12.7-12.8 k; the convolutions on the caller's stream 12.2 k (nothing overlaps).  With more
        # streams than queues (three slots + the convolution stream + the caller's) the CONVOLUTION stream is the one kept apart and the
        # slots double up: their index chains run one after the other anyway.  FNP_TESTED_STREAMS=0: torch's next pool streams;
        # `streams` hands in the caller's own.
        env = {"0": False, "1": True}.get(os.environ.get("FNP_PIPE_SERIAL", ""))
        self.serial_convs = self.depth > 1 and (bool(serial_convs) if serial_convs is not None else env if env is not None else self.batch_size >= 8)
        if streams is not None:     # (the caller's own: depth slot streams, then the convolution stream of the serial form)
            picked = list(streams)
            assert len(picked) == self.depth + (1 if self.serial_convs else 0)
        elif os.environ.get("FNP_TESTED_STREAMS", "1") != "0":
            caller = torch.cuda.current_stream(self.device)
            if self.serial_convs:
                conv = concurrent_streams(self.device, 1, beside=[caller])
                slots, ok = concurrent_streams(self.device, self.depth, beside=[caller] + conv, report=True)
                if not ok:      # (not enough queues: the slots only have to stay clear of the convolutions)
                    slots = concurrent_streams(self.device, min(self.depth, 2), beside=[caller] + conv)
                    slots = [slots[i % len(slots)] for i in range(self.depth)]
                picked = slots + conv
            else:
                picked = concurrent_streams(self.device, self.depth, beside=[caller])
        else:
            picked = [torch.cuda.Stream(self.device) for _ in range(self.depth + (1 if self.serial_convs else 0))]
        self.streams = picked[:self.depth]
        # SERIAL CONVOLUTIONS (round 6).  With `depth` whole forwards in flight on `depth` streams the hardware interleaves them as
        # it likes: two 128 -> 128 launches share the CUs and each takes 1.8x as long (0.78 -> 1.37 ms at 128 scenes).  The gain of
        # the pipeline is elsewhere — a batch's voxeliser and index chain (latency- and atomics-bound, ~2 ms of a 10 ms step, a few
        # waves per CU) running under ANOTHER batch's convolutions — so each slot's capture is cut behind its index chain
        # (_PointsGraph split_index): the index graph goes to the slot's stream, the convolution graphs of ALL slots to one
        # stream in submission order.  Default: on for batches of >= 8 scenes (convolutions that fill the chip); a one-scene
        # stream, whose convolutions leave most CUs idle, keeps the free-for-all.  FNP_PIPE_SERIAL=0 / 1 forces.
        # Measured at 128 scenes (one box, bench.py): one batch at a time 12.50 k scenes/s; serial, two / three in flight 12.79-13.07 /
        # 12.84-13.04 k with the dominant kernel at 0.76-0.78 ms; free-for-all 13.10-13.12 / 13.42 k with it at 1.37 ms.  A kernel
        # trace of the serial form (tools/pipe_trace.sh) shows a gapless convolution chain (0.24 ms of gaps per step) whose early
        # layers run slower under the other batch's index kernels (16 -> 16 123 -> 195 us, 16 -> 32 244 -> 406, 32 -> 64 390 -> 570,
        # tile32 263 -> 340; the wide layers within 3 %): 8.76 ms of convolutions take 9.53.  Starting a batch's index chain only
        # where the batch before it reaches its 128 -> 128 launches (so that it meets matrix-bound kernels only) is WORSE, 12.2 k:
        # those launches hold every CU's registers and LDS, the index kernels crawl (5 ms of kernel time) and the chain waits 1.7 ms.
        # (measured, round 6: the convolution stream at a higher dispatch priority — torch.cuda.Stream(priority=-1) — starves the index
        #  chains it waits for: 13.0-13.3 k -> 12.2-12.4 k scenes/s)
        self.conv_stream = picked[self.depth] if self.serial_convs else None
        if self.serial_convs:
            for e in self.engines:
                e.two_streams = False      # (the index chain must sit on the capture stream itself: it becomes a graph of its own)
        self.slots = [None] * self.depth      # _PointsGraph per slot, captured on first use
        self.pending = []                     # (slot, event, host counts, inputs) in submission order
        self.next_slot = 0

    def _graph(self, d):
        e = self.engines[d]
        e.prepare()
        e._gate_tick()        # (a replayed frame counts the tile gate's period down like an eager one)
        g = self.slots[d]
        if g is None or g.cap_factor != e._graph_key() or g.prep_key != e._prep_key:
            torch.cuda.synchronize(self.device)   # (capture: nothing else of this pipeline may be in flight)
            g = _PointsGraph(e, self.capacity, self.n_feat, self.batch_size, self.cfg, self.device, probe=self.probe, split_index=self.serial_convs)
            self.slots[d] = g
        return g

    def submit(self, points, batch_offsets):
        """enqueue one frame (points (N, C) f32, batch_offsets (B + 1,) i32, both on the device); returns at once.  With
        `depth` frames already in flight the oldest must be taken with result() first."""
        assert len(self.pending) < self.depth, "pipeline full: call result() first"
        n = points.shape[0]
        assert n <= self.capacity and points.shape[1] == self.n_feat
        d = self.next_slot
        self.next_slot = (d + 1) % self.depth
        g = self._graph(d)
        st = self.streams[d]
        st.wait_stream(torch.cuda.current_stream(self.device))   # the caller produced `points` on its own stream
        with torch.cuda.stream(st):
            g.stage(points, batch_offsets)
            # (copies the counts to pinned memory between its two convolution graphs; done_event at the end)
            g.replay(self.profile if self.probe else None, conv_stream=self.conv_stream)
        points.record_stream(st)
        batch_offsets.record_stream(st)
        self.pending.append((d, None, (points, batch_offsets)))

    def result(self):
        """the oldest frame in flight: the dict of forward_points_graphed.  Waits for that frame only."""
        d, _, (points, batch_offsets) = self.pending.pop(0)
        g, e = self.slots[d], self.engines[d]
        counts = g.counts()
        ev = g.done_event
        overflow = e._ell_overflow(counts, g.res['ell_used'], g.res['caps'][0])
        e._check_aborts(counts.pop())
        caps = g.res['caps']
        overflow = overflow or any(counts[l] > caps[l] for l in range(1, 5))
        torch.cuda.current_stream(self.device).wait_event(ev)   # the caller's stream reads the slot's buffers next
        if overflow:
            # a capacity was too small for this frame: the engine's own loop grows it and recaptures (synchronously; rare)
            ev.synchronize()
            for pd, _, _ in self.pending:
                self.slots[pd].done_event.synchronize()
            for l in range(1, 5):
                if counts[l] > caps[l]:
                    e.cap_factor[l - 1] = max(e.cap_factor[l - 1] * 2.0, counts[l] * 1.25 / caps[0])
            torch.cuda.synchronize(self.device)
            for gr in e._get_grids(self.batch_size, self.device):   # rows beyond a capacity were never emitted: the sparse clear missed their cells
                gr.zero_()
            e._dirty = False
            self.slots[d] = None
            return e.run_points_graphed(points, batch_offsets, self.batch_size, self.cfg, self.capacity)
        stage, shapes = g.res['stages'], g.res['shapes']
        tensors = [spconv.SparseConvTensor(x[:counts[l]], idx[:counts[l]], shapes[l], self.batch_size, n_dev=nd)
                   for l, (x, idx, nd, _) in enumerate(stage)]
        n1 = counts[0]
        return {'x_conv1': tensors[0], 'x_conv2': tensors[1], 'x_conv3': tensors[2], 'x_conv4': tensors[3],
                'out': tensors[4], 'counts': counts, 'voxel_coords': g.vox['coords'][:n1],
                'voxel_num_points': g.vox['num_points'][:n1], 'voxel_features': g.vox['mean'][:n1]}

    def map(self, frames):
        """generator over an iterable of (points, batch_offsets): results in order, `depth` frames in flight"""
        for pts, off in frames:
            if len(self.pending) == self.depth:
                yield self.result()
            self.submit(pts, off)
        while self.pending:
            yield self.result()


_ABORTS_SEEN = 0   # last value of the library's tiled-kernel time-out counter any engine has read


class FusedResBackbone:
    """Sync-free executor of VoxelResBackBone8x in eval mode (see module docstring)."""

    def __init__(self, module: VoxelResBackBone8x):
        self.m = module
        self.act = module._act_dtype()
        # bf16x3 (round 4): the f32 result to ~1e-5 relative on the BF16 matrix pipe.  Activations travel as f32 rows plus their
        # split x = hi + lo into two bf16 tensors (fnp_split_bf16), weights (BatchNorm scale folded in) as W = W_hi + W_lo, and a
        # convolution is three launches of the bf16 kernels with f32 outputs chained through `residual`:
        #   t = lo * W_hi (+ identity);  t = hi * W_lo + t;  y = relu(hi * W_hi + shift + t)
        # (the lo * W_lo term, 2^-16 of a product, is dropped).  conv_input reads the f32 voxel means on the f32 kernel.
        self.x3 = getattr(module, 'fnp_dtype', '') == 'bf16x3'
        self._prep = None
        self._prep_key = None
        self._grids = {}
        # capacity of stage l (l = 2..5) as a multiple of the stage-1 capacity; grown on overflow
        self.cap_factor = [3.0, 2.0, 1.0, 1.0]
        # extension-record pools of the compact rulebooks (stage 1, the 16 -> 32 layer) in records per row capacity; measured
        # need on lidar scenes 0.10 / 0.01; grown on overflow like the capacities
        self.ell_pool = [0.25, 0.0625]
        self._vox_ws = None
        self._graphs = {}
        self._ell_ctr = {}
        self._side = {}
        self._counts_pin = None
        self._last_done = None
        # TILE GATE (round 4).  The tile-rulebook kernels of stages 2-3 are ahead of the gather kernels only while (nearly) every
        # neighbour of a tile sits in its LDS image: the rulebook kernel counts the 32-row groups that hold an escape entry, the
        # count comes back with the forward's one synchronisation, and a stage whose share exceeds TILE_ESC_MAX runs on the gather
        # kernels for the next TILE_REPROBE eager forwards (then the tiles are tried, and measured, again).  Single-sweep lidar:
        # ~1e-5; the 10-sweep density of transfusion_lidar.yaml: 2 % / 11 % of the groups, tiled kernels 4-8 % slower (measured).
        self.tile_off = {}            # stage index li (0: stage 2, 1: stage 3) -> forwards (eager OR replayed) left on the gather kernels
        self.tile_period = {}         # li -> length of the current gather-kernel period (doubles while every re-probe fails)
        self.tile_escape_share = {}   # last measured share per stage (diagnostics; bench.py reports it)
        self._gate_hold = False       # a _PointsGraph warm-up / capture is running: not a frame
        self.two_streams = True   # (PointsPipeline clears it for its slots when several frames are in flight: they already overlap)
        self._dirty = False      # a forward is in flight or died before its sparse clear: grids may hold stale bits
        # measurement hooks (bench.py): when `profile` is a list every conv launch is bracketed by
        # stream events, (tag, start, end) appended; `rulebook_log` receives (tag, Rulebook, n_dev)
        self.profile = None
        self.profile_only = None      # optional set of (Cin, Cout, K): bracket only these layer classes
        self.rulebook_log = None
        # hand-over time-outs of the tiled 32-channel kernel (csrc/spconv_tile.hip g_tile_aborts): the library-wide counter
        # is copied behind every forward into a device word that travels with the per-stage counts of the one host sync

    def _aborts_word(self, device):
        """enqueue a copy of the library's time-out counter; returns the (1,) int32 device tensor it lands in"""
        from .. import lib as _l
        t = torch.empty((1,), dtype=torch.int32, device=device)
        _l.check(_l.load().fnp_spconv_tiled_aborts_copy(_l.ptr(t), _l.stream()), "fnp_spconv_tiled_aborts_copy")
        return t

    def _counts_word(self, stage, ell_used, device, host=None):
        """enqueue ONE launch that collects what the host reads in a forward's one synchronisation — the five stage counts, the
        library's time-out counter, the pool counters of the compact rulebooks, in that order — into a fresh int32 tensor.
        host = (pinned int32 tensor, device sequence word): the launch also stores them into the pinned tensor and its own number
        behind them (fnp_gather_counts_host: the host polls for that number instead of waiting for an event)"""
        import ctypes
        from .. import lib as _l
        srcs = [s[2] for s in stage] + [None] + [u for u, _, _ in ell_used]
        arr = (ctypes.c_void_p * len(srcs))(*[None if t is None else t.data_ptr() for t in srcs])
        out = torch.empty((len(srcs),), dtype=torch.int32, device=device)
        reset = sum(1 << (len(stage) + 1 + i) for i in range(len(ell_used)))   # the pool counters are the engine's: zero for the next forward
        if host is not None:
            pin, seq = host
            assert pin.is_pinned() and pin.numel() > 16 and seq.is_cuda
            _l.check(_l.load().fnp_gather_counts_host(ctypes.cast(arr, ctypes.c_void_p), len(srcs), reset, _l.ptr(out), ctypes.c_void_p(pin.data_ptr()),
                                                      _l.ptr(seq), _l.stream()), "fnp_gather_counts_host")
        else:
            _l.check(_l.load().fnp_gather_counts(ctypes.cast(arr, ctypes.c_void_p), len(srcs), reset, _l.ptr(out), _l.stream()), "fnp_gather_counts")
        return out

    def _check_aborts(self, value):
        """raise when the counter has grown since this engine last looked: a tiled convolution ended a workgroup early and
        left rows of its output unwritten (never seen; the protocol's time-out exists so that a bug cannot hang the GPU)"""
        from .. import lib as _l
        global _ABORTS_SEEN
        seen, _ABORTS_SEEN = _ABORTS_SEEN, max(_ABORTS_SEEN, int(value))   # (the counter is the library's, not the engine's)
        if int(value) > seen:
            raise _l.FnpError(f"spconv_tile32_kernel: {int(value) - seen} hand-over wait(s) timed out during this forward; "
                              "its features are incomplete")

    TILE_ESC_MAX = float(os.environ.get("FNP_TILE_ESC_MAX", "0.004"))
    TILE_REPROBE = int(os.environ.get("FNP_TILE_REPROBE", "64"))
    TILE_REPROBE_MAX = int(os.environ.get("FNP_TILE_REPROBE_MAX", "4096"))

    def _gate_tick(self):
        """one frame has been issued — eagerly (_run_once), from a captured graph (run_points_graphed) or through a
        PointsPipeline slot: count the gather-kernel period of every gated stage down.  (only eager forwards used
        to count, so a graph user that met ONE dense frame stayed on the gather kernels for the rest of the process.)  When a
        period ends the stage is tried on tiles again — for a captured graph that is a recapture — and measured: a stream that
        stays dense doubles its period at every failed re-probe (64, 128, ... TILE_REPROBE_MAX frames), so the recaptures of a
        10-sweep stream die out; a stream that went back to single sweeps returns to tiles for good."""
        if self._gate_hold:
            return
        for li in list(self.tile_off):
            self.tile_off[li] = max(0, self.tile_off[li] - 1)

    def _heur_key(self):
        """what a captured graph bakes in besides capacities: which tiled stages run on the gather kernels"""
        return [li for li in sorted(self.tile_off) if self.tile_off[li] > 0]

    def _graph_key(self):
        return list(self.cap_factor) + list(self.ell_pool) + ["off"] + self._heur_key()

    def _ell_overflow(self, counts, ell_used, cap1):
        """pops the engine's extra counters off `counts`: the pool counters of the compact rulebooks — True when a pool was too
        small (its factor is grown: the caller reruns, as for a row capacity) — and the escape-group counters of the tiled stages
        (the tile gate above: never a rerun, the result is correct either way)"""
        over = False
        for used_t, pool, which in reversed(ell_used):
            used = counts.pop()
            if isinstance(which, tuple):          # ("esc", li): 32-row groups of stage li + 2 with an escape entry
                li = which[1]
                share = used / max(1.0, counts[li + 1] / 32.0)
                self.tile_escape_share[li] = share
                if share > self.TILE_ESC_MAX:
                    # first sighting: TILE_REPROBE frames on the gather kernels; a re-probe that fails again: twice the last period
                    period = min(self.tile_period[li] * 2, self.TILE_REPROBE_MAX) if li in self.tile_period else self.TILE_REPROBE
                    self.tile_period[li] = self.tile_off[li] = period
                else:
                    self.tile_period.pop(li, None)
                continue
            if used > pool:
                self.ell_pool[which] = max(self.ell_pool[which] * 2.0, used * 1.25 / max(cap1 * (self.cap_factor[0] if which else 1.0), 1))
                over = True
        return over

    # ---- weights --------------------------------------------------------------------------
    def _fold(self, conv, bn, dtype):
        """packed weight + BatchNorm(eval) as per-channel scale / shift.  The fold runs on the HOST in IEEE f32 —
        inv = 1 / sqrt(var + eps), scale = gamma * inv, shift = beta - mean * scale, each operation rounded once — i.e. the
        arithmetic of the CPU oracle's bn_fold, so that the fp32 engine equals the oracle bit for bit through the whole
        backbone (a device rsqrt differs from 1 / sqrt in the last place); it is cached with the packed weights."""
        import numpy as np
        w = conv.packed_weight(dtype)
        dev = bn.running_var.device
        f32 = lambda t: t.detach().float().cpu().numpy()
        var, mean, gamma, beta = f32(bn.running_var), f32(bn.running_mean), f32(bn.weight), f32(bn.bias)
        inv = np.float32(1.0) / np.sqrt(var + np.float32(bn.eps))     # (numpy f32: IEEE sqrt and divide, as the oracle's C)
        scale = gamma * inv
        shift = beta - mean * scale
        if conv.bias is not None:
            shift = shift + f32(conv.bias) * scale
        scale, shift = torch.from_numpy(np.ascontiguousarray(scale, np.float32)), torch.from_numpy(np.ascontiguousarray(shift, np.float32))
        return w, scale.contiguous().to(dev), shift.contiguous().to(dev)

    def prepare(self):
        m = self.m
        # (the key is taken on EVERY forward — weights may have been loaded or trained in between —, so it must be cheap: the walk over
        #  the module tree (m.parameters() + m.buffers(): ~0.1 ms of a 0.45 ms one-scene forward) is done once, what is read per call
        #  are the (container, name) slots it found — .to() / load_state_dict replace the tensors IN those containers.  The module's
        #  structure is fixed at construction.)
        slots = getattr(self, "_prep_slots", None)
        if slots is None:
            slots = self._prep_slots = ([(mod._parameters, k) for mod in m.modules() for k, v in mod._parameters.items() if v is not None] +
                                        [(mod._buffers, k) for mod in m.modules() for k, v in mod._buffers.items() if v is not None])
        key = tuple([(t.data_ptr(), t._version) for t in [d[k] for d, k in slots]]) + (self.act,)
        if self._prep_key == key:
            return self._prep
        P = {}
        fold = self._fold_x3 if self.x3 else (lambda c, b: self._fold(c, b, self.act))
        P['in'] = self._fold(m.conv_input[0], m.conv_input[1], torch.float32)  # f32 features in
        for name, seq, first in (('1', m.conv1, 0), ('2', m.conv2, 1), ('3', m.conv3, 1), ('4', m.conv4, 1)):
            if first:
                P['down' + name] = fold(seq[0][0], seq[0][1])
            blocks = []
            for blk in list(seq._modules.values())[first:]:
                blocks.append((fold(blk.conv1, blk.bn1), fold(blk.conv2, blk.bn2)))
            P['blocks' + name] = blocks
        P['out'] = fold(m.conv_out[0], m.conv_out[1])
        self._prep, self._prep_key = P, key
        return P

    def _fold_x3(self, conv, bn):
        """bf16x3: (W_hi, W_lo, shift) with BatchNorm(eval)'s scale folded into the weights before the split (the chained
        launches add their f32 partial results unscaled); W_hi + W_lo == scale * W to 2^-17; packed (K, Cout, Cin) bf16."""
        w32, scale, shift = self._fold(conv, bn, torch.float32)
        w32 = w32.as_subclass(torch.Tensor) if isinstance(w32, S.PermutedWeight) else w32
        wf = S.pack_weight(conv.weight, torch.float32).float() * scale.view(1, -1, 1)      # plain (K, Cout, Cin) layout
        hi = wf.to(torch.bfloat16)
        lo = (wf - hi.float()).to(torch.bfloat16)
        return (hi.contiguous(), lo.contiguous(), torch.ones_like(scale)), None, shift

    def _split(self, y, n):
        hi, lo = S.split_bf16(y, n)
        return (hi, lo, y)

    def _conv_x3(self, x, prm, rb, n, residual, out, ranked=False, want_f32=True):
        """one convolution of the bf16x3 engine (see __init__): x = (hi, lo, f32 rows); returns the same triple of the output
        (want_f32=False: the f32 rows are None — the first convolution of a residual block, of which only the split is read)"""
        (whi, wlo, ones), _, shift = prm
        res = None if residual is None else residual[2]
        C = int(whi.shape[1])
        fast = X3_FAST and ranked and (C in getattr(rb, "_tile_rb", {}) or (C == 128 and getattr(rb, "_sorted", None) is not None))
        if X3_FAST and X3_FUSED and (fast or (int(whi.shape[2]), C) in S.SPLIT_SHAPES):
            # the two cross terms are 2^-8 of the result: bf16 precision is enough for them, so they run with bf16 outputs (on the
            # bf16 engine's tile-rulebook / class-sorted kernels where the stage has them) chained through a bf16 residual; only the
            # main product keeps f32, and ITS epilogue adds them, applies the ReLU and writes the f32 rows with their (hi, lo) split
            # (conv_forward_split; FNP_X3_FUSED=0: f32-out gather kernel + a split pass)
            kw = dict(ranked=True) if fast else dict(tile=False)
            t = S.conv_forward(x[1], whi, rb, n, **kw)
            t = S.conv_forward(x[0], wlo, rb, n, residual=t, out=t, **kw)
            y, hi, lo = S.conv_forward_split(x[0], whi, rb, n, scale=ones, shift=shift, residual=res, addend=t, relu=True, ranked=ranked,
                                             tile=None if fast else False, want_f32=want_f32)
            return (hi, lo, y)
        if fast:
            t = S.conv_forward(x[1], whi, rb, n, ranked=True)
            t = S.conv_forward(x[0], wlo, rb, n, residual=t, ranked=True, out=t)
            y = S.conv_forward(x[0], whi, rb, n, out_dtype=torch.float32, scale=ones, shift=shift, residual=res, relu=False, tile=False)
            hi, lo = S.split_bf16_add(y, t, n, relu=True)
            return (hi, lo, y)
        t = S.conv_forward(x[1], whi, rb, n, out_dtype=torch.float32, residual=res, tile=False)
        t = S.conv_forward(x[0], wlo, rb, n, out_dtype=torch.float32, residual=t, out=t, tile=False)
        y = S.conv_forward(x[0], whi, rb, n, out_dtype=torch.float32, scale=ones, shift=shift, residual=t, relu=True, out=t if out is None else out,
                           tile=False)
        return self._split(y, n)

    # ---- persistent rank grids (zero between calls; cleared sparsely after use) -----------
    def _stage_shapes(self):
        m = self.m
        shapes = [list(m.sparse_shape)]
        for conv in (m.conv2[0][0], m.conv3[0][0], m.conv4[0][0], m.conv_out[0]):
            s = shapes[-1]
            shapes.append([(s[d] + 2 * conv.padding[d] - conv.kernel_size[d]) // conv.stride[d] + 1 for d in range(3)])
        return shapes

    def _get_grids(self, batch_size, device):
        key = (batch_size, str(device))
        g = self._grids.get(key)
        if g is None:
            g = [S.alloc_grid(batch_size, shp, device) for shp in self._stage_shapes()]
            self._grids[key] = g
        return g

    # ---- execution ------------------------------------------------------------------------
    def _ensure_clean(self):
        if self._dirty:
            for gs in self._grids.values():
                for g in gs:
                    g.zero_()
            for c in self._ell_ctr.values():
                c.zero_()
            self._dirty = False

    def _side_stream(self, device):
        """the engine's own stream for its index chain (_run_once)"""
        key = str(device)
        st = self._side.get(key)
        if st is None:
            # (eager two-stream forwards: a stream seen to run beside the caller's — sparse.concurrent_streams; inside a capture the
            #  stream only names a branch, and the test, which synchronises, cannot run)
            if torch.cuda.is_current_stream_capturing():
                st = torch.cuda.Stream(device)
            else:
                st = S.concurrent_streams(device, 1, beside=[torch.cuda.current_stream(device)])[0]
            self._side[key] = st
        return st

    def _ell_counter(self, which, device):
        """the engine's pool counter of compact rulebook `which`: zero between forwards (the counts launch resets what it reads)"""
        key = (str(device), which)
        c = self._ell_ctr.get(key)
        if c is None:
            c = self._ell_ctr[key] = torch.zeros((1,), dtype=torch.int32, device=device)
        return c

    def run_points(self, points, batch_offsets, batch_size, voxel_cfg, sync=True):
        if points.is_cuda and self._last_done is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(points.device).wait_event(self._last_done)   # (see _run_once: forwards return early)
        self._ensure_clean()
        self._dirty = True
        grids = self._get_grids(batch_size, points.device)
        vox = S.voxelize(points, batch_offsets, batch_size, voxel_cfg, grid=grids[0], workspace=self._vox_ws)
        self._vox_ws = vox['workspace']
        res = self.run(vox['mean'], vox['coords'], vox['n'], batch_size, grid1=grids[0], sync=sync, n_cells=vox['n_cells'])
        res['voxel_coords'], res['voxel_num_points'], res['voxel_features'] = vox['coords'], vox['num_points'], vox['mean']
        if sync:
            n1 = res['counts'][0]
            res['voxel_coords'] = vox['coords'][:n1]
            res['voxel_num_points'] = vox['num_points'][:n1]
            res['voxel_features'] = vox['mean'][:n1]
        return res

    def run_points_graphed(self, points, batch_offsets, batch_size, voxel_cfg, capacity=None, probe=False):
        # probe: the two-graph form of _PointsGraph; self.profile (a list) then receives the event pairs of the launches between
        assert (self.profile is None or probe) and self.rulebook_log is None, "measurement hooks are not capturable"
        n, C = points.shape
        capacity = int(capacity) if capacity else max(65536, (n + 65535) // 65536 * 65536)
        assert n <= capacity
        self.prepare()
        key = (batch_size, capacity, C, str(points.device), bool(probe))
        self._gate_tick()
        while True:
            g = self._graphs.get(key)
            if g is None or g.cap_factor != self._graph_key() or g.prep_key != self._prep_key:
                prof, self.profile = self.profile, None   # (the capture itself is not a measurement)
                try:
                    g = _PointsGraph(self, capacity, C, batch_size, voxel_cfg, points.device, probe=probe)
                finally:
                    self.profile = prof
                self._graphs[key] = g
            g.stage(points, batch_offsets)
            g.replay(self.profile if probe else None)
            stage, caps, shapes = g.res['stages'], g.res['caps'], g.res['shapes']
            counts = g.counts()   # the one host sync: the counts' copy between the two graphs
            overflow = self._ell_overflow(counts, g.res['ell_used'], caps[0])
            self._check_aborts(counts.pop())
            for l in range(1, 5):
                if counts[l] > caps[l]:
                    self.cap_factor[l - 1] = max(self.cap_factor[l - 1] * 2.0, counts[l] * 1.25 / caps[0])
                    overflow = True
            if not overflow:
                break
            torch.cuda.synchronize(points.device)                  # (the second graph of the failed forward may still be running)
            for gr in self._get_grids(batch_size, points.device):   # see _run_once: the sparse clear missed cells
                gr.zero_()
            del self._graphs[key]                                  # recapture with the larger buffers
        tensors = [spconv.SparseConvTensor(x[:counts[l]], idx[:counts[l]], shapes[l], batch_size, n_dev=nd)
                   for l, (x, idx, nd, _) in enumerate(stage)]
        n1 = counts[0]
        return {'x_conv1': tensors[0], 'x_conv2': tensors[1], 'x_conv3': tensors[2], 'x_conv4': tensors[3],
                'out': tensors[4], 'counts': counts, 'voxel_coords': g.vox['coords'][:n1],
                'voxel_num_points': g.vox['num_points'][:n1], 'voxel_features': g.vox['mean'][:n1]}

    def run(self, feats, indices, n1, batch_size, grid1=None, sync=True, n_cells=None, final_dtype=None):
        """feats (cap1,Cin) f32, indices (cap1,4) i32, n1 (1,) i32 device.
        n_cells: rows [n1, n_cells) of indices list further cells set in grid1 (voxels the voxeliser dropped).
        final_dtype: dtype conv_out writes (default: the engine's activation dtype)."""
        while True:
            res = self._run_once(feats, indices, n1, batch_size, grid1, sync, n_cells, final_dtype)
            if res is not None:
                return res
            # overflow: capacities were grown and grids cleared; rebuild grid1 from the indices
            grid1 = None

    def _run_once(self, feats, indices, n1, batch_size, grid1, sync, n_cells=None, final_dtype=None, probe=None):
        # The persistent rank grids must be all-zero on entry and are only cleared sparsely at the end of a run: if a
        # previous run died in between (FnpError, out of memory, KeyboardInterrupt), wipe them before they are reused
        # (run_points checks before it voxelises into grid1)
        if grid1 is None:
            self._ensure_clean()
        self._dirty = True
        # (a forward returns before the GPU has finished it — the counts leave early —: a caller that comes back on ANOTHER stream
        #  must not touch the persistent grids and workspaces before the previous forward is through)
        capturing = feats.is_cuda and torch.cuda.is_current_stream_capturing()
        # (bf16x3 parameters are ((W_hi, W_lo, ones), None, shift) triples and a layer is several launches: the per-launch
        #  measurement hooks and the probe's deferred launches are written for the one-launch engines)
        assert not (self.x3 and (self.profile is not None or self.rulebook_log is not None or (probe is not None and probe.probe))), \
            "bf16x3: profile / rulebook_log / probe graphs are not supported (time the whole forward instead: tools/bench_x3.py)"
        if feats.is_cuda and not capturing and self._last_done is not None:
            torch.cuda.current_stream(feats.device).wait_event(self._last_done)
        if not capturing:
            self._gate_tick()      # (a captured forward is counted where it is replayed: run_points_graphed, PointsPipeline)
        m, P, act = self.m, self.prepare(), self.act
        if final_dtype not in (None, act, torch.float32):
            final_dtype = None       # (the conv epilogue writes the activation dtype or f32; anything else is cast by the caller)
        dev = feats.device
        grids = self._get_grids(batch_size, dev)
        cap1 = max(indices.shape[0], 1)
        if grid1 is None:
            grid1 = S.build_grid(indices, n1, batch_size, m.sparse_shape, keep_order=True, grid=grids[0])
        caps = [cap1] + [max(256, int(cap1 * f)) for f in self.cap_factor]

        def conv(x, prm, rb, n, residual=None, out_dtype=act, ranked=False, want_f32=True):
            if self.x3 and isinstance(prm[0], tuple):
                return self._conv_x3(x, prm, rb, n, residual, None, ranked, want_f32)
            w, sc, sh = prm
            tag = (int(w.shape[2]), int(w.shape[1]), int(w.shape[0]), residual is not None, ranked)  # Cin, Cout, K, res
            if self.rulebook_log is not None:
                self.rulebook_log.append((tag, rb, n))
            if getattr(rb, "_ell", None) is not None and (rb.nbr is None or int(w.shape[2]) <= 8):   # on the compact rulebook
                if self.profile is None or (self.profile_only is not None and tag[:3] not in self.profile_only):
                    return S.conv_forward_ell(x, w, rb, n, out_dtype=out_dtype, scale=sc, shift=sh, residual=residual, relu=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y = S.conv_forward_ell(x, w, rb, n, out_dtype=out_dtype, scale=sc, shift=sh, residual=residual, relu=True)
                e1.record()
                self.profile.append((tag, e0, e1))
                return y
            if rb.nbr is None:   # fused strided layer
                if self.profile is None or (self.profile_only is not None and tag[:3] not in self.profile_only):
                    return S.conv_forward_strided(x, w, rb, scale=sc, shift=sh, relu=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y = S.conv_forward_strided(x, w, rb, scale=sc, shift=sh, relu=True)
                e1.record()
                self.profile.append((tag, e0, e1))
                return y
            if self.profile is None or (self.profile_only is not None and tag[:3] not in self.profile_only):
                return S.conv_forward(x, w, rb, n, out_dtype=out_dtype, scale=sc, shift=sh, residual=residual, relu=True,
                                      ranked=ranked)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = S.conv_forward(x, w, rb, n, out_dtype=out_dtype, scale=sc, shift=sh, residual=residual, relu=True,
                               ranked=ranked)
            e1.record()
            self.profile.append((tag, e0, e1))
            return y

        def blocks(x, rb, n, prms, ranked=False):
            for p1, p2 in prms:
                t = conv(x, p1, rb, n, ranked=ranked, want_f32=False)     # (bf16x3: only the split of t is read)
                x = conv(t, p2, rb, n, residual=x, ranked=ranked)
            return x

        # stage 1 (conv_input + conv1): one SubM rulebook serves indice_keys 'subm1' and 'res1'.  16-bit engines: the rulebook
        # kernel writes the compact form only (32 bytes per row, the neighbours that exist; no (27, cap) table): conv_input (5
        # input channels) sums the records on the VALU (2.3x the table kernel), the four 16 -> 16 layers and the strided
        # 16 -> 32 layer run the matrix kernel with the records of a tile's rows expanded into LDS at the top of the tile
        # (fnp_spconv_forward_ell_mfma: the table kernel's values bit for bit, +3 % end to end: the table was 108 of the ~170
        # bytes such a row moves).  FNP_ELL_MFMA=0: the round's first form — table beside the records, matrix kernels on the
        # table; FNP_ELL=1 with it: 16-channel rows on the VALU kernel.  f32, and while rulebooks are logged: tables only.
        ell = (act in (torch.bfloat16, torch.float16) and self.rulebook_log is None and S.ELL_MODE is not False
               and (int(P['in'][0].shape[2]), 16) in S.ELL_SHAPES and not isinstance(P['in'][0], S.PermutedWeight))
        ell_used = []
        ell_all = ell and (S.ELL_MODE is True or (S.ELL_MODE is None and S.ELL_MFMA))

        # TWO STREAMS (round 3).  Everything that builds indices — rank grids, output coordinates, rulebooks, records, the class
        # sort — depends on coordinates only, never on features: the whole chain of the five stages is enqueued first, on the
        # engine's side stream, one event per stage; the convolutions follow on the caller's stream and wait for their stage's
        # event.  The index kernels (bound by atomics and latency, 1.2 ms of a 5.4 ms step at 64 scenes) then run beside the
        # convolutions of the stage before them instead of between them.  Same kernels, same values.  Default: inside a hipGraph
        # capture only (see TWO_STREAMS above); never while a profile or a rulebook log is being taken.
        two = (self.profile is None and self.rulebook_log is None and feats.is_cuda and self.two_streams
               and (TWO_STREAMS if TWO_STREAMS is not None else torch.cuda.is_current_stream_capturing()))
        main = torch.cuda.current_stream(dev) if feats.is_cuda else None
        side = self._side_stream(dev) if two else None
        events = []

        class _Index:   # index work goes to the side stream; leaving the block records the stage's event
            def __enter__(ctx):
                if two:
                    ctx.cm = torch.cuda.stream(side)
                    ctx.cm.__enter__()
                return ctx

            def __exit__(ctx, *exc):
                if two:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    events.append(ev)
                    ctx.cm.__exit__(*exc)
                return False

        if two:
            side.wait_stream(main)   # (inputs, and the sparse clear of the previous forward, are the caller's stream's)

        # ---- index chain ------------------------------------------------------------------------------------------------
        with _Index():
            if ell:
                rb1 = S.rulebook_subm_ell(indices, n1, grid1, int(cap1 * self.ell_pool[0]) + 64, with_table=not ell_all, used=self._ell_counter(0, dev))
                ell_used.append((rb1._ell[2], rb1._ell[1], 0))
            else:
                # (f32: the 16 -> 16 layers sweep their ranges class by class, like every f32 SubM stage below)
                srt1 = S.f32_sorted_by_default(16, act, cap1) and self.rulebook_log is None and not self.x3
                rb1 = S.rulebook_subm(indices, n1, grid1, 3, masks=srt1)
                if srt1:
                    S.classsort_f32(rb1, n1, 16)
        down_convs = (m.conv2[0][0], m.conv3[0][0], m.conv4[0][0], m.conv_out[0])
        premarked = False     # did the previous stage's rulebook kernel mark this strided layer's output sites already?
        idx_prev, n_prev, g_prev = indices, n1, grid1
        books = []
        for li, (down_key, blk_key, dconv) in enumerate((('down2', 'blocks2', m.conv2[0][0]),
                                                          ('down3', 'blocks3', m.conv3[0][0]),
                                                          ('down4', 'blocks4', m.conv4[0][0]))):
            # the down-sampling layers resolve their neighbours inside the convolution (their rulebook has no other
            # user): no (27, cap) table; the table path stays for f32 and when the rulebooks are being logged
            wd = P[down_key][0]
            fused = (act in (torch.bfloat16, torch.float16) and self.rulebook_log is None and tuple(dconv.kernel_size) == (3, 3, 3)
                     and (int(wd.shape[2]), int(wd.shape[1])) in S.FUSED_STRIDED_SHAPES)
            ell_down = ell_all and li == 0 and tuple(dconv.kernel_size) == (3, 3, 3) and (int(wd.shape[2]), int(wd.shape[1])) in S.ELL_SHAPES
            with _Index():
                rbs = S.rulebook_strided(idx_prev, n_prev, g_prev, dconv.kernel_size, dconv.stride, dconv.padding,
                                         caps[li + 1], out_grid=grids[li + 1], want_nbr=not (fused or ell_down), premarked=premarked)
                if ell_down:
                    S.ell_for_strided(rbs, int(caps[li + 1] * self.ell_pool[1]) + 64, used=self._ell_counter(1, dev))
                    ell_used.append((rbs._ell[2], rbs._ell[1], 1))
                # stages 2-4: rows are in rank-grid order on both sides of the SubM convolutions; where they run on the tile
                # rulebook (stage 2, 32 channels), the rulebook kernel writes it in the same pass
                w0 = P[blk_key][0][0][0]
                ch = int((w0[0] if isinstance(w0, tuple) else w0).shape[1])
                act_k = torch.bfloat16 if (self.x3 and X3_FAST) else act     # (bf16x3: its cross terms run on the bf16 engine's kernels)
                srt = S.sorted_by_default(ch, ch, act_k, caps[li + 1])
                tiled = (S.tiled_by_default(ch, act_k, caps[li + 1]) and S.tiled_fits(caps[li + 1], ch, caps[li + 1], caps[li + 1])
                         and self.tile_off.get(li, 0) <= 0)
                # the SubM rulebook kernel of this stage also marks the output sites of the NEXT strided layer (the coordinates are
                # in its registers): that layer's own marking launch goes
                nxt = down_convs[li + 1]
                lean = tiled and self.rulebook_log is None and (not self.x3 or (X3_FAST and X3_FUSED))   # (all four layers of the stage run tiled)
                mark_next = (grids[li + 2], nxt.kernel_size, nxt.stride, nxt.padding) if (lean or srt) and S.MARK_FUSED else None
                srt32 = S.f32_sorted_by_default(ch, act, caps[li + 1]) and self.rulebook_log is None and not self.x3
                esc_ctr = self._ell_counter(("esc", li), dev) if (lean and S.TILE_MODE is None) else None
                rb = S.rulebook_subm(rbs.out_indices, rbs.out_n, rbs.out_grid, 3, tile_channels=ch if tiled else None, masks=srt or srt32,
                                     lean_table=lean, mark_next=mark_next, esc_counter=esc_ctr)
                if esc_ctr is not None:
                    ell_used.append((esc_ctr, None, ("esc", li)))
                if self.tile_off.get(li, 0) > 0:
                    rb._no_tile = True
                premarked = bool(getattr(rb, "_marked_next", False))
                if srt:
                    S.classsort(rb, rbs.out_n, ch)   # stage 4: the 128-channel layers sweep their rows class by class
                if srt32:
                    S.classsort_f32(rb, rbs.out_n, ch)   # f32 engine: every stage's ranges in class order
            books.append((down_key, blk_key, rbs, rb))
            idx_prev, n_prev, g_prev = rbs.out_indices, rbs.out_n, rbs.out_grid
        oconv = m.conv_out[0]
        # HOST COUNTS (_PointsGraph.host_counts): the counts launch stores into pinned host memory itself, so it needs no cut and no
        # event — and it can sit where the STAGE counts are final, at the end of the index chain (on the index branch), instead of behind the
        # strided layer of stage 4: the host has its counts after ~half of a one-scene forward and sizes the outputs, returns, and
        # issues the next frame while the convolutions still run.  (What is not final there is the tiled kernels' time-out counter: in
        # this form a time-out — never seen; the protocol's guard against a hang — raises with the NEXT forward's counts.)
        early_counts = probe is not None and getattr(probe, "host_counts", False)
        counts_box = {}
        with _Index():
            rbo = S.rulebook_strided(idx_prev, n_prev, g_prev, oconv.kernel_size, oconv.stride, oconv.padding, caps[4],
                                     out_grid=grids[4], premarked=premarked)
            if early_counts:
                srcs = [(None, None, n1)] + [(None, None, b[2].out_n) for b in books] + [(None, None, rbo.out_n)]
                counts_box['dev'] = self._counts_word(srcs, ell_used, dev, host=probe.counts_host())
                probe.counts_ready(counts_box['dev'])

        # ---- convolutions -----------------------------------------------------------------------------------------------
        if probe is not None and getattr(probe, "split_index", False):
            assert not two, "a split capture keeps its index chain on the capture stream"
            probe.index_ready()      # (_PointsGraph: the voxeliser + index chain are a graph of their own)
        joined, cleared = [False], [None]

        def clear_jobs():
            jobs = [(grid1, indices, n_cells if n_cells is not None else n1)]
            jobs += [(b[2].out_grid, b[2].out_indices, b[2].out_n) for b in books]
            return jobs + [(rbo.out_grid, rbo.out_indices, rbo.out_n)]

        def clear_beside():
            # the last reader of a rank grid has been issued (the strided layers resolve their neighbours in the input grid; the
            # layers behind the strided layer of stage 4 run on tables): the sparse clear of all five grids goes to a branch of
            # its own and runs beside the convolutions that follow instead of behind conv_out, at the end of the chain
            br = side
            ev_c = torch.cuda.Event()
            ev_c.record(main)
            br.wait_event(ev_c)
            with torch.cuda.stream(br):
                S.clear_grids(clear_jobs())
                cleared[0] = torch.cuda.Event()
                cleared[0].record(br)

        def ready(i):
            if two and not joined[0]:
                main.wait_event(events[i])

        def blocks_deferred(x, rb, n, prms):
            # (probe: _PointsGraph) the stage's output buffers are allocated in the first graph's pool, the capture is cut, and
            # the launches are handed over as closures: replay() issues them between the two graphs
            plan = []
            for p1, p2 in prms:
                t = torch.empty((rb.cap_out, int(p1[0].shape[1])), dtype=act, device=dev)
                y = torch.empty((rb.cap_out, int(p2[0].shape[1])), dtype=act, device=dev)
                plan.append((x, t, y, p1, p2))
                x = y
            counts_now()     # (probe: the capture is cut inside)
            for xin, t, y, p1, p2 in plan:
                tag1 = (int(p1[0].shape[2]), int(p1[0].shape[1]), int(p1[0].shape[0]), False, True)
                tag2 = (int(p2[0].shape[2]), int(p2[0].shape[1]), int(p2[0].shape[0]), True, True)
                probe.deferred.append((tag1, lambda xin=xin, t=t, p1=p1: S.conv_forward(xin, p1[0], rb, n, out_dtype=act, scale=p1[1], shift=p1[2],
                                                                                     relu=True, ranked=True, out=t)))
                probe.deferred.append((tag2, lambda xin=xin, t=t, y=y, p2=p2: S.conv_forward(t, p2[0], rb, n, out_dtype=act, scale=p2[1], shift=p2[2],
                                                                                          residual=xin, relu=True, ranked=True, out=y)))
            return x

        # THE COUNTS LEAVE EARLY (round 3, late).  Everything the host reads in a forward's one synchronisation — the five stage
        # counts, the tiled kernels' time-out counter (their last launch is in stage 3), the pool counters of the compact
        # rulebooks — is final once the index chain is done and the strided layer of stage 4 has been issued: the counts launch
        # and (stream path) their copy to pinned memory go in front of the four 128 -> 128 convolutions, 30 % of the step.  The
        # host then waits for THAT event, sizes the outputs and returns while the GPU is still convolving; the caller's next
        # forward queues up behind it, and the ~0.1 ms the GPU used to idle between two forwards (copy back, Python, the first
        # launches of the next forward) is gone.  The returned tensors are ordinary stream-ordered torch tensors.
        def counts_now():
            if early_counts:
                return
            srcs = [(None, None, n1)] + [(None, None, b[2].out_n) for b in books] + [(None, None, rbo.out_n)]
            cd = counts_box['dev'] = self._counts_word(srcs, ell_used, dev, host=probe.counts_host() if probe is not None else None)
            if probe is not None:
                probe.counts_ready(cd)          # (_PointsGraph: the first captured graph ends here)
            elif sync:
                pin = self._counts_pin
                if pin is None or pin.numel() < cd.numel():
                    pin = self._counts_pin = torch.empty((16,), dtype=torch.int32, pin_memory=True)
                pin[:cd.numel()].copy_(cd, non_blocking=True)
                counts_box['ev'] = torch.cuda.Event()
                counts_box['ev'].record()

        ready(0)
        x = conv(feats, P['in'], rb1, n1)
        if self.x3:
            x = self._split(x, n1)      # (conv_input ran on the f32 kernel: exact; from here on (hi, lo, f32) triples)
        f32_of = (lambda t: t[2]) if self.x3 else (lambda t: t)
        x1 = blocks(x, rb1, n1, P['blocks1'])
        stage = [(f32_of(x1), indices, n1, grid1)]
        x_prev = x1
        for li, (down_key, blk_key, rbs, rb) in enumerate(books):
            ready(li + 1)
            x = conv(x_prev, P[down_key], rbs, rbs.out_n)
            if li == len(books) - 1:
                ready(4)            # (every index kernel is behind us; a capture may end only with every forked stream rejoined)
                joined[0] = True
                if two and (probe is None or early_counts):   # (a two-graph capture is cut right here: no open branch across the cut)
                    clear_beside()
                if probe is not None and probe.probe:
                    x = blocks_deferred(x, rb, rbs.out_n, P[blk_key])
                else:
                    counts_now()
                    x = blocks(x, rb, rbs.out_n, P[blk_key], ranked=True)
            else:
                x = blocks(x, rb, rbs.out_n, P[blk_key], ranked=True)
            stage.append((f32_of(x), rbs.out_indices, rbs.out_n, rbs.out_grid))
            x_prev = x
        # (one-stream captures — the slots of a PointsPipeline — stay LINEAR: with the clear on a branch beside conv_out the runtime's
        #  fork / join serialised a batch's index graph behind the other batch's convolutions, 13.2 k -> 12.2 k scenes/s at 128 scenes)
        xo = conv(x_prev, P['out'], rbo, rbo.out_n, out_dtype=final_dtype or act)
        stage.append((f32_of(xo), rbo.out_indices, rbo.out_n, rbo.out_grid))

        # leave every persistent grid zeroed for the next call (O(rows) sparse clear, all five grids in one launch)
        if cleared[0] is not None:
            main.wait_event(cleared[0])
        else:
            S.clear_grids(clear_jobs())
        self._dirty = False
        counts_dev = counts_box['dev']
        if feats.is_cuda and not capturing:
            self._last_done = torch.cuda.Event()
            self._last_done.record()

        shapes = self._stage_shapes()
        if not sync:
            return {'stages': stage, 'shapes': shapes, 'caps': caps, 'batch_size': batch_size, 'counts_dev': counts_dev, 'ell_used': ell_used}
        counts_box['ev'].synchronize()   # the one host sync: the counts' copy, not the end of the forward
        counts = self._counts_pin[:counts_dev.numel()].tolist()
        overflow = self._ell_overflow(counts, ell_used, cap1)
        self._check_aborts(counts.pop())
        for l in range(1, 5):
            if counts[l] > caps[l]:
                self.cap_factor[l - 1] = max(self.cap_factor[l - 1] * 2.0, counts[l] * 1.25 / cap1)
                overflow = True
        if overflow:
            # rows beyond a capacity were never emitted, so the sparse clear missed their cells:
            # wipe the persistent grids before the retry with larger buffers
            for g in grids:
                g.zero_()
            self._dirty = False
            if feats.is_cuda and not capturing:
                self._last_done = torch.cuda.Event()     # (a caller on another stream must come behind the wipe too)
                self._last_done.record()
            return None
        tensors = []
        for l, (x, idx, n, g) in enumerate(stage):
            c = counts[l]
            tensors.append(spconv.SparseConvTensor(x[:c], idx[:c], shapes[l], batch_size, n_dev=n))
        return {'x_conv1': tensors[0], 'x_conv2': tensors[1], 'x_conv3': tensors[2], 'x_conv4': tensors[3],
                'out': tensors[4], 'counts': counts}


def _with_perm(grid, cap, device):
    if grid.perm is None or grid.perm.numel() < cap:
        grid.perm = torch.empty((cap,), dtype=torch.int32, device=device)
    return grid
