"""numpy front-end of oracle/liboracle.so (oracle/fnp_oracle.c) plus the layer graph of
VoxelResBackBone8x restated on top of it.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by findnpropagate_amd/.  Each function cites the reference file:line
(relative to the reference tree) it follows.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int, c_int64, c_size_t, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    """Compile the C restatement (and, when /root/reference exists, oracle/_ref)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_box_overlap.restype = c_float
        _lib.orc_iou_bev.restype = c_float
        _lib.orc_iou_normal.restype = c_float
        _lib.orc_nms.restype = c_int
        _lib.orc_voxelize.restype = c_int
        _lib.orc_rulebook_strided.restype = c_int
        _lib.orc_max_threads.restype = c_int
        _lib.orc_set_threads(1)      # deterministic default; bench.py's all-core leg raises it
    return _lib


def set_threads(n):
    """Threads of the OpenMP loop in orc_spconv_apply (results do not depend on it)."""
    lib().orc_set_threads(int(n))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(c_void_p)


# ---------------------------------------------------------------- roiaware_pool3d
def points_in_boxes(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,M) int32: roiaware_pool3d_kernel.cu:313-336."""
    points, boxes = _f32(points), _f32(boxes)
    B, M, _ = points.shape
    T = boxes.shape[1]
    out = np.empty((B, M), np.int32)
    lib().orc_points_in_boxes(_p(boxes), _p(points), _p(out), B, T, M)
    return out


def points_in_boxes_count(points, boxes):
    """points (M,3), boxes (T,7) -> (T,) counts: frustum_proposals_v1.py:930-932."""
    points, boxes = _f32(points), _f32(boxes)
    out = np.empty((boxes.shape[0],), np.int32)
    lib().orc_points_in_boxes_count(_p(boxes), _p(points), _p(out), boxes.shape[0], points.shape[0])
    return out


def points_in_boxes_dense(points, boxes):
    """points (M,3), boxes (T,7) -> (T,M) 0/1: roiaware_pool3d.cpp:143-168 (MARGIN 1e-2)."""
    points, boxes = _f32(points), _f32(boxes)
    out = np.empty((boxes.shape[0], points.shape[0]), np.int32)
    lib().orc_points_in_boxes_dense(_p(boxes), _p(points), _p(out), boxes.shape[0], points.shape[0])
    return out


# ---------------------------------------------------------------- iou3d_nms
def _pairwise(fn, a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    fn(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
    return out


def boxes_overlap_bev(a, b):
    return _pairwise(lib().orc_boxes_overlap_bev, a, b)


def boxes_iou_bev(a, b):
    return _pairwise(lib().orc_boxes_iou_bev, a, b)


def boxes_iou3d(a, b):
    return _pairwise(lib().orc_boxes_iou3d, a, b)


def boxes_aligned_overlap_bev(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0],), np.float32)
    lib().orc_boxes_aligned_overlap_bev(_p(a), _p(b), a.shape[0], _p(out))
    return out


def iou_normal(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_iou_normal(_p(a), _p(b)))


def nms(boxes_sorted, thresh, rotated):
    """boxes pre-sorted by score desc -> kept indices (iou3d_nms.cpp:113-209)."""
    boxes_sorted = _f32(boxes_sorted)
    keep = np.empty((max(boxes_sorted.shape[0], 1),), np.int64)
    n = lib().orc_nms(_p(boxes_sorted), boxes_sorted.shape[0], c_float(thresh), int(bool(rotated)), _p(keep))
    return keep[:n].copy()


def nms_gpu(boxes, scores, thresh, rotated=True):
    """iou3d_nms_utils.nms_gpu / nms_normal_gpu (iou3d_nms_utils.py:120-152): stable argsort here;
    the reference uses torch's (unstable) sort, ties are the caller's concern."""
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    keep = nms(np.asarray(boxes, np.float32)[order], thresh, rotated)
    return order[keep]


# ---------------------------------------------------------------- voxeliser + VFE
def voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """Point2VoxelCPU3d.point_to_voxel (data_processor.py:38-61): returns
    voxels (M,max_points,C), coords (M,3)[z,y,x], num_points (M,)."""
    points = _f32(points)
    n, C = points.shape
    vs = _f32(voxel_size)
    rng = _f32(coors_range)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int32)  # data_processor.py:257-258
    voxels = np.zeros((max_voxels, max_points, C), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    m = lib().orc_voxelize(_p(points), n, C, _p(rng[:3].copy()), _p(vs), _p(grid), max_points, max_voxels,
                           _p(voxels), _p(coords), _p(num))
    return voxels[:m].copy(), coords[:m].copy(), num[:m].copy()


def mean_vfe(voxels, num_points):
    """MeanVFE.forward (mean_vfe.py:25-29)."""
    voxels = _f32(voxels)
    M, P, C = voxels.shape
    out = np.empty((M, C), np.float32)
    lib().orc_mean_vfe(_p(voxels), _p(_i32(num_points)), M, P, C, _p(out))
    return out


# ---------------------------------------------------------------- sparse conv
class SparseTensor:
    """features (N,C) f32, indices (N,4) [b,z,y,x] int32, spatial_shape [D,H,W]."""

    def __init__(self, features, indices, spatial_shape, batch_size):
        self.features = _f32(features)
        self.indices = _i32(indices)
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.rulebooks = {}

    def dense(self):
        """SparseConvTensor.dense() (height_compression.py:21): (B,C,D,H,W)."""
        C = self.features.shape[1]
        out = np.zeros((self.batch_size, C, *self.spatial_shape), np.float32)
        shape = _i32(self.spatial_shape)
        lib().orc_sparse_to_dense(_p(self.features), _p(self.indices), self.features.shape[0], C, _p(shape), _p(out))
        return out


def _triple(v):
    return [int(v)] * 3 if np.isscalar(v) else [int(x) for x in v]


def rulebook_subm(indices, spatial_shape, ksize):
    indices = _i32(indices)
    n = indices.shape[0]
    ksize = _i32(_triple(ksize))
    K = int(np.prod(ksize))
    pin = np.empty((K, max(n, 1)), np.int32)
    pout = np.empty((K, max(n, 1)), np.int32)
    pn = np.zeros((K,), np.int32)
    lib().orc_rulebook_subm(_p(indices), n, _p(_i32(spatial_shape)), _p(ksize), _p(pin), _p(pout), _p(pn))
    return pin, pout, pn


def rulebook_strided(indices, spatial_shape, ksize, stride, padding):
    indices = _i32(indices)
    n = indices.shape[0]
    ksize, stride, padding = _i32(_triple(ksize)), _i32(_triple(stride)), _i32(_triple(padding))
    shape = _i32(spatial_shape)
    out_shape = ((shape + 2 * padding - ksize) // stride + 1).astype(np.int32)
    K = int(np.prod(ksize))
    cap = max(n * K, 1)
    out_idx = np.empty((cap, 4), np.int32)
    pin = np.empty((K, max(n, 1)), np.int32)
    pout = np.empty((K, max(n, 1)), np.int32)
    pn = np.zeros((K,), np.int32)
    m = lib().orc_rulebook_strided(_p(indices), n, _p(shape), _p(ksize), _p(stride), _p(padding), _p(out_shape),
                                   _p(out_idx), cap, _p(pin), _p(pout), _p(pn))
    return out_idx[:m].copy(), [int(s) for s in out_shape], pin, pout, pn


def conv_apply(features, weight, pin, pout, pn, n_out):
    """weight (Cout, kD, kH, kW, Cin) — spconv 2.x layout (detector3d_template.py:401-433)."""
    features = _f32(features)
    weight = _f32(weight)
    Cout, Cin = weight.shape[0], weight.shape[-1]
    K = int(np.prod(weight.shape[1:4]))
    w = weight.reshape(Cout, K, Cin)
    out = np.zeros((n_out, Cout), np.float32)
    lib().orc_spconv_apply(_p(features), _p(_f32(w)), _p(pin), _p(pout), _p(pn), pin.shape[1], K, Cin, Cout, _p(out))
    return out


def conv_backward(features, weight, pin, pout, pn, grad_out):
    """Backward of conv_apply (spconv's autograd), f64 accumulation:
    dx[i] += W_k dy[o],  dW[co, k, ci] += dy[o, co] x[i, ci]  over the rulebook pairs (k, i, o).
    Returns (dx (n_in, Cin), dW in the weight's (Cout, kD, kH, kW, Cin) layout), both f32."""
    features, weight, grad_out = _f32(features), _f32(weight), _f32(grad_out)
    Cout, Cin = weight.shape[0], weight.shape[-1]
    K = int(np.prod(weight.shape[1:4]))
    w = weight.reshape(Cout, K, Cin).astype(np.float64)
    dx = np.zeros((features.shape[0], Cin), np.float64)
    dw = np.zeros((Cout, K, Cin), np.float64)
    for k in range(K):
        m = int(pn[k])
        if m == 0:
            continue
        i, o = pin[k, :m], pout[k, :m]
        dy = grad_out[o].astype(np.float64)
        np.add.at(dx, i, dy @ w[:, k, :])                       # (m, Cout) @ (Cout, Cin)
        dw[:, k, :] += dy.T @ features[i].astype(np.float64)     # (Cout, m) @ (m, Cin)
    return dx.astype(np.float32), dw.reshape(weight.shape).astype(np.float32)


def subm_conv(x, weight, indice_key=None):
    """spconv.SubMConv3d forward (spconv_backbone.py:39-46): rulebook cached per indice_key."""
    ksize = weight.shape[1:4]
    key = ("subm", indice_key)
    if indice_key is None or key not in x.rulebooks:
        rb = rulebook_subm(x.indices, x.spatial_shape, ksize)
        if indice_key is not None:
            x.rulebooks[key] = rb
    else:
        rb = x.rulebooks[key]
    out = SparseTensor(conv_apply(x.features, weight, *rb, x.features.shape[0]), x.indices, x.spatial_shape, x.batch_size)
    out.rulebooks = x.rulebooks
    return out


def sparse_conv(x, weight, stride, padding):
    """spconv.SparseConv3d forward (spconv_backbone.py:14-15)."""
    ksize = weight.shape[1:4]
    out_idx, out_shape, pin, pout, pn = rulebook_strided(x.indices, x.spatial_shape, ksize, stride, padding)
    return SparseTensor(conv_apply(x.features, weight, pin, pout, pn, out_idx.shape[0]), out_idx, out_shape, x.batch_size)


def inverse_conv(x, weight, paired):
    """spconv.SparseInverseConv3d forward (post_act_block 'inverseconv', spconv_backbone.py:16-17; spconv_unet.py): the indice
    pairs of the SparseConv3d with the same indice_key, the two sides swapped — out[i] += in[o] W_k for every pair (k, i, o) of
    `paired` = (in_indices, in_shape, pin, pout, pn) of that layer; weight (Cout, kD, kH, kW, Cin).  Output sites = the paired
    layer's input sites.  numpy, f64 accumulation (small cases)."""
    in_idx, in_shape, pin, pout, pn = paired
    weight = _f32(weight)
    Cout, Cin = weight.shape[0], weight.shape[-1]
    K = int(np.prod(weight.shape[1:4]))
    w = weight.reshape(Cout, K, Cin).astype(np.float64)
    out = np.zeros((in_idx.shape[0], Cout), np.float64)
    f = _f32(x.features).astype(np.float64)
    for k in range(K):
        m = int(pn[k])
        if m:
            np.add.at(out, pin[k, :m], f[pout[k, :m]] @ w[:, k, :].T)
    return SparseTensor(out.astype(np.float32), in_idx, in_shape, x.batch_size)


def bn_fold(bn, eps=1e-3):
    """bn = dict(weight, bias, running_mean, running_var) -> (scale, shift) f32."""
    C = bn["weight"].shape[0]
    scale = np.empty((C,), np.float32)
    shift = np.empty((C,), np.float32)
    lib().orc_bn_fold(_p(_f32(bn["weight"])), _p(_f32(bn["bias"])), _p(_f32(bn["running_mean"])),
                      _p(_f32(bn["running_var"])), c_float(eps), C, _p(scale), _p(shift))
    return scale, shift


def scale_shift_act(feats, scale, shift, residual=None, relu=True, bf16=False):
    feats = np.array(feats, dtype=np.float32, order="C", copy=True)
    n, C = feats.shape
    lib().orc_scale_shift_act(_p(feats), n, C, _p(scale) if scale is not None else None,
                              _p(shift) if shift is not None else None,
                              _p(_f32(residual)) if residual is not None else None, int(relu))
    if bf16:
        round_bf16(feats)
    return feats


def round_bf16(a):
    """In-place round-to-nearest-even to bf16 precision (emulates bf16 feature storage)."""
    assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().orc_round_bf16(_p(a), c_size_t(a.size))
    return a


# ---------------------------------------------------------------- VoxelResBackBone8x
# The layer graph as data (spconv_backbone.py:193-234); backbone_forward executes this table and
# tests/test_backbone_tree.py holds it to the module tree the reference's own constructor builds
# (tests/golden/backbone_tree.json).  ("subm" | "down", prefix, cout, kernel, stride, padding, indice_key)
# is a convolution + BatchNorm + ReLU (post_act_block, :8-27 / conv_input, conv_out); ("block", prefix,
# planes, indice_key) a SparseBasicBlock (:30-67): conv1-bn1-relu-conv2-bn2-(+identity)-relu.
RES_BACKBONE8X = (
    ("subm", "conv_input", 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm1", None),          # :193-197
    ("block", "conv1.0", 16, "res1", None),                                             # :200-203
    ("block", "conv1.1", 16, "res1", "x_conv1"),
    ("down", "conv2.0", 32, (3, 3, 3), (2, 2, 2), (1, 1, 1), "spconv2", None),           # :205-210
    ("block", "conv2.1", 32, "res2", None),
    ("block", "conv2.2", 32, "res2", "x_conv2"),
    ("down", "conv3.0", 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), "spconv3", None),           # :212-217
    ("block", "conv3.1", 64, "res3", None),
    ("block", "conv3.2", 64, "res3", "x_conv3"),
    ("down", "conv4.0", 128, (3, 3, 3), (2, 2, 2), (0, 1, 1), "spconv4", None),          # :219-224
    ("block", "conv4.1", 128, "res4", None),
    ("block", "conv4.2", 128, "res4", "x_conv4"),
    ("down", "conv_out", 128, (3, 1, 1), (2, 1, 1), "last_pad", "spconv_down2", "out"),  # :228-234
)


# VoxelBackBone8x (spconv_backbone.py:70-181; the plain variant: post_act_block rows only, 64 channels in stage 4)
PLAIN_BACKBONE8X = (
    ("subm", "conv_input", 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm1", None),          # :90-94
    ("subm", "conv1.0", 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm1", "x_conv1"),         # :97-99
    ("down", "conv2.0", 32, (3, 3, 3), (2, 2, 2), (1, 1, 1), "spconv2", None),           # :101-106
    ("subm", "conv2.1", 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm2", None),
    ("subm", "conv2.2", 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm2", "x_conv2"),
    ("down", "conv3.0", 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), "spconv3", None),           # :108-113
    ("subm", "conv3.1", 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm3", None),
    ("subm", "conv3.2", 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm3", "x_conv3"),
    ("down", "conv4.0", 64, (3, 3, 3), (2, 2, 2), (0, 1, 1), "spconv4", None),           # :115-120
    ("subm", "conv4.1", 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm4", None),
    ("subm", "conv4.2", 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm4", "x_conv4"),
    ("down", "conv_out", 128, (3, 1, 1), (2, 1, 1), "last_pad", "spconv_down2", "out"),  # :126-132
)


def backbone_layers(input_channels=5, last_pad=0, table=None):
    """RES_BACKBONE8X (or `table`) flattened to its convolutions in execution order: dicts with the conv's
    state_dict prefix, its BatchNorm's prefix, subm, in/out channels, kernel, stride, padding,
    indice_key, whether a residual is added before the ReLU, and the output name it closes."""
    out, cin = [], int(input_channels)
    for e in (RES_BACKBONE8X if table is None else table):
        if e[0] == "block":
            _, prefix, planes, key, name = e
            for j, (c, b) in enumerate((("conv1", "bn1"), ("conv2", "bn2"))):
                out.append(dict(conv=f"{prefix}.{c}", bn=f"{prefix}.{b}", subm=True, cin=cin, cout=planes,
                                kernel=[3, 3, 3], stride=[1, 1, 1], padding=[1, 1, 1], indice_key=key,
                                residual=(j == 1), output=name if j == 1 else None))
                cin = planes
        else:
            kind, prefix, cout, k, s, p, key, name = e
            p = _triple(last_pad) if isinstance(p, str) else list(p)
            out.append(dict(conv=f"{prefix}.0", bn=f"{prefix}.1", subm=(kind == "subm"), cin=cin, cout=cout,
                            kernel=list(k), stride=list(s), padding=p, indice_key=key, residual=False, output=name))
            cin = cout
    return out


def backbone_forward(params, voxel_features, voxel_coords, batch_size, sparse_shape, bf16=False, last_pad=0, table=None):
    """VoxelResBackBone8x.forward (spconv_backbone.py:243-295) in eval mode (table=PLAIN_BACKBONE8X: VoxelBackBone8x.forward,
    :134-181).

    params: dict name -> ndarray with the reference state_dict keys
    ('conv_input.0.weight', 'conv_input.1.weight', ..., 'conv1.0.conv1.weight', ...), conv
    weights in spconv 2.x layout (Cout,kD,kH,kW,Cin).  bf16=True emulates the MFMA path's bf16
    storage: weights and every layer output are rounded to bf16, accumulation stays f32.
    Returns dict of SparseTensor: x_conv1..x_conv4, out.
    """

    def W(name):
        w = _f32(params[name]).copy()
        return round_bf16(w) if (bf16 and name != "conv_input.0.weight") else w

    def BN(prefix):
        return bn_fold({k: params[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")})

    def post(x, feats, bn_prefix, residual=None):
        sc, sh = BN(bn_prefix)
        y = SparseTensor(scale_shift_act(feats, sc, sh, residual, True, bf16), x.indices, x.spatial_shape, x.batch_size)
        y.rulebooks = x.rulebooks
        return y

    x = SparseTensor(voxel_features, voxel_coords, sparse_shape, batch_size)
    outs, identity = {}, None
    for L in backbone_layers(np.asarray(voxel_features).shape[1], last_pad, table):
        w = W(L["conv"] + ".weight")
        assert list(w.shape) == [L["cout"], *L["kernel"], L["cin"]], (L["conv"], w.shape)
        if L["subm"]:
            if L["conv"].endswith(".conv1"):          # SparseBasicBlock.forward (:51-67): identity = x
                identity = x.features
            o = subm_conv(x, w, L["indice_key"])
        else:
            o = sparse_conv(x, w, L["stride"], L["padding"])
        x = post(o, o.features, L["bn"], residual=identity if L["residual"] else None)
        if L["output"]:
            outs[L["output"]] = x
    return outs


# ---- recall bookkeeping and class NMS (host logic around the IoU / NMS operators) ----------------
ALL_CLASS_NAMES = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle', 'bicycle',
                   'pedestrian', 'traffic_cone']                                   # detector3d_template.py:15-16
KNOWN3 = [ALL_CLASS_NAMES.index(x) + 1 for x in ('car', 'bicycle', 'pedestrian')]   # :17,21
KNOWN6 = [ALL_CLASS_NAMES.index(x) + 1 for x in ('car', 'construction_vehicle', 'trailer', 'barrier', 'bicycle',
                                                  'pedestrian')]                   # :18-19,22


def generate_recall_record(box_preds, recall_dict, gt_boxes, rois, thresh_list):
    """detector3d_template.py:314-399, statement by statement, in numpy (gt_boxes (G, 7+..+1) with the
    class label last and all-zero padding rows at the end; rois (R,7+) or None)."""
    if len(recall_dict) == 0:
        recall_dict = {'gt': 0, 'num_3known': 0, 'num_6known': 0, 'num_4unknown': 0, 'num_7unknown': 0}
        for t in thresh_list:
            for stem in ('roi_%s', 'rcnn_%s', 'rcnn_3known_%s', 'rcnn_6known_%s', 'rcnn_4unknown_%s', 'rcnn_7unknown_%s'):
                recall_dict[stem % str(t)] = 0
    cur_gt = np.asarray(gt_boxes, np.float32)
    k = len(cur_gt) - 1
    while k >= 0 and cur_gt[k].sum() == 0:
        k -= 1
    cur_gt = cur_gt[:k + 1]
    if cur_gt.shape[0] > 0:
        labels = cur_gt[:, -1].astype(np.int64)
        k3 = np.array([l in KNOWN3 for l in labels]); k6 = np.array([l in KNOWN6 for l in labels])
        recall_dict['num_3known'] += int(k3.sum()); recall_dict['num_6known'] += int(k6.sum())
        recall_dict['num_7unknown'] += int((~k3).sum()); recall_dict['num_4unknown'] += int((~k6).sum())
        iou_rcnn = boxes_iou3d(box_preds[:, :7], cur_gt[:, :7]) if box_preds.shape[0] > 0 else np.zeros((0, cur_gt.shape[0]), np.float32)
        iou_roi = boxes_iou3d(rois[:, :7], cur_gt[:, :7]) if rois is not None else None
        for t in thresh_list:
            if iou_rcnn.shape[0] > 0:
                hit = iou_rcnn.max(axis=0) > np.float32(t)
                recall_dict['rcnn_%s' % str(t)] += int(hit.sum())
                recall_dict['rcnn_3known_%s' % str(t)] += int((hit & k3).sum())
                recall_dict['rcnn_6known_%s' % str(t)] += int((hit & k6).sum())
                recall_dict['rcnn_7unknown_%s' % str(t)] += int((hit & ~k3).sum())
                recall_dict['rcnn_4unknown_%s' % str(t)] += int((hit & ~k6).sum())
            if iou_roi is not None:
                recall_dict['roi_%s' % str(t)] += int((iou_roi.max(axis=0) > np.float32(t)).sum())
        recall_dict['gt'] += cur_gt.shape[0]
    return recall_dict


def class_agnostic_nms(box_scores, box_preds, nms_type, thresh, pre_max, post_max, score_thresh=None):
    """model_nms_utils.py:6-27 in numpy (stable top-k; ties are the caller's concern)."""
    src = np.asarray(box_scores, np.float32)
    idx0 = np.arange(src.shape[0])
    if score_thresh is not None:
        m = src >= np.float32(score_thresh)
        idx0 = idx0[m]
    sc, bx = src[idx0], np.asarray(box_preds, np.float32)[idx0]
    if sc.shape[0] == 0:
        return np.zeros((0,), np.int64), np.zeros((0,), np.float32)
    order = np.argsort(-sc, kind="stable")[:min(pre_max, sc.shape[0])]
    keep = nms_gpu(bx[order][:, :7], sc[order], thresh, rotated=(nms_type == "nms_gpu"))
    sel = idx0[order[keep[:post_max]]]
    return sel, src[sel]


# ---- pseudo-label mixing host operators (pcdet/datasets/augmentor/pseudo_loader.py) ----------------
def bev_nms_cpu(boxes, scores, thresh=0.5):
    """pseudo_loader.py:29-55 statement by statement (the double Python loop), on the oracle's IoU."""
    boxes, scores = np.asarray(boxes, np.float32), np.asarray(scores, np.float32)
    N = boxes.shape[0]
    keep = np.ones((N,), bool)
    order = np.argsort(-scores, kind="stable")
    ious = boxes_iou_bev(boxes, boxes) if N else np.zeros((0, 0), np.float32)
    for i in range(N):
        if keep[i]:
            curr = order[i]
            for j in range(i + 1, N):
                if ious[curr, order[j]] > thresh:
                    keep[j] = False
    return order[keep]


def pseudo_points_in_boxes(points, boxes3d):
    """PseudoSampler.points_in_boxes (pseudo_loader.py:270-316) in numpy f32: corner-template extents,
    points centred and rotated by -heading (common_utils.rotate_points_along_z :35-57), inclusive faces."""
    points, boxes3d = np.asarray(points, np.float32), np.asarray(boxes3d, np.float32)
    N = boxes3d.shape[0]
    template = (np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]) / 2).astype(np.float32)
    corners = boxes3d[:, None, 3:6].repeat(8, axis=1) * template[None]
    p = points[:, None, :].repeat(N, axis=1)
    p[..., :3] = p[..., :3] - boxes3d[None, :, 0:3]
    p = p.transpose((1, 0, 2)).copy()
    a = -boxes3d[:, 6]
    ca, sa = np.cos(a).astype(np.float32), np.sin(a).astype(np.float32)
    x = p[..., 0] * ca[:, None] + p[..., 1] * (-sa[:, None])
    y = p[..., 0] * sa[:, None] + p[..., 1] * ca[:, None]
    p[..., 0], p[..., 1] = x, y
    lo, hi = corners.min(axis=1, keepdims=True), corners.max(axis=1, keepdims=True)
    in_box = ((p[..., 0] >= lo[..., 0]) & (p[..., 0] <= hi[..., 0]) & (p[..., 1] >= lo[..., 1]) & (p[..., 1] <= hi[..., 1]) &
              (p[..., 2] >= lo[..., 2]) & (p[..., 2] <= hi[..., 2]))
    return in_box, p
