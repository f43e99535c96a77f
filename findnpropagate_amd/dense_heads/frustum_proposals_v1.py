"""Greedy Box Seeker (FrustumProposerOG) on MI355X.

Mirrors pcdet/models/dense_heads/frustum_proposals_v1.py: same constructor signature and
PARAMS handling (:142-318), same batch_dict contract (:535-552), same return types of
get_proposals (:1055-1067) / get_bboxes / forward (:1547-1573).  The per-frustum work (hot loops
2-4: point selection, depth quantiles, frustum geometry, candidate generation, 2D IoU, point
density, scoring, top-1) runs as ONE launch of csrc/boxseeker.hip for all frustums of the batch;
the host only enumerates the frustums (per-camera 2D NMS of a few dozen CPU boxes, :582-594) and
prepares the 3x3 camera matrices.  Host syncs per batch: 2 (scene sizes, result read-back) —
the reference does >= 60 per frustum.
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import lib as _l


# the kernel's matrices made on the device (fnp_seeker_prepare_matrices); FNP_SEEKER_HOST_MATRICES=1: on the host with torch.inverse
MATRICES_ON_DEVICE = os.environ.get("FNP_SEEKER_HOST_MATRICES", "0") != "1"


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def boxes_to_corners_3d(boxes3d):
    """pcdet/utils/box_utils.py:28-53 with rotate_points_along_z (common_utils.py:35-57)."""
    template = boxes3d.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                                   [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    corners3d = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    angle = boxes3d[:, 6]
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(angle.shape[0]), angle.new_ones(angle.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    corners3d = torch.matmul(corners3d.view(-1, 8, 3), rot).view(-1, 8, 3)
    corners3d += boxes3d[:, None, 0:3]
    return corners3d


def box_iou(boxes1, boxes2):
    """torchvision.ops.box_iou (the reference calls it on CPU tensors, :1409)."""
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms, coordinate-trick form (:587).  A few dozen CPU boxes per
    camera: host work in the reference too.  Returns kept indices, score descending."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    b = boxes + offsets[:, None]
    order = torch.argsort(scores, descending=True, stable=True)
    iou = box_iou(b[order], b[order])
    n = order.numel()
    removed = torch.zeros(n, dtype=torch.bool)
    later = torch.arange(n)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(int(order[i]))
        removed |= (iou[i] > iou_threshold) & (later > i)
    return torch.tensor(keep, dtype=torch.int64)


class FrustumProposerOG(nn.Module):
    def __init__(self, model_cfg=None, input_channels=None, num_class=None, class_names=None, grid_size=None,
                 point_cloud_range=None, voxel_size=None, predict_boxes_when_training=True,
                 lq=0.336, uq=0.356, iou_w=0.95, dst_w=0.226, dns_w=0.05,
                 min_cam_iou=0.3, size_min=0.957, size_max=1.2, ry_min=0.0, ry_max=math.pi, cq=0.46, num_mags=6,
                 max_dist=50, num_sizes=4, num_rotations=10, topk=1, nms_2d=0.7, nms_3d=1.0, score_thr=0.1,
                 nms_normal=0.7, clamp_bottom=0, image_detector=None):
        super().__init__()
        self.MULTICAM_IOU = bool(_get(model_cfg, 'MULTICAM_IOU', False))
        self.OCCL_MULT = bool(_get(model_cfg, 'OCCL_MULT', False))
        self.MULT = bool(_get(model_cfg, 'MULT', False))
        self.search_depth, self.clamp_bottom, self.rand_center = None, 0, False
        self.aln_w = self.ego_w = self.occl_w = 0
        p = _get(model_cfg, 'PARAMS', None)
        if p is not None:   # :167-197
            g = lambda k, d: p.get(k, d)
            lq, uq, iou_w, dst_w, dns_w = g('lq', lq), g('uq', uq), g('iou_w', iou_w), g('dst_w', dst_w), g('dns_w', dns_w)
            self.aln_w, self.ego_w = g('aln_w', 0), g('ego_w', 0)
            min_cam_iou, size_min, size_max, cq = g('min_cam_iou', min_cam_iou), g('size_min', size_min), g('size_max', size_max), g('cq', cq)
            num_mags, max_dist, num_sizes, num_rotations = g('num_mags', num_mags), g('max_dist', max_dist), g('num_sizes', num_sizes), g('num_rotations', num_rotations)
            topk, score_thr, nms_2d, nms_3d, nms_normal = g('topk', topk), g('score_thr', score_thr), g('nms_2d', nms_2d), g('nms_3d', nms_3d), g('nms_normal', nms_normal)
            self.clamp_bottom, self.rand_center = g('clamp_bottom', clamp_bottom), g('rand_center', False)
            self.occl_w, ry_min, ry_max = g('occl_w', 0), g('ry_min', ry_min), g('ry_max', ry_max)
            self.search_depth = g('search_depth', None)
        self.nms_normal, self.topk, self.nms_3d, self.nms_2d = nms_normal, topk, nms_3d, nms_2d
        assert self.nms_3d == 0, 'DO NOT USE!'   # :209
        self.image_order = [2, 0, 1, 5, 3, 4]
        self.image_size = [900, 1600]
        self.point_cloud_range = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
        self.mags_min, self.mags_max = 0.0, 1.0
        self.ry_min, self.ry_max, self.size_min, self.size_max = ry_min, ry_max, size_min, size_max
        self.num_mags, self.num_sizes, self.num_rotations = num_mags, num_sizes, num_rotations
        self.max_dist, self.score_thr = max_dist, score_thr
        self.lq, self.uq, self.cq = lq, uq, cq
        self.iou_w, self.dst_w, self.dns_w, self.min_cam_iou = iou_w, dst_w, dns_w, min_cam_iou
        self.box_fmt = _get(model_cfg, 'BOX_FORMAT', 'xyxy')
        # aln_w: the reference's own code path raises as soon as a candidate holds more than three points — IndexError at :988,
        # a (1, N) mask on an (N, 3) tensor (fixture tests/golden/boxseeker_seed26.npz records it) — and would otherwise draw a
        # randomised torch.pca_lowrank per candidate: there is no behaviour to mirror.  Everything else the constructor reads
        # is built (topk through the 3D NMS, search_depth, rand_center, MULTICAM_IOU, occl_w / OCCL_MULT, xywh detections,
        # num_mags 0): none of it is set by a shipped config (tools/cfgs/nuscenes_box_seeker_proposals.yaml:83).
        if self.aln_w:
            raise NotImplementedError("FrustumProposerOG: aln_w > 0 is not built — the reference itself raises IndexError "
                                      "(frustum_proposals_v1.py:988) whenever a candidate box holds more than three points")
        if int(self.topk) < 1:
            raise ValueError("topk must be >= 1")
        self.rand_noise = None   # rand_center: optional callable (F, num_mags, device) -> (F, num_mags, 3) draws (default torch.randn)

        anchors = torch.tensor([[4.63, 1.97, 1.74], [6.93, 2.51, 2.84], [6.37, 2.85, 3.19], [10.5, 2.94, 3.47],
                                [12.29, 2.90, 3.87], [0.50, 2.53, 0.98], [2.11, 0.77, 1.47], [1.70, 0.60, 1.28],
                                [0.73, 0.67, 1.77], [0.41, 0.41, 1.07]], dtype=torch.float32)   # :270-281
        self.anchors = anchors
        size_variations = torch.linspace(self.size_min, self.size_max, steps=self.num_sizes)
        base_rotations = torch.linspace(self.ry_min, self.ry_max, steps=self.num_rotations)
        base_boxes = torch.zeros((anchors.shape[0], self.num_rotations, self.num_sizes, 7))
        for i in range(anchors.shape[0]):
            base_boxes[i, :, :, [3, 4, 5]] = anchors[i]
        for i in range(self.num_rotations):
            base_boxes[:, i, :, -1] = base_rotations[i]
        for i, m in enumerate(size_variations):
            base_boxes[:, :, i, [3, 4, 5]] = base_boxes[:, :, i, [3, 4, 5]] * m
        self.register_buffer("base_corners", boxes_to_corners_3d(base_boxes.reshape(-1, 7)).reshape(anchors.shape[0], -1, 8, 3).contiguous(), persistent=False)
        self.register_buffer("base_boxes", base_boxes.reshape(anchors.shape[0], -1, 7).contiguous(), persistent=False)
        # num_mags 0: one search position at the near face of the frustum (:831-834)
        self.register_buffer("mags", torch.linspace(self.mags_min, self.mags_max, self.num_mags) if self.num_mags > 0 else torch.zeros(1),
                             persistent=False)

        self.image_detector = image_detector
        if self.image_detector is None:
            self.image_detector = self._default_detector(model_cfg, class_names)
        self.last_debug = None
        self._dev_tables = {}
        self._ws_cache = {}      # (device, stream) -> the Box Seeker kernel's scratch buffer (launch)
        self._order_c = None

    @staticmethod
    def _default_detector(model_cfg, class_names):
        """PreprocessedGLIP / PreprocessedDetector as the reference constructs them (:254-267), from this package's
        own loaders (findnpropagate_amd.preprocessed_detector: no maskrcnn_benchmark needed to read the GLIP file)."""
        from ..preprocessed_detector import PreprocessedDetector, PreprocessedGLIP
        preds_path = _get(model_cfg, 'PREDS_PATH', '')
        if 'PreprocessedGLIP' in preds_path:
            kw = {k: _get(model_cfg, key) for k, key in (('pred_pth', 'GLIP_PRED_PTH'), ('meta_coco', 'GLIP_META_COCO'))
                  if _get(model_cfg, key) is not None}      # (optional overrides of the reference's hard-coded paths)
            return PreprocessedGLIP(class_names=class_names, **kw)
        cams = ['CAM_BACK', 'CAM_BACK_LEFT', 'CAM_BACK_RIGHT', 'CAM_FRONT', 'CAM_FRONT_LEFT', 'CAM_FRONT_RIGHT']
        paths = _get(model_cfg, 'PREDS_PATHS', [preds_path + f"{c}.json" for c in cams])
        return PreprocessedDetector(paths, class_names=class_names)

    # ------------------------------------------------------------------------------------------
    def _params(self, point_stride, xyz_offset):
        p = _l.SeekerParams()
        p.lq, p.uq, p.cq = float(self.lq), float(self.uq), float(self.cq)
        p.iou_w, p.dst_w, p.dns_w = float(self.iou_w), float(self.dst_w), float(self.dns_w)
        p.min_cam_iou, p.max_dist = float(self.min_cam_iou), float(self.max_dist)
        p.num_mags, p.num_rotations, p.num_sizes = max(int(self.num_mags), 1), int(self.num_rotations), int(self.num_sizes)
        p.topk, p.clamp_bottom = int(self.topk), int(self.clamp_bottom)
        p.nms_normal = float(self.nms_normal)
        p.search_depth = float(self.search_depth) if self.search_depth is not None else 0.0
        p.occl_w, p.occl_mult, p.multicam = float(self.occl_w), int(bool(self.OCCL_MULT)), int(bool(self.MULTICAM_IOU))
        p.count_only, p.num_frustums, p.npts_all, p.rand_noise = 0, 0, None, None
        p.image_h, p.image_w = int(self.image_size[0]), int(self.image_size[1])
        p.point_stride, p.xyz_offset = int(point_stride), int(xyz_offset)
        p.has_img_aug, p.mult, p.ego_w = 0, int(bool(self.MULT)), float(self.ego_w)
        return p

    def enumerate_frustums(self, batch_dict):
        """Rows [scene, cam, x1, y1, x2, y2, label, score] in the reference's order (:561-594):
        per scene, per camera in image_order, torchvision.batched_nms (coordinate trick, f32) then the
        score threshold — host work on a few dozen CPU boxes, done by the native host loop
        fnp_host_enumerate_frustums (include/fnp.h)."""
        if self.image_detector is None:
            raise RuntimeError("FrustumProposerOG needs an image_detector (PreprocessedGLIP predictions)")
        det_boxes, det_labels, det_scores, det_batch_idx, det_cam_idx = self.image_detector(batch_dict)
        L = _l.load()

        def host(t, dtype):   # (the detectors hand out CPU tensors of these dtypes already: nothing to convert then)
            if t.device.type == 'cpu' and t.dtype == dtype and t.is_contiguous():
                return t
            return t.detach().to('cpu', dtype).contiguous()
        boxes, scores = host(det_boxes, torch.float32), host(det_scores, torch.float32)
        labels, bidx, cidx = host(det_labels, torch.int64), host(det_batch_idx, torch.int64), host(det_cam_idx, torch.int64)
        D = boxes.shape[0]
        rows = torch.empty((max(D, 1), 8), dtype=torch.float32)
        order = self._order_c
        if order is None:
            order = self._order_c = (ctypes.c_int * len(self.image_order))(*self.image_order)
        n = L.fnp_host_enumerate_frustums(_l.ptr(boxes), _l.ptr(labels), _l.ptr(scores), _l.ptr(bidx), _l.ptr(cidx), D,
                                          int(batch_dict['batch_size']), order, len(self.image_order),
                                          float(self.nms_2d), float(self.score_thr), _l.ptr(rows), rows.shape[0])
        if n < 0:
            _l.check(n, "fnp_host_enumerate_frustums")
        rows = rows[:n]      # (a view of this call's own buffer)
        if self.box_fmt != 'xyxy':   # :596-601: [x, y, w, h] detections; the 2D NMS above ran on the raw numbers, as the reference's does
            rows[:, 4:6] += rows[:, 2:4]
        return rows

    @staticmethod
    def _matrices(batch_dict):
        """(B,21) scene and (B,6,45) camera matrices in f32, computed like :1431-1475 / :1509-1545, and whether a
        non-identity img_aug_matrix is among them."""
        aug = batch_dict['lidar_aug_matrix'].detach().cpu().float()
        l2i = batch_dict['lidar2image'].detach().cpu().float()
        c2l = batch_dict['camera2lidar'].detach().cpu().float()
        K = batch_dict['camera_intrinsics'].detach().cpu().float()
        B = aug.shape[0]
        R = aug[:, :3, :3]
        ia = batch_dict['img_aug_matrix'].detach().cpu().float() if 'img_aug_matrix' in batch_dict else None
        has_ia = ia is not None and not torch.equal(ia, torch.eye(4).expand_as(ia))
        # a few dozen 3x3 inverses: keep LAPACK on one thread (the intra-op pool costs milliseconds to wake up
        # for this; measured 0.9 -> 9 ms per call between 16 and 32 scenes)
        nthreads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            Rinv = torch.inverse(R)
            Kinv = torch.inverse(K[..., :3, :3])
            post_inv = torch.inverse(ia[..., :3, :3]) if has_ia else torch.eye(3).expand(B, 6, 3, 3)
        finally:
            torch.set_num_threads(nthreads)
        scene = torch.cat([R.reshape(B, 9), Rinv.reshape(B, 9), aug[:, :3, 3]], dim=1).contiguous()
        combine = c2l[..., :3, :3].matmul(Kinv)
        post = ia[..., :3, :3] if has_ia else torch.eye(3).expand(B, 6, 3, 3)
        post_t = ia[..., :3, 3] if has_ia else torch.zeros((B, 6, 3))
        cam = torch.cat([l2i[..., :3, :3].reshape(B, 6, 9), l2i[..., :3, 3], combine.reshape(B, 6, 9), c2l[..., :3, 3],
                         post.reshape(B, 6, 9), post_t, post_inv.reshape(B, 6, 9)], dim=2)
        return scene, cam.contiguous(), has_ia

    @staticmethod
    def _matrices_device(batch_dict, dev):
        """The same matrices made on the device by fnp_seeker_prepare_matrices from the batch's device tensors (no copy to the
        host, no synchronisation, no LAPACK call): (B,21), (B,6,45), and whether the batch carries an img_aug_matrix (applied
        whenever present, as the reference does: an identity changes no value)."""
        def f(k):   # (the collate hands out device f32 tensors already: three torch calls per matrix saved then)
            t = batch_dict[k]
            if t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad:
                return t
            return t.detach().to(dev, torch.float32, non_blocking=True).contiguous()
        aug, l2i, c2l, K = f('lidar_aug_matrix'), f('lidar2image'), f('camera2lidar'), f('camera_intrinsics')
        ia = f('img_aug_matrix') if 'img_aug_matrix' in batch_dict else None
        B = aug.shape[0]
        assert tuple(l2i.shape) == (B, 6, 4, 4) and tuple(c2l.shape) == (B, 6, 4, 4) and tuple(K.shape) == (B, 6, 4, 4)
        scene = torch.empty((B, 21), dtype=torch.float32, device=dev)
        cam = torch.empty((B, 6, 45), dtype=torch.float32, device=dev)
        _l.check(_l.load().fnp_seeker_prepare_matrices(_l.ptr(aug), _l.ptr(l2i), _l.ptr(c2l), _l.ptr(K), _l.ptr(ia), B, _l.ptr(scene), _l.ptr(cam),
                                                       _l.stream()), "fnp_seeker_prepare_matrices")
        return scene, cam, ia is not None

    def _tables(self, dev):
        """Device copies of the constant tables, kept alive across (asynchronous) launches."""
        key = str(dev)
        if key not in self._dev_tables:
            self._dev_tables[key] = (self.base_boxes.to(dev).contiguous(), self.base_corners.to(dev).contiguous(),
                                     self.mags.to(dev).contiguous())
        return self._dev_tables[key]

    def launch(self, batch_dict, debug=False, noise=None):
        """Enqueue the Box Seeker for every frustum of the batch and return WITHOUT synchronising:
        dict(frustums (F,8) f32 HOST rows [scene, cam, x1, y1, x2, y2, label, score] in the reference's enumeration
        order, d_frustums (the same on the device), out_valid (F,) i32, out_box (F,7), out_score (F,), out_best (F,) on
        the device, debug tensors), or None when the batch has no frustum.  batch_dict may carry 'points_per_scene' (host list of
        ints, the per-scene row counts the collate already knows): without it a batch of more than one scene costs one
        host sync for the scene sizes.
        With topk > 1 the tables are in ROW form: frustum f takes rows f * topk .. f * topk + topk - 1 of frustums / d_frustums /
        out_valid (1 for the boxes the 3D NMS kept) / out_box / out_score / out_best; out_count (F,) holds the number per frustum.
        noise: rand_center's draws, (F, num_mags, 3) on the device (default: torch.randn there, as the reference draws them)."""
        L = _l.load()
        points = batch_dict['points']
        _l.require_device(points)
        if points.dtype != torch.float32 or not points.is_contiguous() or points.requires_grad:
            points = points.detach().float().contiguous()
        dev = points.device
        B = int(batch_dict['batch_size'])
        frusts = self.enumerate_frustums(batch_dict)
        F = frusts.shape[0]
        if F == 0:
            return None
        # scenes are contiguous row ranges of the collated point tensor (dataset.py:221-245)
        pps = batch_dict.get('points_per_scene')
        if B == 1:      # the extraction script's batch size (:36,54): the scene is the whole tensor, no sync needed
            pps = [int(points.shape[0])]
        if pps is not None:
            assert len(pps) == B and sum(int(v) for v in pps) == points.shape[0]
            # scene offsets and the frustum table cross to the device in ONE pinned transfer (offsets as int32 bits in the f32 buffer)
            # (filled through numpy views: a torch indexing call costs ~5 us of host time, and this loop is bound by the host)
            host = torch.empty((B + 1 + F * 8,), dtype=torch.float32, pin_memory=True)
            hn = host.numpy()
            ho = hn[:B + 1].view(np.int32)
            ho[0] = 0
            np.cumsum(np.asarray(pps, dtype=np.int64), out=ho[1:], dtype=np.int32) if B > 1 else ho.__setitem__(1, int(pps[0]))
            hn[B + 1:] = frusts.numpy().reshape(-1)
            d_host = host.to(dev, non_blocking=True)
            offsets, d_fr = d_host[:B + 1].view(torch.int32), d_host[B + 1:].view(F, 8)
            max_pts = max(int(v) for v in pps)
        else:
            counts = torch.bincount(points[:, 0].long(), minlength=B)[:B]
            offsets = torch.zeros((B + 1,), dtype=torch.int32, device=dev)
            offsets[1:] = torch.cumsum(counts, 0).int()
            max_pts = int(counts.max().item())                                 # host sync (scene sizes)
            d_fr = frusts.to(dev, non_blocking=True)
        if MATRICES_ON_DEVICE:
            scene_m, cam_m, has_img_aug = self._matrices_device(batch_dict, dev)
        else:
            scene_m, cam_m, has_img_aug = self._matrices(batch_dict)
            scene_m, cam_m = scene_m.to(dev, non_blocking=True), cam_m.to(dev, non_blocking=True)
        prm = self._params(points.shape[1], 1)
        prm.has_img_aug = int(has_img_aug)
        NM, TK = max(int(self.num_mags), 1), int(self.topk)
        NC = NM * self.num_rotations * self.num_sizes
        # the kernel's scratch: one buffer per (device, stream), grown as needed — launches of a stream run in order, so the next one may
        # have it (a debug launch hands views of it out and takes a buffer of its own)
        need = int(L.fnp_boxseeker_workspace_bytes(F, max_pts))
        if debug:
            ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        else:
            wkey = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
            ws = self._ws_cache.get(wkey)
            if ws is None or ws.numel() < need:
                ws = self._ws_cache[wkey] = torch.empty((max(need, 1 << 20),), dtype=torch.uint8, device=dev)
        # (every output element is written by the kernel, frustums without points included: no clearing launches)
        out_all = torch.empty((F * (1 + TK * 9),), dtype=torch.float32, device=dev)
        out_valid = out_all[:F].view(torch.int32)
        out_box = out_all[F:F + F * TK * 7].view(F * TK, 7)
        out_score = out_all[F + F * TK * 7:F + F * TK * 8]
        out_best = out_all[F + F * TK * 8:].view(torch.int32)
        keep = []
        if self.rand_center:         # :847: the weighted centre plus unit Gaussian draws instead of positions along the frustum axis
            if noise is None:
                noise = self.rand_noise(F, NM, dev) if self.rand_noise is not None else torch.randn((F, NM, 3), dtype=torch.float32, device=dev)
            noise = noise.to(dev).float().contiguous()
            assert tuple(noise.shape) == (F, NM, 3)
            prm.rand_noise = _l.ptr(noise)
            keep.append(noise)
        t_boxes, t_corners, t_mags = self._tables(dev)
        if self.MULTICAM_IOU:        # pre-pass: which frustums hold points at all (only those enter the reference's lists, :683-692)
            npts_all = torch.zeros((F,), dtype=torch.int32, device=dev)
            prm.count_only = 1
            rc = L.fnp_boxseeker(_l.ptr(points), _l.ptr(offsets), B, max_pts, prm, _l.ptr(scene_m), _l.ptr(cam_m),
                                 _l.ptr(d_fr), F, _l.ptr(t_boxes), _l.ptr(t_corners), _l.ptr(t_mags), _l.ptr(ws), ws.numel(),
                                 _l.ptr(out_valid), _l.ptr(out_box), _l.ptr(out_score), _l.ptr(out_best),
                                 _l.ptr(npts_all), None, None, None, None, None, _l.stream())
            _l.check(rc, "fnp_boxseeker (point counts)")
            prm.count_only, prm.num_frustums, prm.npts_all = 0, F, _l.ptr(npts_all)
            keep.append(npts_all)
        dbg = {}
        if debug:
            dbg = dict(npts=torch.zeros((F,), dtype=torch.int32, device=dev), frust=torch.zeros((F, 8, 3), device=dev),
                       cand=torch.zeros((F, NC, 7), device=dev), iou=torch.zeros((F, NC), device=dev),
                       count=torch.zeros((F, NC), dtype=torch.int32, device=dev),
                       valid=torch.zeros((F, NC), dtype=torch.int32, device=dev))
        rc = L.fnp_boxseeker(_l.ptr(points), _l.ptr(offsets), B, max_pts, prm, _l.ptr(scene_m), _l.ptr(cam_m),
                             _l.ptr(d_fr), F, _l.ptr(t_boxes), _l.ptr(t_corners),
                             _l.ptr(t_mags), _l.ptr(ws), ws.numel(),
                             _l.ptr(out_valid), _l.ptr(out_box), _l.ptr(out_score), _l.ptr(out_best),
                             _l.ptr(dbg.get('npts')), _l.ptr(dbg.get('frust')), _l.ptr(dbg.get('cand')),
                             _l.ptr(dbg.get('iou')), _l.ptr(dbg.get('count')), _l.ptr(dbg.get('valid')), _l.stream())
        _l.check(rc, "fnp_boxseeker")
        if debug:
            # the lidar-frame points of every frustum as the kernel counted them (second half of its workspace: (F, stride, 3) f32,
            # the first npts[f] rows of frustum f) — what the reference hands to points_in_boxes_gpu per candidate (:812-815,930-932)
            stride = max(max_pts, 1)
            dbg['points_xyz'] = ws[F * stride * 12: 2 * F * stride * 12].view(torch.float32).view(F, stride, 3)
        out_count = out_valid
        if TK > 1:   # row form (device ops, no sync): one row per (frustum, rank), valid while rank < the frustum's box count
            out_valid = (torch.arange(TK, device=dev, dtype=torch.int32)[None, :] < out_count[:, None]).to(torch.int32).reshape(-1)
            frusts, d_fr = frusts.repeat_interleave(TK, 0), d_fr.repeat_interleave(TK, 0)
        # (the launch is asynchronous: the tensors it reads must outlive this call)
        return dict(frustums=frusts, d_frustums=d_fr, out_valid=out_valid, out_count=out_count, out_box=out_box, out_score=out_score,
                    out_best=out_best, dbg=dbg, _keep=(points, offsets, scene_m, cam_m, ws, keep))

    def get_proposals(self, batch_dict, debug=False, noise=None):
        """-> proposal_boxes (K,7) device f32, frust_labels (K,) long CPU, frust_scores (K,) f32 CPU,
        frust_batch_idx (K,) long CPU — the tuple of :1055-1067."""
        dev = batch_dict['points'].device
        empty = (torch.zeros((0, 7), device=dev), torch.zeros((0,), dtype=torch.long), torch.zeros((0,)),
                 torch.zeros((0,), dtype=torch.long))
        r = self.launch(batch_dict, debug=debug, noise=noise)
        if r is None:
            return empty
        frusts, out_box = r['frustums'], r['out_box']
        valid = r['out_valid'].bool().cpu()                                     # host sync (result read-back)
        if debug:
            self.last_debug = dict(frustums=frusts, has_box=valid, second_stage_scores=r['out_score'], best=r['out_best'], **r['dbg'])
        if not bool(valid.any()):
            return empty
        proposal_boxes = out_box[valid.to(dev)].reshape(-1, 7)
        frust_labels = frusts[valid, 6].long()
        frust_scores = frusts[valid, 7].clone()
        frust_batch_idx = frusts[valid, 0].long()
        return proposal_boxes, frust_labels, frust_scores, frust_batch_idx

    def forward(self, batch_dict):
        bboxes = self.get_bboxes(batch_dict)
        batch_dict['final_box_dicts'] = bboxes
        assert not self.training, "not trainable!"
        return batch_dict

    def get_bboxes(self, batch_dict):
        """:1554-1573 (one dict per scene; the reference shares ONE dict object between scenes, a bug
        that only stays hidden at batch size 1 — here each scene gets its own)."""
        proposed_boxes, proposed_labels, proposed_scores, proposed_batch_idx = self.get_proposals(batch_dict)
        ret = []
        for k in range(int(batch_dict['batch_size'])):
            mask = proposed_batch_idx == k
            ret.append(dict(pred_boxes=proposed_boxes[mask.to(proposed_boxes.device)], pred_scores=proposed_scores[mask],
                            pred_labels=proposed_labels[mask].int()))
        return ret
