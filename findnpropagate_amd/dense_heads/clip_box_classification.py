"""CLIPBoxClassification (pcdet/models/dense_heads/clip_box_classification.py:68-379): re-label 3D boxes by
CLIP similarity of their image crops — the "CLIP-crop scoring" of the pseudo-label extraction.

Same forward(batch_dict, pred_dicts, keep_crops=False, relabel=True) contract: per scene, every box is
projected into the six cameras, each (box, camera) that shows it contributes one 224x224 crop (square,
>= 64 px, anchored at the clipped corner box), the crops go through the image encoder, the class
probabilities are averaged over the cameras that see the box, and pred_labels / pred_scores are replaced
(orig_labels keeps the old labels).

What differs: the geometry and the resampling run in two launches for all (box, camera) pairs
(`fnp_clipcrop_plan`, `fnp_clipcrop_sample`) instead of a Python loop with one grid_sample and several host
round trips per pair, and the encoder is called once per scene on all crops (the reference calls it once
per camera).  The encoder is third-party: pass `clip_model` as the reference does (needs the `clip`
package, absent from this image) or any object with `encode_image(images)` and `logit_scale`, plus
`text_features` (10, D)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lib as _l

ALL_CLASS_NAMES = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle', 'bicycle',
                   'pedestrian', 'traffic_cone']


class CLIPBoxClassification(nn.Module):
    def __init__(self, image_size=[900, 1600], clip_model="ViT-L/14", ensembling=None, text_features=None):
        super().__init__()
        self.image_order = [2, 0, 1, 5, 3, 4]
        self.image_size = list(image_size)
        self.all_class_names = list(ALL_CLASS_NAMES)
        if isinstance(clip_model, str):
            try:
                import clip  # type: ignore
            except ImportError as e:
                raise _l.FnpError("the `clip` package is not installed: pass an encoder object "
                                  "(encode_image, logit_scale) and text_features instead of a model name") from e
            model, _ = clip.load(clip_model, device='cuda')
            self.clip = model
            if text_features is None:
                with torch.no_grad():
                    tok = clip.tokenize(self.all_class_names).cuda()          # ensembling=None branch (:94-95)
                    text_features = model.encode_text(tok)
        else:
            self.clip = clip_model
        assert text_features is not None and text_features.shape[0] == len(self.all_class_names)
        self.text_features = text_features
        self.min_crop_size = 64
        self.crop_size = 224
        # the [0, 1] sampling positions of the reference's grid (:99-100, 318-319), from the same torch calls
        g = F.affine_grid(theta=torch.eye(2, 3).unsqueeze(0), size=[1, 3, self.crop_size, self.crop_size], align_corners=False)
        g = (g - g.min()) / (g.max() - g.min())
        self._unit = g[0, 0, :, 0].contiguous()          # along the width == along the height
        self.crop_infos = None

    # ---- the two launches -----------------------------------------------------------------------
    def plan(self, batch_dict, boxes, b):
        L = _l.load()
        dev = boxes.device
        aug = batch_dict['lidar_aug_matrix'][b].detach().float().cpu()
        rinv = torch.inverse(aug[:3, :3]).contiguous()                      # (:205-207)
        trans = aug[:3, 3].contiguous()
        l2i = batch_dict['lidar2image'][b].detach().float().to(dev).contiguous()
        iaug = batch_dict['img_aug_matrix'][b].detach().float().to(dev).contiguous()
        n = boxes.shape[0]
        rect = torch.empty((n, 6, 4), dtype=torch.float32, device=dev)
        mask = torch.empty((n, 6), dtype=torch.uint8, device=dev)
        b7 = boxes[:, :7].detach().float().contiguous()
        rc = L.fnp_clipcrop_plan(_l.ptr(b7), n, rinv.data_ptr(), trans.data_ptr(), _l.ptr(l2i), _l.ptr(iaug), self.image_size[0],
                                 self.image_size[1], self.min_crop_size, _l.ptr(rect), _l.ptr(mask), _l.stream())
        _l.check(rc, "fnp_clipcrop_plan")
        return rect, mask

    def sample(self, images, rect, pairs):
        L = _l.load()
        images = images.contiguous()
        assert images.dim() == 4 and images.shape[0] == 6 and tuple(images.shape[2:]) == tuple(self.image_size)
        m, C = pairs.shape[0], images.shape[1]
        crops = torch.empty((m, C, self.crop_size, self.crop_size), dtype=images.dtype, device=images.device)
        unit = self._unit.to(images.device)
        rc = L.fnp_clipcrop_sample(_l.ptr(images), _l.dtype_code(images), C, self.image_size[0], self.image_size[1], _l.ptr(rect),
                                   _l.ptr(pairs), m, _l.ptr(unit), self.crop_size, _l.ptr(crops), _l.stream())
        _l.check(rc, "fnp_clipcrop_sample")
        return crops

    def get_clip_logits(self, images):
        """:168-185 — cosine similarity x exp(logit_scale)."""
        image_features = self.clip.encode_image(images)
        image_features = image_features / image_features.norm(dim=1, keepdim=True)
        text = self.text_features.to(image_features.device).to(image_features.dtype)
        text = text / text.norm(dim=1, keepdim=True)
        logit_scale = self.clip.logit_scale.exp()
        logits_per_image = logit_scale * image_features @ text.t()
        return logits_per_image, logits_per_image.t()

    # ---- reference contract ---------------------------------------------------------------------
    def forward(self, batch_dict, pred_dicts, keep_crops=False, relabel=True):
        batch_size = batch_dict['batch_size']
        images = batch_dict['camera_imgs']
        if keep_crops:
            self.crop_infos = dict(crops=[], logits=[])
        for b in range(batch_size):
            cur = pred_dicts[b]
            boxes = cur['pred_boxes']
            N = boxes.shape[0]
            if N == 0:
                continue
            dev = boxes.device
            rect, mask = self.plan(batch_dict, boxes, b)
            # crops in the reference's order: cameras in image_order, boxes ascending inside a camera
            order = torch.tensor(self.image_order, device=dev)
            has = rect[:, order, 3].t() > 0                                  # (6 in image_order, N)
            cam_pos, box_idx = torch.nonzero(has, as_tuple=True)             # one host sync (sizes the crop batch)
            pairs = torch.stack([box_idx, order[cam_pos]], 1).to(torch.int32).contiguous()
            box_probs = torch.zeros((N, 6, len(self.all_class_names)), device=dev, dtype=torch.half)
            if pairs.shape[0] > 0:
                crops = self.sample(images[b].to(dev), rect, pairs)
                with torch.no_grad():
                    logits, _ = self.get_clip_logits(crops)
                    probs = logits.softmax(dim=-1)
                box_probs[pairs[:, 0].long(), pairs[:, 1].long()] = probs.to(torch.half)
                if keep_crops:
                    self.crop_infos['crops'].append(crops.detach().cpu())
                    self.crop_infos['logits'].append(probs.detach().cpu())
            # mean over the cameras that show the box (:357-366)
            box_probs_mean = box_probs.sum(dim=1) / (1e-5 + mask.float().sum(dim=-1).unsqueeze(1))
            probs_max = torch.max(box_probs_mean.cpu(), dim=-1)
            pred_scores = torch.nan_to_num(probs_max.values, nan=0.0)
            pred_labels = probs_max.indices.flatten()
            if relabel:
                pred_dicts[b]['orig_labels'] = pred_dicts[b]['pred_labels'].clone()
                pred_dicts[b]['pred_labels'] = pred_labels + 1
                pred_dicts[b]['pred_scores'] = pred_scores
        if keep_crops and self.crop_infos['crops']:
            self.crop_infos['crops'] = torch.cat(self.crop_infos['crops'], dim=0)
            self.crop_infos['logits'] = torch.cat(self.crop_infos['logits'], dim=0)
        return pred_dicts
