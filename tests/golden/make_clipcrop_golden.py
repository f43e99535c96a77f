#!/usr/bin/env python3
"""Golden vectors for the CLIP-crop scoring made by RUNNING THE REFERENCE's own
pcdet/models/dense_heads/clip_box_classification.py::CLIPBoxClassification.forward on CPU tensors.

Build container only (needs /root/reference); the committed tests/golden/clipcrop_seed*.npz hold inputs and
expected outputs, never reference source.  The third-party `clip` package (and its ViT weights) is absent, so
the harness gives the reference object a deterministic stand-in encoder (28x28 average pooling + a fixed
random projection, defined identically in tests/test_gpu_clipcrop.py); everything else that runs — corner
projection, integer truncation, on-image test, clipped bounding box, square crop >= 64 px, F.grid_sample,
softmax, half-precision accumulation over the cameras, arg-max — is the reference's code.
Images are procedural (function of pixel coordinates) so that they need not be stored."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_boxseeker_golden as H   # noqa: E402  (stubs, shells, cpu_only_torch)

from findnpropagate_amd import synthetic as syn  # noqa: E402


def procedural_images(h, w):
    """(6, 3, h, w) f32 in [0, 1]."""
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    out = np.empty((6, 3, h, w), np.float32)
    for c in range(6):
        for ch in range(3):
            out[c, ch] = 0.5 + 0.5 * np.sin(np.float32(0.013 * (ch + 1)) * x + np.float32(0.7 * c)) * np.cos(np.float32(0.011) * y + np.float32(0.3 * ch))
    return out


def projection_matrix():
    return torch.from_numpy(np.random.default_rng(77).standard_normal((192, 32)).astype(np.float32))


def text_features():
    return torch.from_numpy(np.random.default_rng(78).standard_normal((10, 32)).astype(np.float32))


class FakeClip:
    def __init__(self):
        self.logit_scale = torch.tensor(float(np.log(100.0))).half()
        self.P = projection_matrix()
        self.seen = []

    def encode_image(self, images):
        pooled = F.avg_pool2d(images.float(), 28).reshape(images.shape[0], -1)   # (M, 192)
        self.seen.append(pooled.numpy().copy())
        return (pooled @ self.P).half()        # the real CLIP runs in fp16 on the GPU (clip.load(..., device='cuda'))


def main():
    H.cpu_only_torch()
    H.load_reference()
    H.stub("pcdet.models.dense_heads.clip_box_cls_maskclip", CLIPTextEnsembling=object)
    H.stub("PIL", Image=object)
    sys.modules["PIL.Image"] = H.stub("PIL.Image")
    sys.modules["torchvision.utils"].make_grid = lambda *a, **k: None
    sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    sys.modules["torchvision.transforms"].ToPILImage = object
    mod = importlib.import_module("pcdet.models.dense_heads.clip_box_classification")
    for seed in ([int(a) for a in sys.argv[1:]] or [0, 1, 2]):
        head = object.__new__(mod.CLIPBoxClassification)
        torch.nn.Module.__init__(head)
        head.image_order = [2, 0, 1, 5, 3, 4]
        head.image_size = [900, 1600]
        head.all_class_names = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle', 'bicycle',
                                'pedestrian', 'traffic_cone']
        head.clip = FakeClip()
        head.text_features = text_features().half()
        head.min_crop_size = 64
        head.unif_grid = F.affine_grid(theta=torch.eye(2, 3).unsqueeze(0), size=[1, 3, 224, 224])

        _, boxes, cls = syn.make_scene(seed, return_boxes=True)
        rng = np.random.default_rng(500 + seed)
        boxes = boxes.astype(np.float32)
        boxes[:, :2] += rng.normal(scale=0.3, size=(boxes.shape[0], 2)).astype(np.float32)
        if seed == 2:
            # boxes the projection has to clip: around the ego itself (corners behind every camera and in front of it), a tall one
            # right at a camera (its rectangle runs off two image borders), one wholly behind the front camera, one whose corners
            # straddle a camera plane at a grazing angle (clip_box_classification.py:217-377: on-image test, clipped bounding box,
            # square crop >= 64 px)
            extra = np.array([[0.6, 0.1, -0.9, 4.6, 2.0, 1.7, 0.3],
                              [2.2, 1.6, 0.2, 1.2, 1.0, 3.2, 0.0],
                              [-6.0, 0.4, -0.8, 4.2, 1.9, 1.6, 3.0],
                              [1.0, -2.4, -1.0, 3.8, 1.8, 1.5, 1.35],
                              [0.2, 2.0, -1.2, 0.7, 0.7, 1.8, 0.8]], np.float32)
            boxes = np.concatenate([boxes[:6], extra]).astype(np.float32)
            cls = np.concatenate([cls[:6], np.array([0, 8, 0, 1, 8])])
        cams = syn.make_cameras(1)
        lidar_aug = np.eye(4, dtype=np.float32)
        if seed == 1:                                  # a non-trivial lidar augmentation (rotation + shift)
            a = 0.2
            lidar_aug[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            lidar_aug[:3, 3] = [0.5, -0.3, 0.1]
            boxes[:, :3] = boxes[:, :3] @ lidar_aug[:3, :3].T + lidar_aug[:3, 3]
            boxes[:, 6] += a
        img_aug = np.repeat(np.eye(4, dtype=np.float32)[None, None], 6, axis=1)
        images = procedural_images(900, 1600)
        bd = {"batch_size": 1, "camera_imgs": torch.from_numpy(images)[None], "camera_intrinsics": torch.from_numpy(cams["camera_intrinsics"]),
              "camera2lidar": torch.from_numpy(cams["camera2lidar"]), "img_aug_matrix": torch.from_numpy(img_aug),
              "lidar_aug_matrix": torch.from_numpy(lidar_aug)[None], "lidar2image": torch.from_numpy(cams["lidar2image"])}
        pd = [{"pred_boxes": torch.from_numpy(boxes), "pred_labels": torch.from_numpy(cls.astype(np.int64) + 1),
               "pred_scores": torch.rand(boxes.shape[0])}]
        head.forward(bd, pd, keep_crops=True)
        pooled = np.concatenate(head.clip.seen, 0) if head.clip.seen else np.zeros((0, 192), np.float32)
        out = os.path.join(HERE, f"clipcrop_seed{seed}.npz")
        np.savez_compressed(out, boxes=boxes, lidar_aug=lidar_aug, img_aug=img_aug[0], lidar2image=cams["lidar2image"][0],
                            pooled=pooled.astype(np.float32), probs=head.crop_infos["logits"].float().numpy(),
                            pred_labels=pd[0]["pred_labels"].numpy(), pred_scores=pd[0]["pred_scores"].float().numpy(),
                            orig_labels=pd[0]["orig_labels"].numpy())
        print(out, "boxes", boxes.shape[0], "crops", pooled.shape[0], "labels", pd[0]["pred_labels"].tolist())


if __name__ == "__main__":
    main()
