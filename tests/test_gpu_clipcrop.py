"""CLIP-crop scoring against golden vectors produced by the reference's own
CLIPBoxClassification.forward (tests/golden/make_clipcrop_golden.py): same (box, camera) crop list in the same
order, crops equal through the stand-in encoder's 28x28 average pooling (2e-4), class probabilities (fp16 in
the reference) within 1e-2, re-labelled classes equal wherever the top-2 margin exceeds that tolerance."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def procedural_images(h, w):
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    out = np.empty((6, 3, h, w), np.float32)
    for c in range(6):
        for ch in range(3):
            out[c, ch] = 0.5 + 0.5 * np.sin(np.float32(0.013 * (ch + 1)) * x + np.float32(0.7 * c)) * np.cos(np.float32(0.011) * y + np.float32(0.3 * ch))
    return out


class FakeClip:   # the stand-in encoder of the golden generator
    def __init__(self, dev):
        self.logit_scale = torch.tensor(float(np.log(100.0)), device=dev).half()
        self.P = torch.from_numpy(np.random.default_rng(77).standard_normal((192, 32)).astype(np.float32)).to(dev)
        self.seen = []

    def encode_image(self, images):
        pooled = F.avg_pool2d(images.float(), 28).reshape(images.shape[0], -1)
        self.seen.append(pooled.cpu().numpy())
        return (pooled @ self.P).half()


@pytest.mark.parametrize("seed", [0, 1, 2])   # (2: boxes around the ego and at a camera — corners behind cameras, rectangles clipped at two borders)
def test_clip_crop_scoring_matches_reference_golden(cuda, seed):
    from findnpropagate_amd.dense_heads import CLIPBoxClassification
    g = np.load(os.path.join(GOLD, f"clipcrop_seed{seed}.npz"))
    text = torch.from_numpy(np.random.default_rng(78).standard_normal((10, 32)).astype(np.float32)).half().to(cuda)
    clip = FakeClip(cuda)
    head = CLIPBoxClassification(image_size=[900, 1600], clip_model=clip, text_features=text)
    images = torch.from_numpy(procedural_images(900, 1600)).to(cuda)
    bd = {"batch_size": 1, "camera_imgs": images[None], "img_aug_matrix": torch.from_numpy(g["img_aug"])[None].to(cuda),
          "lidar_aug_matrix": torch.from_numpy(g["lidar_aug"])[None].to(cuda), "lidar2image": torch.from_numpy(g["lidar2image"])[None].to(cuda)}
    pd = [{"pred_boxes": torch.from_numpy(g["boxes"]).to(cuda), "pred_labels": torch.from_numpy(g["orig_labels"]).to(cuda),
           "pred_scores": torch.zeros(g["boxes"].shape[0], device=cuda)}]
    head.forward(bd, pd, keep_crops=True)
    pooled = np.concatenate(clip.seen, 0) if clip.seen else np.zeros((0, 192), np.float32)
    assert pooled.shape == g["pooled"].shape, "same (box, camera) crop list"
    np.testing.assert_allclose(pooled, g["pooled"], rtol=0, atol=2e-4)
    probs = head.crop_infos["logits"].float().numpy()
    np.testing.assert_allclose(probs, g["probs"], rtol=0, atol=1e-2)
    np.testing.assert_allclose(pd[0]["pred_scores"].float().numpy(), g["pred_scores"], rtol=0, atol=1e-2)
    assert np.array_equal(pd[0]["orig_labels"].cpu().numpy(), g["orig_labels"])
    got, want = pd[0]["pred_labels"].cpu().numpy(), g["pred_labels"]
    assert (got == want).mean() >= 0.9 and got.min() >= 1 and got.max() <= 10


def test_crop_equals_grid_sample(cuda, rng):
    """fnp_clipcrop_sample against torch's own F.grid_sample on the grid the reference builds (:313-334)."""
    from findnpropagate_amd.dense_heads import CLIPBoxClassification

    class Enc:
        logit_scale = torch.zeros((), device="cuda")

        def encode_image(self, x):
            return x.mean(dim=(2, 3))
    head = CLIPBoxClassification(image_size=[120, 200], clip_model=Enc(), text_features=torch.ones(10, 3))
    images = torch.from_numpy(rng.uniform(0, 1, (6, 3, 120, 200)).astype(np.float32)).to(cuda)
    rect = torch.zeros((3, 6, 4), device=cuda)
    rect[0, 2] = torch.tensor([10., 20., 64., 1.])
    rect[1, 5] = torch.tensor([150., 0., 90., 1.])          # runs off the right and bottom edges: zero padding
    rect[2, 0] = torch.tensor([0., 60., 70., 1.])
    pairs = torch.tensor([[0, 2], [1, 5], [2, 0]], dtype=torch.int32, device=cuda)
    for dt in (torch.float32, torch.float16):
        got = head.sample(images.to(dt), rect, pairs)
        unit = head._unit.to(cuda)
        for m, (b, c) in enumerate(pairs.tolist()):
            x1, y1, side, _ = rect[b, c].tolist()
            gx, gy = unit[None, :].expand(224, 224) * side + x1, unit[:, None].expand(224, 224) * side + y1
            grid = torch.stack([(gx / 200) * 2 - 1, (gy / 120) * 2 - 1], -1)[None]
            # (torch's f16 grid_sample also rounds the sampling positions to f16; the kernel keeps them in f32,
            #  so both dtypes are held against the f32 result)
            want = F.grid_sample(images[[c]], grid, align_corners=False)
            tol = 1e-4 if dt == torch.float32 else 2e-3   # 1e-4: the fp32 bar of BASELINE.json (white-noise image, rounding of the sampling positions)
            assert (got[m].float() - want[0].float()).abs().max() <= tol
