"""HeightCompression (pcdet/models/backbones_2d/map_to_bev/height_compression.py:4-26): the sparse
encoded tensor (B sites x 128 ch on a (2, 180, 180) grid) becomes the dense BEV map (B, 256, 180, 180).

Same constructor, attributes (`num_bev_features`) and batch_dict keys.  The densification is one pass of
`fnp_sparse_to_dense` that writes the whole map once (zeros included) from a cell -> row index map; the
module keeps the index map and, with `REUSE_OUTPUT: True` in the model cfg (our addition, default off: the
reference returns a fresh tensor per call), also the output buffer.  `spatial_features` has the dtype of the encoded
tensor's features — float32 by default (VoxelResBackBone8x FNP_OUT_DTYPE, the reference contract); OUT_DTYPE in this
module's cfg ('fp32' | 'bf16' | 'fp16') overrides it (the features are cast before the one densifying pass)."""
import torch.nn as nn

from ... import sparse as S


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = _get(model_cfg, "NUM_BEV_FEATURES")
        self.reuse_output = bool(_get(model_cfg, "REUSE_OUTPUT", False))
        od = _get(model_cfg, "OUT_DTYPE", "keep")
        import torch
        self.out_dtype = {"keep": None, "fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[str(od).lower()]
        self._ws = None
        self._out = None

    def forward(self, batch_dict):
        t = batch_dict["encoded_spconv_tensor"]
        feats = t.features.contiguous()
        if self.out_dtype is not None and feats.dtype != self.out_dtype:
            feats = feats.to(self.out_dtype)
        need = int(S._l.load().fnp_sparse_to_dense_workspace_bytes(t.batch_size, *t.spatial_shape))
        if self._ws is None or self._ws.numel() < need or self._ws.device != feats.device:
            import torch
            self._ws = torch.empty((need,), dtype=torch.uint8, device=feats.device)
        out = None
        if self.reuse_output and self._out is not None and self._out.dtype == feats.dtype and \
                tuple(self._out.shape) == (t.batch_size, feats.shape[1], *t.spatial_shape) and self._out.device == feats.device:
            out = self._out
        dense = S.to_dense(feats, t.indices, t.n_dev(), t.batch_size, list(t.spatial_shape), workspace=self._ws, out=out)
        if self.reuse_output:
            self._out = dense
        N, C, D, H, W = dense.shape
        batch_dict["spatial_features"] = dense.view(N, C * D, H, W)
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        if default is None and key not in cfg:
            raise KeyError(key)
        return cfg.get(key, default)
    return getattr(cfg, key) if default is None else getattr(cfg, key, default)
