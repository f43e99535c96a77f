"""Registry entries the reference looks up by NAME (pcdet/models/backbones_3d/__init__.py:9-20,
vfe/__init__.py:8-16)."""
from .spconv_backbone import VoxelBackBone8x, VoxelResBackBone8x
from .vfe.mean_vfe import MeanVFE

__all__ = {
    "VoxelBackBone8x": VoxelBackBone8x,
    "VoxelResBackBone8x": VoxelResBackBone8x,
    "MeanVFE": MeanVFE,
}
