"""pcdet/models/backbones_2d: the map_to_bev step that follows the 3D backbone, and BaseBEVBackbone with its first block
evaluated on the sparse rows (the only 2D layer that touches the sparse tensor; the rest stays the reference's torch modules)."""
from .base_bev_backbone import BaseBEVBackbone
from .map_to_bev import HeightCompression

__all__ = {"HeightCompression": HeightCompression, "BaseBEVBackbone": BaseBEVBackbone}
