"""pcdet/models/backbones_2d: only the map_to_bev step that follows the 3D backbone is built."""
from .map_to_bev import HeightCompression

__all__ = {"HeightCompression": HeightCompression}
