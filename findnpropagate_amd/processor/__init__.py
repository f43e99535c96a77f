from .data_processor import VoxelGeneratorWrapper  # noqa: F401
