from .data_processor import DataProcessor, VoxelGeneratorWrapper, mask_points_by_range  # noqa: F401
