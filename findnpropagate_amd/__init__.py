"""findnpropagate_amd — MI355X-native hot path of djamahl99/findnpropagate.

Hand-written gfx950 HIP kernels (csrc/) behind a C ABI (include/fnp.h), with a host side that
mirrors the reference's operator interfaces for this path:

  findnpropagate_amd.roiaware_pool3d   <-> pcdet.ops.roiaware_pool3d
  findnpropagate_amd.iou3d_nms         <-> pcdet.ops.iou3d_nms
  findnpropagate_amd.spconv            <-> spconv / spconv.pytorch (subset pcdet uses)
  findnpropagate_amd.processor         <-> pcdet.datasets.processor (VoxelGeneratorWrapper)
  findnpropagate_amd.backbones_3d      <-> pcdet.models.backbones_3d (MeanVFE, VoxelResBackBone8x)

There is no CPU fallback: operators raise if libfnp_hip.so is missing or a CPU tensor is passed.
"""
__all__ = ["lib", "sparse", "spconv", "iou3d_nms", "roiaware_pool3d", "processor", "backbones_3d"]
