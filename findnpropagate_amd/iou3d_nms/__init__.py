from . import iou3d_nms_cuda, iou3d_nms_utils  # noqa: F401
