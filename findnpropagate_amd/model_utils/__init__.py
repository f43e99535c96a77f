from . import model_nms_utils

__all__ = ["model_nms_utils"]
