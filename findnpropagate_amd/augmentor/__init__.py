from . import pseudo_loader

__all__ = ["pseudo_loader"]
