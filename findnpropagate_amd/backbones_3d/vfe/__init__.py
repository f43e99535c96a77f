from .mean_vfe import MeanVFE

__all__ = {"MeanVFE": MeanVFE}
