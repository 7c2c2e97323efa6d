from .TFE import TFE, TFEBatch

__all__ = ["TFE", "TFEBatch"]
