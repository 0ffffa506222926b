from .backbone import Backbone, FrozenBatchNorm2d, Joiner

__all__ = ["Backbone", "FrozenBatchNorm2d", "Joiner"]
