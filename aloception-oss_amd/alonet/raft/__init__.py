from .corr import CorrBlock

__all__ = ["CorrBlock"]
