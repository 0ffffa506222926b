from .detr_panoptic import PanopticHead
from .nn import FPNstyleCNN, MHAttentionMap

__all__ = ["PanopticHead", "MHAttentionMap", "FPNstyleCNN"]
