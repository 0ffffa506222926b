from .detr_panoptic import PanopticHead
from .nn import FPNstyleCNN, MHAttentionMap
from .detr_r50_panoptic import DetrR50Panoptic
from .detr_r50_panoptic_finetune import DetrR50PanopticFinetune

__all__ = ["PanopticHead", "MHAttentionMap", "FPNstyleCNN", "DetrR50Panoptic", "DetrR50PanopticFinetune"]
