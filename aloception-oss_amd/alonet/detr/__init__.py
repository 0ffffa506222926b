from .backbone import Backbone, FrozenBatchNorm2d, Joiner
from .transformer import Transformer
from .detr import Detr
from .detr_r50 import DetrR50
from .detr_r50_finetune import DetrR50Finetune

__all__ = ["Backbone", "FrozenBatchNorm2d", "Joiner", "Transformer", "Detr", "DetrR50", "DetrR50Finetune"]
