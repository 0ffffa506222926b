from .ops.modules import MSDeformAttn
from .deformable_transformer import (
    DeformableTransformer,
    DeformableTransformerDecoder,
    DeformableTransformerDecoderLayer,
    DeformableTransformerEncoder,
    DeformableTransformerEncoderLayer,
)
from .deformable_detr import DeformableDETR
from .deformable_detr_r50 import DeformableDetrR50, DeformableDetrR50Refinement
from .deformable_detr_r50_finetune import DeformableDetrR50Finetune, DeformableDetrR50RefinementFinetune

__all__ = ["MSDeformAttn", "DeformableTransformer", "DeformableTransformerEncoder", "DeformableTransformerEncoderLayer",
           "DeformableTransformerDecoder", "DeformableTransformerDecoderLayer", "DeformableDETR", "DeformableDetrR50",
           "DeformableDetrR50Refinement", "DeformableDetrR50Finetune", "DeformableDetrR50RefinementFinetune"]
