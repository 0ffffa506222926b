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

__all__ = ["MSDeformAttn", "DeformableTransformer", "DeformableTransformerEncoder", "DeformableTransformerEncoderLayer",
           "DeformableTransformerDecoder", "DeformableTransformerDecoderLayer", "DeformableDETR", "DeformableDetrR50",
           "DeformableDetrR50Refinement"]
