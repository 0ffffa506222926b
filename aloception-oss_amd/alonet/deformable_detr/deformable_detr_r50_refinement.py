"""Import path of the reference (alonet/deformable_detr/deformable_detr_r50_refinement.py): the iterative box-refinement variant."""
from .deformable_detr_r50 import DeformableDetrR50Refinement

__all__ = ["DeformableDetrR50Refinement"]
