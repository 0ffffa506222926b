from .deformable_detr_r50_panoptic import DeformableDetrR50Panoptic

__all__ = ["DeformableDetrR50Panoptic"]
