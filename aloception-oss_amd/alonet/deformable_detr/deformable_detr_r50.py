"""Deformable-DETR with a ResNet-50 backbone (reference: alonet/deformable_detr/deformable_detr_r50.py:13-32)."""
from .backbone import Joiner
from .deformable_detr import DeformableDETR


class DeformableDetrR50(DeformableDETR):
    def __init__(self, *args, return_intermediate_dec=True, num_classes=91, **kwargs):
        backbone = Joiner(self.build_backbone("resnet50", True, True, False), self.build_positional_encoding(256))
        transformer = self.build_transformer(hidden_dim=256, dropout=0.1, nheads=8, dim_feedforward=1024, enc_layers=6,
                                             dec_layers=6, num_feature_levels=4, dec_n_points=4, enc_n_points=4,
                                             return_intermediate_dec=return_intermediate_dec)
        kwargs.setdefault("with_box_refine", False)
        super().__init__(backbone, transformer, *args, num_classes=num_classes, **kwargs)


class DeformableDetrR50Refinement(DeformableDetrR50):
    """Iterative bounding-box refinement variant (reference: deformable_detr_r50_refinement.py)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, with_box_refine=True, **kwargs)
