"""PanopticHead over Deformable-DETR R50 (reference: alonet/deformable_detr_panoptic/deformable_detr_r50_panoptic.py)."""
from alonet.common import load_weights
from alonet.deformable_detr import DeformableDetrR50, DeformableDetrR50Refinement
from alonet.detr_panoptic import PanopticHead


class DeformableDetrR50Panoptic(PanopticHead):
    def __init__(self, num_classes=250, activation_fn="sigmoid", with_box_refine=False, return_intermediate_dec=True,
                 deformable_weights=None, weights=None, strict_load_weights=True, *args, **kwargs):
        base = DeformableDetrR50Refinement if with_box_refine else DeformableDetrR50
        detector = base(num_classes=num_classes, weights=deformable_weights, activation_fn=activation_fn,
                        return_intermediate_dec=return_intermediate_dec, device=kwargs.get("device", None))
        super().__init__(*args, DETR_module=detector, weights=None, **kwargs)
        if weights is not None:
            load_weights(self, weights, self.device, strict_load_weights=strict_load_weights)
