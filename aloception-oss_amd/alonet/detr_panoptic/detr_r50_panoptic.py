"""PanopticHead over DETR R50 (reference: alonet/detr_panoptic/detr_r50_panoptic.py:10-45)."""
from alonet.detr import DetrR50

from .detr_panoptic import PanopticHead


class DetrR50Panoptic(PanopticHead):
    def __init__(self, num_classes=250, background_class=None, detr_weights=None, weights=None, *args, **kwargs):
        # background_class=None reaches Detr as None: the background id is then num_classes (the last id), not DetrR50's default 91
        base_model = DetrR50(num_classes=num_classes, background_class=background_class, weights=detr_weights,
                             device=kwargs.get("device", None))
        super().__init__(*args, DETR_module=base_model, weights=weights, **kwargs)
