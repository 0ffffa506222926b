"""DeformableDetrR50Panoptic re-headed for a custom class set (reference:
alonet/deformable_detr_panoptic/deformable_detr_r50_panoptic_finetune.py:13-86): the detector's classification head becomes ONE
``Linear(d_model, num_classes [+ 1 under softmax])`` shared by the decoder layers, bias at the 0.01 focal prior; optionally the
mask head's GroupNorm layers become BatchNorm2d; then a fine-tuned checkpoint is loaded on top."""
import math

import torch

from alonet.detr.detr_r50_finetune import load_finetuned
from alonet.detr_panoptic.detr_r50_panoptic_finetune import group_to_batch_norm

from .deformable_detr_r50_panoptic import DeformableDetrR50Panoptic


class DeformableDetrR50PanopticFinetune(DeformableDetrR50Panoptic):
    def __init__(self, num_classes, base_weights="deformable-detr-r50-panoptic", weights=None, use_bn_layers=False, *args, **kwargs):
        super().__init__(*args, weights=base_weights, **kwargs)
        det = self.detr
        det.background_class = num_classes if det.activation_fn == "softmax" else None
        num_classes += 1 if det.activation_fn == "softmax" else 0
        head = torch.nn.Linear(det.transformer.d_model, num_classes)
        head.bias.data = torch.ones(num_classes) * -math.log((1 - 0.01) / 0.01)
        det.class_embed = torch.nn.ModuleList([head for _ in range(det.transformer.decoder.num_layers)])
        if use_bn_layers:
            group_to_batch_norm(self.mask_head)
        if self.device is not None:
            self.to(self.device)
        import alo_hip

        alo_hip.invalidate_caches(self)
        load_finetuned(self, weights)
