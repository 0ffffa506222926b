"""DetrR50Panoptic re-headed for a custom class set (reference: alonet/detr_panoptic/detr_r50_panoptic_finetune.py:11-72): the
detector's ``class_embed`` becomes ``Linear(hidden_dim, num_classes + 1)``, optionally the mask head's five GroupNorm layers
become BatchNorm2d (``use_bn_layers``), then a fine-tuned checkpoint is loaded on top."""
import torch

from alonet.detr.detr_r50_finetune import load_finetuned

from .detr_r50_panoptic import DetrR50Panoptic


def group_to_batch_norm(mask_head):
    for i in range(1, 6):
        name = "gn" + str(i)
        setattr(mask_head, name, torch.nn.BatchNorm2d(getattr(mask_head, name).num_channels))


class DetrR50PanopticFinetune(DetrR50Panoptic):
    def __init__(self, num_classes, background_class=None, base_weights="detr-r50-panoptic", weights=None, use_bn_layers=False,
                 *args, **kwargs):
        super().__init__(*args, weights=base_weights, **kwargs)
        self.detr.background_class = num_classes if background_class is None else background_class
        self.detr.num_classes = num_classes + 1
        self.detr.class_embed = torch.nn.Linear(self.detr.hidden_dim, self.detr.num_classes)
        if use_bn_layers:
            group_to_batch_norm(self.mask_head)
        if self.device is not None:
            self.to(self.device)
        import alo_hip

        alo_hip.invalidate_caches(self)
        load_finetuned(self, weights)
