"""DETR R50 re-headed for a custom class set (reference: alonet/detr/detr_r50_finetune.py:12-57): the base model is built with
the checkpoint's own classes so that ``base_weights`` loads strictly, then ``class_embed`` becomes a fresh
``Linear(hidden_dim, num_classes + 1)`` (the extra output is the background class, by default the last id) and a fine-tuned
checkpoint, if given, is loaded on top.  ``base_weights=None`` skips the base checkpoint (no download in this offline build)."""
from torch import nn

from alonet.common import load_weights

from .detr_r50 import DetrR50


def load_finetuned(model, weights):
    if weights is not None:
        if ".pth" in weights or ".ckpt" in weights:
            load_weights(model, weights, model.device)
        else:
            raise ValueError(f"Unknown weights: '{weights}'")


class DetrR50Finetune(DetrR50):
    def __init__(self, num_classes, background_class=None, base_weights="detr-r50", weights=None, *args, **kwargs):
        super().__init__(*args, background_class=background_class, weights=base_weights, **kwargs)
        self.background_class = num_classes if background_class is None else background_class
        self.num_classes = num_classes + 1
        self.class_embed = nn.Linear(self.hidden_dim, self.num_classes)
        if self.device is not None:
            self.class_embed = self.class_embed.to(self.device)
        load_finetuned(self, weights)
