"""Deformable-DETR R50 re-headed for a custom class set (reference: alonet/deformable_detr/deformable_detr_r50_finetune.py:10-136).

The base model is built with the checkpoint's own 91 classes so that ``base_weights`` loads strictly, then the classification
head is replaced: ``num_classes`` outputs (+ 1 background class under ``activation_fn="softmax"``), bias at the focal-loss prior
0.01, one head shared by the 6 decoder layers (plain variant) or 6 independent clones (box-refinement variant) — exactly the
state-dict layout the reference's fine-tuning checkpoints carry.  ``weights`` (a ``.pth`` / ``.ckpt`` of the re-headed model)
is loaded last.  ``base_weights=None`` skips the base checkpoint (this offline build cannot download it; see
``alonet.common.load_weights`` for where a local copy is looked up).
"""
import math

import torch
from torch import nn

from alonet.common import load_weights

from .deformable_detr import _get_clones
from .deformable_detr_r50 import DeformableDetrR50, DeformableDetrR50Refinement


def _rehead(model, num_classes, activation_fn, weights, clone):
    if activation_fn not in ("sigmoid", "softmax"):
        raise Exception(f"activation_fn = {activation_fn} must be one of this two values: 'sigmoid' or 'softmax'.")
    model.activation_fn = activation_fn
    model.background_class = num_classes if activation_fn == "softmax" else None
    num_classes += 1 if activation_fn == "softmax" else 0   # background class
    head = nn.Linear(model.transformer.d_model, num_classes)
    prior_prob = 0.01
    head.bias.data = torch.ones(num_classes) * -math.log((1 - prior_prob) / prior_prob)
    head = head.to(model.device)
    num_pred = model.transformer.decoder.num_layers
    model.class_embed = _get_clones(head, num_pred) if clone else nn.ModuleList([head for _ in range(num_pred)])
    import alo_hip

    alo_hip.invalidate_caches(model)   # packed / merged inference-time copies of the old head
    if weights is not None:
        if ".pth" in weights or ".ckpt" in weights:
            load_weights(model, weights, model.device)
        else:
            raise ValueError(f"Unknown weights: '{weights}'")


class DeformableDetrR50Finetune(DeformableDetrR50):
    """``DeformableDetrR50Finetune(num_classes, activation_fn="sigmoid", base_weights="deformable-detr-r50", weights=None, **kw)``."""

    def __init__(self, num_classes, activation_fn="sigmoid", base_weights="deformable-detr-r50", weights=None, **kwargs):
        if activation_fn not in ("sigmoid", "softmax"):
            raise Exception(f"activation_fn = {activation_fn} must be one of this two values: 'sigmoid' or 'softmax'.")
        super().__init__(weights=base_weights, **kwargs)
        _rehead(self, num_classes, activation_fn, weights, clone=False)


class DeformableDetrR50RefinementFinetune(DeformableDetrR50Refinement):
    """The same for the iterative-refinement model (``base_weights="deformable-detr-r50-refinement"``; independent heads)."""

    def __init__(self, num_classes, activation_fn="sigmoid", base_weights="deformable-detr-r50-refinement", weights=None, **kwargs):
        if activation_fn not in ("sigmoid", "softmax"):
            raise Exception(f"activation_fn = {activation_fn} must be one of this two values: 'sigmoid' or 'softmax'.")
        super().__init__(weights=base_weights, **kwargs)
        _rehead(self, num_classes, activation_fn, weights, clone=True)
