"""Training step of Deformable-DETR without the Lightning harness: criterion / matcher / optimizer as the reference's
``LitDeformableDetr`` builds them (alonet/deformable_detr/train.py:88-175), plus the DDP wrapping a multi-GPU job uses.

Gradients of the 12 deformable-attention calls run through ``alo_msda_backward``; the gradient all-reduce is stock
``DistributedDataParallel`` over RCCL (backend ``"nccl"``), bucketed and overlapped with backward; the matcher's
assignment step stays on the host CPU, as in the reference.
"""
import torch

from .criterion import DeformableCriterion
from .matcher import DeformableDetrHungarianMatcher


def build_matcher(cost_class=1, cost_boxes=5, cost_giou=2):
    return DeformableDetrHungarianMatcher(cost_class=cost_class, cost_boxes=cost_boxes, cost_giou=cost_giou)


def build_criterion(matcher=None, loss_label_weight=1, loss_boxes_weight=5, loss_giou_weight=2,
                    losses=("labels", "boxes"), aux_loss_stage=6, eos_coef=0.1):
    return DeformableCriterion(matcher=matcher or build_matcher(), loss_label_weight=loss_label_weight,
                               loss_boxes_weight=loss_boxes_weight, loss_giou_weight=loss_giou_weight,
                               losses=list(losses), aux_loss_stage=aux_loss_stage, eos_coef=eos_coef)


def configure_optimizers(model, lr=1e-4, lr_backbone=1e-5, lr_linear_proj=1e-5, weight_decay=1e-4):
    """AdamW with the reference's three groups: sampling_offsets / reference_points projections, backbone, the rest."""
    proj_keys = ("reference_points", "sampling_offsets")
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [
        {"params": [p for n, p in named if any(k in n for k in proj_keys)], "lr": lr_linear_proj},
        {"params": [p for n, p in named if "backbone" in n], "lr": lr_backbone},
        {"params": [p for n, p in named if "backbone" not in n and not any(k in n for k in proj_keys)]},
    ]
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay)


def wrap_ddp(model, device_index):
    """One replica per process; stock DDP = bucketed all-reduce of the 159 MB of fp32 gradients over RCCL / xGMI."""
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=[device_index], find_unused_parameters=False)


def training_step(model, criterion, optimizer, frames, clip_grad=0.1):
    """forward -> set loss -> backward -> clip (reference default gradient_clip_val 0.1) -> AdamW step."""
    optimizer.zero_grad(set_to_none=True)
    outputs = model(frames)
    total, parts = criterion(outputs, frames)
    total.backward()
    if clip_grad:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad)
    optimizer.step()
    return total.detach(), parts
