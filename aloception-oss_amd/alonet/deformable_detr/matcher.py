"""Matcher of Deformable-DETR: focal-style class cost with sigmoid outputs (reference: deformable_detr/matcher.py:8-42)."""
import torch

from alonet.detr.matcher import DetrHungarianMatcher


class DeformableDetrHungarianMatcher(DetrHungarianMatcher):
    @torch.no_grad()
    def hungarian_cost_class(self, tgt_boxes, m_outputs, **kwargs):
        if "activation_fn" not in m_outputs:
            raise Exception("'activation_fn' must be declared in forward output.")
        ids = tgt_boxes.labels.as_tensor().long()
        logits = m_outputs["pred_logits"].flatten(0, 1).float()
        if m_outputs["activation_fn"] == "softmax":
            return -logits.softmax(-1)[:, ids]
        prob = logits.sigmoid()
        alpha, gamma = 0.25, 2.0
        neg = (1 - alpha) * (prob ** gamma) * (-(1 - prob + 1e-8).log())
        pos = alpha * ((1 - prob) ** gamma) * (-(prob + 1e-8).log())
        return pos[:, ids] - neg[:, ids]
