"""Loss of Deformable-DETR: the DETR criterion with a sigmoid focal classification term
(reference: alonet/deformable_detr/criterion.py:10-154)."""
import torch
import torch.nn.functional as F

from alonet.detr.criterion import DetrCriterion


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha=0.25, gamma=2):
    """Mean focal loss over all (query, class) logits; ``num_boxes`` is accepted for API parity and unused, as in the
    reference."""
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean()


class DeformableCriterion(DetrCriterion):
    def __init__(self, loss_label_weight, focal_alpha=0.25, **kwargs):
        if "loss_ce_weight" in kwargs or "loss_focal_label" in kwargs:
            raise Exception("the weight of the label loss is given by 'loss_label_weight'")
        kwargs["loss_ce_weight"] = loss_label_weight
        super().__init__(**kwargs)
        self.focal_alpha = focal_alpha
        self.loss_weights["loss_focal_label"] = loss_label_weight
        for i in range(max(kwargs["aux_loss_stage"] - 1, 0)):
            self.loss_weights[f"loss_focal_label_{i}"] = loss_label_weight

    def loss_labels(self, outputs, frames, indices, num_boxes, **kwargs):
        if "activation_fn" not in outputs:
            raise Exception("'activation_fn' must be declared in forward output.")
        if outputs["activation_fn"] == "softmax":
            return super().loss_labels(outputs, frames, indices, num_boxes, **kwargs)
        logits = outputs["pred_logits"].float()
        target = self._target_classes(logits, frames, indices, self._num_classes(frames))
        onehot = torch.zeros(logits.shape[0], logits.shape[1], logits.shape[2] + 1, dtype=logits.dtype, device=logits.device)
        onehot.scatter_(2, target.unsqueeze(-1), 1)
        loss = sigmoid_focal_loss(logits, onehot[:, :, :-1], num_boxes, alpha=self.focal_alpha, gamma=2) * logits.shape[1]
        return {"loss_focal_label": loss}

    @torch.no_grad()
    def get_metrics(self, outputs, frames, indices, num_boxes, **kwargs):
        """Precision / recall of the matched predictions at the reference's fixed score threshold 0.3
        (deformable_detr/criterion.py:156-229)."""
        if "activation_fn" not in outputs:
            raise Exception("'activation_fn' must be declared in forward output.")
        if outputs["activation_fn"] == "softmax":
            return super().get_metrics(outputs, frames, indices, num_boxes, **kwargs)
        if num_boxes == 0:
            return {}
        background = self._num_classes(frames)
        scores, pred = outputs["pred_logits"].sigmoid().max(-1)
        target = self._target_classes(outputs["pred_logits"], frames, indices, background)
        confident = scores >= 0.3
        true_pos = (pred == target)[confident].sum()
        n_pos, n_gt = confident.sum(), int((target != background).sum())
        zero = torch.zeros((), device=pred.device)
        return {"precision": true_pos / n_pos if n_pos > 0 else zero, "recall": true_pos / n_gt if n_gt > 0 else zero}
