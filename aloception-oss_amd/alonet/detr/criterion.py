"""DETR set-prediction loss (reference: alonet/detr/criterion.py:14-446): Hungarian matching, then classification
(cross entropy with a down-weighted no-object class), L1 and GIoU box losses on the matched pairs, repeated for every
auxiliary decoder output.  ``num_boxes`` is summed over ranks with one ``all_reduce`` — the only explicit collective
of the reference on this path (:411-413)."""
import torch
import torch.nn.functional as F
from torch import nn

import aloscene
from alonet.multi_gpu import get_world_size, is_dist_avail_and_initialized


class DetrCriterion(nn.Module):
    def __init__(self, matcher, loss_ce_weight, loss_boxes_weight, loss_giou_weight, eos_coef, aux_loss_stage, losses):
        super().__init__()
        self.matcher, self.eos_coef, self.losses = matcher, eos_coef, losses
        weights = {"loss_ce": loss_ce_weight, "loss_bbox": loss_boxes_weight, "loss_giou": loss_giou_weight}
        if aux_loss_stage > 0:
            for i in range(aux_loss_stage - 1):
                weights.update({f"{k}_{i}": v for k, v in list(weights.items()) if k in ("loss_ce", "loss_bbox", "loss_giou")})
        self.loss_weights = weights

    @staticmethod
    def _num_classes(frames):
        return len(frames.boxes2d[0].labels.labels_names)

    def _get_src_permutation_idx(self, indices, **kwargs):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for src, _ in indices])

    def _target_classes(self, logits, frames, indices, background):
        matched = torch.cat([b.labels.as_tensor()[indices[i][1]] for i, b in enumerate(frames.boxes2d)]).long()
        target = torch.full(logits.shape[:2], background, dtype=torch.int64, device=logits.device)
        target[self._get_src_permutation_idx(indices)] = matched.to(logits.device)
        return target

    def loss_labels(self, outputs, frames, indices, num_boxes, **kwargs):
        logits = outputs["pred_logits"]
        num_classes = self._num_classes(frames)
        target = self._target_classes(logits, frames, indices, num_classes)
        weight = torch.ones(logits.shape[-1], device=logits.device, dtype=logits.dtype)
        weight[num_classes] = self.eos_coef
        return {"loss_ce": F.cross_entropy(logits.transpose(1, 2), target, weight)}

    def loss_boxes(self, outputs, frames, indices, num_boxes, **kwargs):
        if num_boxes == 0:
            return {}
        pred = outputs["pred_boxes"][self._get_src_permutation_idx(indices)].float()
        tgt = torch.cat([b.xcyc().rel_pos().as_tensor()[indices[i][1]] for i, b in enumerate(frames.boxes2d)], 0)
        tgt = tgt.to(pred.device).float()
        losses = {"loss_bbox": F.l1_loss(pred, tgt, reduction="none").sum() / num_boxes}
        giou = aloscene.BoundingBoxes2D(pred, "xcyc", False).giou_with(aloscene.BoundingBoxes2D(tgt, "xcyc", False))
        losses["loss_giou"] = (1 - torch.diag(giou)).sum() / num_boxes
        return losses

    def get_loss(self, loss, outputs, frames, indices, num_boxes, update_loss_map=None, **kwargs):
        loss_map = {"labels": self.loss_labels, "boxes": self.loss_boxes}
        if update_loss_map is not None:
            loss_map.update(update_loss_map)
        assert loss in loss_map, f"do you really want to compute {loss} loss?"
        return loss_map[loss](outputs, frames, indices, num_boxes, **kwargs)

    @torch.no_grad()
    def get_metrics(self, outputs, frames, indices, num_boxes, **kwargs):
        """Slot-level monitoring metrics of the matched predictions, with the reference's definitions and quirks
        (criterion.py:214-301): a ratio is reported when numerator and denominator are both non-zero (0 when only the
        numerator is), ``recall`` / ``precision`` are means taken in float16."""
        if num_boxes == 0:
            return {}
        background = self._num_classes(frames)
        pred = outputs["pred_logits"].argmax(-1)
        target = self._target_classes(outputs["pred_logits"], frames, indices, background)

        def ratio(metrics, key, num, den):
            if num != 0 and den != 0:
                metrics[key] = num / den
            elif num > 0:
                metrics[key] = 0

        metrics = {}
        is_obj, pred_pos = target != background, pred != background
        ratio(metrics, "objectness_recall", int((is_obj & pred_pos).sum()), int(is_obj.sum()))
        metrics["recall"] = (pred[is_obj] == target[is_obj]).to(torch.float16).mean()
        ratio(metrics, "objectness_true_pos", int((pred_pos & is_obj).sum()), int(pred_pos.sum()))
        if pred_pos.any():
            metrics["precision"] = (target[pred_pos] == pred[pred_pos]).to(torch.float16).mean()
        pred_neg = ~pred_pos
        ratio(metrics, "true_neg", int((pred_neg & ~is_obj).sum()), int(pred_neg.sum()))
        ratio(metrics, "slot_true_neg", int((~is_obj & pred_neg).sum()), int((~is_obj).sum()))
        return metrics

    def forward(self, m_outputs, frames, matcher_frames=None, compute_statistical_metrics=False, **kwargs):
        assert isinstance(frames, aloscene.Frame) and isinstance(frames.boxes2d[0], aloscene.BoundingBoxes2D)
        matcher_frames = matcher_frames if matcher_frames is not None else frames
        main = {k: v for k, v in m_outputs.items() if k != "aux_outputs"}
        indices = self.matcher(main, matcher_frames, **kwargs)

        num_boxes = sum(b.shape[0] for b in frames.boxes2d)
        num_boxes = torch.as_tensor([num_boxes], dtype=torch.float, device=m_outputs["pred_logits"].device)
        if is_dist_avail_and_initialized():
            torch.distributed.all_reduce(num_boxes)  # 4 bytes: the path's one explicit collective
        num_boxes = torch.clamp(num_boxes / get_world_size(), min=1).item()

        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, m_outputs, frames, indices, num_boxes))
        metrics = self.get_metrics(m_outputs, frames, indices, num_boxes)
        for i, aux in enumerate(m_outputs.get("aux_outputs", [])):
            aux_indices = self.matcher(aux, matcher_frames, **kwargs)
            for loss in self.losses:
                if loss == "masks":
                    continue
                l_dict = self.get_loss(loss, aux, frames, aux_indices, num_boxes, **kwargs)
                losses.update({f"{k}_{i}": v for k, v in l_dict.items()})
        total = sum(losses[k] * self.loss_weights[k] for k in losses)
        losses.update(metrics)
        return total, losses
