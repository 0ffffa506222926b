"""Hungarian matching between predictions and ground-truth boxes (reference: alonet/detr/matcher.py:8-185).

The cost matrix (class + L1 + GIoU terms) is built on the device in one shot for the whole batch; the assignment
itself is scipy's ``linear_sum_assignment`` on the host, one image at a time, exactly as the reference does.
"""
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

import aloscene


class DetrHungarianMatcher(nn.Module):
    def __init__(self, cost_class=1, cost_boxes=1, cost_giou=1):
        super().__init__()
        assert cost_class != 0 or cost_boxes != 0 or cost_giou != 0, "all costs cant be 0"
        self.cost_class, self.cost_boxes, self.cost_giou = cost_class, cost_boxes, cost_giou

    @torch.no_grad()
    def hungarian_cost_class(self, tgt_boxes, m_outputs, **kwargs):
        prob = m_outputs["pred_logits"].flatten(0, 1).softmax(-1)
        return -prob[:, tgt_boxes.labels.as_tensor().long()]  # 1 - p[target] up to a constant

    @torch.no_grad()
    def hungarian_cost_l1_boxes(self, tgt_boxes, m_outputs, **kwargs):
        assert tgt_boxes.boxes_format == "xcyc" and not tgt_boxes.absolute
        return torch.cdist(m_outputs["pred_boxes"].flatten(0, 1).float(), tgt_boxes.as_tensor().float(), p=1)

    @torch.no_grad()
    def hungarian_cost_giou_boxes(self, tgt_boxes, m_outputs, **kwargs):
        pred = aloscene.BoundingBoxes2D(m_outputs["pred_boxes"].flatten(0, 1).float(), "xcyc", False, names=("N", None))
        return -pred.giou_with(tgt_boxes)

    def hungarian(self, batch_cost_matrix, **kwargs):
        out = []
        for c in batch_cost_matrix:
            rows, cols = linear_sum_assignment(c)
            out.append((torch.as_tensor(rows, dtype=torch.int64), torch.as_tensor(cols, dtype=torch.int64)))
        return out

    @torch.no_grad()
    def forward(self, m_outputs, frames, **kwargs):
        """-> list (one per image) of (prediction indices, target indices), len = min(num_queries, num_targets)."""
        assert isinstance(frames, aloscene.Frame) and isinstance(frames.boxes2d[0], aloscene.BoundingBoxes2D)
        assert frames.boxes2d[0].labels is not None and frames.boxes2d[0].labels.encoding == "id"
        bs, num_queries = m_outputs["pred_logits"].shape[:2]
        per_image = [b.rel_pos().xcyc().remove_padding() for b in frames.boxes2d]
        sizes = [b.shape[0] for b in per_image]
        if sum(sizes) == 0:
            empty = torch.as_tensor([], dtype=torch.int64)
            return [(empty, empty) for _ in range(bs)]
        device = m_outputs["pred_logits"].device
        labels = aloscene.Labels(torch.cat([b.labels.as_tensor() for b in per_image]).to(device), encoding="id")
        tgt = aloscene.BoundingBoxes2D(torch.cat([b.as_tensor() for b in per_image]).to(device), "xcyc", False,
                                       labels=labels)
        cost = (self.cost_boxes * self.hungarian_cost_l1_boxes(tgt, m_outputs, **kwargs)
                + self.cost_class * self.hungarian_cost_class(tgt, m_outputs, **kwargs)
                + self.cost_giou * self.hungarian_cost_giou_boxes(tgt, m_outputs, **kwargs))
        cost = cost.view(bs, num_queries, -1).float().cpu()  # device -> host: the assignment runs on the CPU
        return self.hungarian([c[i] for i, c in enumerate(cost.split(sizes, -1))], **kwargs)
