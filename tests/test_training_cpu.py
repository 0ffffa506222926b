"""Training-side pieces (matcher, criteria, optimizer groups) checked against closed-form answers.  CPU only.

The reference's own criterion cannot be imported here (it needs the full aloscene / torchvision stack), so these are
known-answer tests of the published formulas rather than golden comparisons.
"""
import math

import numpy as np
import torch

import aloscene
from alonet.deformable_detr.criterion import sigmoid_focal_loss
from alonet.deformable_detr.matcher import DeformableDetrHungarianMatcher
from alonet.deformable_detr.training import build_criterion, configure_optimizers
from alonet.detr.criterion import DetrCriterion
from alonet.detr.matcher import DetrHungarianMatcher

NAMES = [f"c{i}" for i in range(5)]


def boxes(xcyc, labels):
    lab = aloscene.Labels(torch.tensor(labels, dtype=torch.float32), encoding="id", labels_names=NAMES)
    return aloscene.BoundingBoxes2D(torch.tensor(xcyc, dtype=torch.float32).view(-1, 4), "xcyc", False, labels=lab)


def frames_with(targets):
    fs = [aloscene.Frame(torch.zeros(3, 8, 8), normalization="resnet", boxes2d=b) for b in targets]
    return aloscene.Frame.batch_list(fs)


def test_giou_known_values():
    a = aloscene.BoundingBoxes2D(torch.tensor([[0.0, 0.0, 2.0, 2.0]]), "xyxy", False)
    same = a.giou_with(a)
    assert same.shape == (1, 1) and abs(same.item() - 1.0) < 1e-6
    b = aloscene.BoundingBoxes2D(torch.tensor([[1.0, 1.0, 3.0, 3.0], [3.0, 3.0, 4.0, 4.0]]), "xyxy", False)
    g = a.giou_with(b)
    # overlap 1, union 7, hull 9 -> 1/7 - 2/9 ; disjoint: iou 0, union 5, hull 16 -> -11/16
    np.testing.assert_allclose(g.numpy(), [[1 / 7 - 2 / 9, -11 / 16]], atol=1e-6)
    c = aloscene.BoundingBoxes2D(torch.tensor([[1.0, 1.0, 2.0, 2.0]]), "xcyc", False)  # == a in xcyc
    assert abs(c.giou_with(a).item() - 1.0) < 1e-6


def test_frame_batch_collects_boxes_per_image():
    fr = frames_with([boxes([[0.5, 0.5, 0.2, 0.2]], [1]), boxes([[0.3, 0.3, 0.1, 0.1], [0.7, 0.6, 0.2, 0.4]], [0, 4])])
    assert isinstance(fr.boxes2d, list) and [b.shape[0] for b in fr.boxes2d] == [1, 2]
    assert fr.boxes2d[1].labels.labels_names == NAMES


def test_hungarian_matcher_finds_the_obvious_assignment():
    tgt = [boxes([[0.2, 0.2, 0.1, 0.1], [0.8, 0.8, 0.2, 0.2]], [1, 3]), boxes([[0.5, 0.5, 0.4, 0.4]], [2])]
    fr = frames_with(tgt)
    pred_boxes = torch.rand(2, 6, 4) * 0.1 + 0.45
    pred_boxes[0, 4] = torch.tensor([0.2, 0.2, 0.1, 0.1])  # slot 4 <- target 0
    pred_boxes[0, 1] = torch.tensor([0.8, 0.8, 0.2, 0.2])  # slot 1 <- target 1
    pred_boxes[1, 3] = torch.tensor([0.5, 0.5, 0.4, 0.4])
    logits = torch.zeros(2, 6, 6)
    out = {"pred_logits": logits, "pred_boxes": pred_boxes, "activation_fn": "sigmoid"}
    for matcher in (DetrHungarianMatcher(1, 5, 2), DeformableDetrHungarianMatcher(1, 5, 2)):
        idx = matcher(out, fr)
        assert sorted(zip(idx[0][0].tolist(), idx[0][1].tolist())) == [(1, 1), (4, 0)]
        assert (idx[1][0].tolist(), idx[1][1].tolist()) == ([3], [0])
    empty = frames_with([boxes([], []), boxes([], [])])
    assert all(len(i) == 0 and len(j) == 0 for i, j in DetrHungarianMatcher()(out, empty))


def test_sigmoid_focal_loss_closed_form():
    x = torch.tensor([[2.0, -1.0], [0.0, 3.0]])
    t = torch.tensor([[1.0, 0.0], [0.0, 1.0]])
    p = torch.sigmoid(x)
    expect = []
    for ti, pi in zip(t.flatten(), p.flatten()):
        pt = pi if ti == 1 else 1 - pi
        at = 0.25 if ti == 1 else 0.75
        expect.append(at * (1 - pt) ** 2 * -math.log(pt))
    assert abs(sigmoid_focal_loss(x, t, 1).item() - float(np.mean(expect))) < 1e-6


def test_criteria_on_perfect_and_imperfect_predictions():
    tgt = [boxes([[0.3, 0.3, 0.2, 0.2]], [2]), boxes([[0.6, 0.5, 0.3, 0.4]], [4])]
    fr = frames_with(tgt)
    pred_boxes = torch.full((2, 4, 4), 0.5)
    pred_boxes[0, 1] = torch.tensor([0.3, 0.3, 0.2, 0.2])
    pred_boxes[1, 2] = torch.tensor([0.6, 0.5, 0.3, 0.4])
    logits = torch.full((2, 4, 5), -12.0)
    logits[0, 1, 2] = 12.0
    logits[1, 2, 4] = 12.0
    crit = build_criterion(aux_loss_stage=1)
    total, parts = crit({"pred_logits": logits.clone().requires_grad_(True),
                         "pred_boxes": pred_boxes.clone().requires_grad_(True), "activation_fn": "sigmoid"}, fr)
    assert parts["loss_bbox"].item() < 1e-6 and parts["loss_giou"].item() < 1e-6 and parts["loss_focal_label"].item() < 1e-4
    assert float(parts["recall"]) == 1.0
    shifted = pred_boxes.clone()
    shifted[0, 1, 0] += 0.1  # one matched box off by 0.1 in x: L1 = 0.1 / num_boxes(2)
    total2, parts2 = crit({"pred_logits": logits, "pred_boxes": shifted, "activation_fn": "sigmoid"}, fr)
    assert abs(parts2["loss_bbox"].item() - 0.05) < 1e-6 and parts2["loss_giou"].item() > 0 and total2 > total
    # softmax variant goes through the cross-entropy of the DETR criterion
    sm_logits = torch.full((2, 4, 6), -12.0)
    sm_logits[..., 5] = 12.0
    sm_logits[0, 1] = torch.tensor([-12.0, -12, 12, -12, -12, -12])
    sm_logits[1, 2] = torch.tensor([-12.0, -12, -12, -12, 12, -12])
    _, parts3 = crit({"pred_logits": sm_logits, "pred_boxes": pred_boxes, "activation_fn": "softmax"}, fr)
    assert parts3["loss_ce"].item() < 1e-6
    det = DetrCriterion(DetrHungarianMatcher(1, 5, 2), 1, 5, 2, eos_coef=0.1, aux_loss_stage=3, losses=["labels", "boxes"])
    assert {"loss_ce", "loss_bbox_0", "loss_giou_1"} <= set(det.loss_weights)


def test_optimizer_param_groups_follow_the_reference():
    from alonet.deformable_detr import DeformableDetrR50

    model = DeformableDetrR50(device=None, aux_loss=True)
    opt = configure_optimizers(model)
    g = opt.param_groups
    assert [x["lr"] for x in g] == [1e-5, 1e-5, 1e-4] and all(x["weight_decay"] == 1e-4 for x in g)
    n_proj = sum(p.numel() for n, p in model.named_parameters() if ("sampling_offsets" in n or "reference_points" in n))
    assert sum(p.numel() for p in g[0]["params"]) == n_proj
    n_bb = sum(p.numel() for n, p in model.named_parameters() if "backbone" in n and p.requires_grad)
    assert sum(p.numel() for p in g[1]["params"]) == n_bb
    total = sum(p.numel() for x in g for p in x["params"])
    assert abs(total - 39.85e6) < 0.05e6  # the 159 MB of fp32 gradients DDP all-reduces per step
