"""Training-side pieces (matcher, criteria, optimizer groups) checked against closed-form answers.  CPU only.

Known-answer tests of the published formulas, plus G13: the outputs of the reference's OWN criterion / matcher classes on a
seeded batch (generated in the build container with the reference's real aloscene under inert torchvision / cv2 shells,
tests/golden/make_golden_criterion.py).
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


# ---- G13: the reference's own criterion / matcher outputs (tests/golden/make_golden_criterion.py) -----------------------------
def _g13_frames(g):
    names = [f"c{i}" for i in range(int(g["num_classes"]))]
    fs = []
    for i, n in enumerate(g["counts"]):
        lab = aloscene.Labels(torch.from_numpy(g[f"tgt_labels{i}"]).float(), encoding="id", labels_names=names)
        bx = aloscene.BoundingBoxes2D(torch.from_numpy(g[f"tgt_boxes{i}"]).float().view(-1, 4), "xcyc", False, labels=lab)
        fs.append(aloscene.Frame(torch.zeros(3, 16, 24), normalization="resnet", boxes2d=bx))
    return aloscene.Frame.batch_list(fs)


def _g13_outputs(g, tag, activation):
    levels = []
    s = 0
    while f"{tag}.logits{s}" in g.files:
        levels.append({"pred_logits": torch.from_numpy(g[f"{tag}.logits{s}"]), "pred_boxes": torch.from_numpy(g[f"{tag}.boxes{s}"]),
                       "activation_fn": activation})
        s += 1
    out = dict(levels[0])
    out["aux_outputs"] = levels[1:]
    return out, levels


import pytest  # noqa: E402


@pytest.mark.parametrize("tag", ["detr", "deformable"])
def test_criterion_and_matcher_match_the_reference(golden, tag):
    """Matched indices of every decoder level, every loss term, the weighted total and the monitoring metrics against what
    the reference's own DetrCriterion / DeformableCriterion (+ Hungarian matchers) return on the same batch — one image
    without ground truth included."""
    from alonet.deformable_detr.criterion import DeformableCriterion

    g = golden("g13_criterion.npz")
    frames = _g13_frames(g)
    if tag == "detr":
        out, levels = _g13_outputs(g, tag, "softmax")
        crit = DetrCriterion(matcher=DetrHungarianMatcher(1, 5, 2), loss_ce_weight=1, loss_boxes_weight=5, loss_giou_weight=2,
                             eos_coef=0.1, aux_loss_stage=len(levels), losses=["labels", "boxes"])
    else:
        out, levels = _g13_outputs(g, tag, "sigmoid")
        crit = DeformableCriterion(matcher=DeformableDetrHungarianMatcher(1, 5, 2), loss_label_weight=1, loss_boxes_weight=5,
                                   loss_giou_weight=2, eos_coef=0.1, aux_loss_stage=len(levels), losses=["labels", "boxes"],
                                   focal_alpha=0.25)
    for s, lvl in enumerate(levels):
        idx = crit.matcher(lvl, frames)
        for bi, (pi, ti) in enumerate(idx):
            want = g[f"{tag}.match{s}.{bi}"]
            assert pi.tolist() == want[0].tolist() and ti.tolist() == want[1].tolist(), (s, bi)
    total, parts = crit(out, frames)
    assert abs(float(total) - float(g[f"{tag}.total"])) <= 1e-5 * abs(float(g[f"{tag}.total"]))
    ref_parts = {k[len(tag) + 6:]: float(g[k]) for k in g.files if k.startswith(f"{tag}.part.")}
    assert set(ref_parts) == set(parts), (sorted(ref_parts), sorted(parts))
    for k, v in ref_parts.items():
        assert abs(float(parts[k]) - v) <= 1e-5 * max(1.0, abs(v)), (k, float(parts[k]), v)
