#!/usr/bin/env python
"""G13 — the reference's own set-prediction losses and Hungarian matchers on a seeded batch (build container only).

Unlike the hot-path modules of make_golden.py, the criterion needs the reference's REAL ``aloscene`` (``Frame`` children,
``BoundingBoxes2D.giou_with`` / ``rel_pos`` / ``remove_padding``, ``Labels``).  That package imports torchvision, cv2,
matplotlib, pytorch_lightning ... at module level; none of them is used by the code exercised here, so they are replaced
by inert module shells (any attribute resolves to a dummy class).  ``alonet`` is imported piecewise under package shells as
in make_golden.py.  Run in its own interpreter (make_golden.py spawns it): its ``aloscene`` must not meet the 5-line shim the
other generators install.

Writes tests/golden/g13_criterion.npz: inputs (logits, boxes, targets) and the reference's outputs — matched indices per
decoder level, every loss term of DetrCriterion (softmax / cross-entropy) and DeformableCriterion (sigmoid / focal), totals.
"""
import importlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _dummy(name)


def _dummy(name):
    return _Meta(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return sys.modules.get(self.__name__ + "." + name) or _dummy(name)


def _inert(name):
    mod = _Inert(name)
    mod.__path__ = []
    sys.modules[name] = mod


def load():
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.ops",
                 "torchvision.ops.boxes", "torchvision.io", "torchvision.io.image", "torchvision.utils", "torchvision.models",
                 "torchvision.models._utils", "cv2", "matplotlib", "matplotlib.pyplot", "matplotlib.cm", "pytorch_lightning",
                 "pytorch_lightning.callbacks", "pytorch_lightning.loggers", "wandb", "more_itertools", "pycocotools",
                 "pycocotools.coco", "pycocotools.mask", "PIL", "PIL.Image", "imageio", "open3d"):
        _inert(name)
    import pkg_resources

    real = pkg_resources.get_distribution
    pkg_resources.get_distribution = lambda n: types.SimpleNamespace(version="0.6.0beta") if n == "aloception" else real(n)
    sys.path.insert(0, REF)
    import aloscene  # the reference's own package

    def shell(name, path):
        mod = types.ModuleType(name)
        mod.__path__ = [path]
        sys.modules[name] = mod
        return mod

    shell("alonet", REF + "/alonet")
    shell("alonet.detr", REF + "/alonet/detr")
    shell("alonet.deformable_detr", REF + "/alonet/deformable_detr")
    importlib.import_module("alonet.multi_gpu")
    crit = importlib.import_module("alonet.detr.criterion")
    match = importlib.import_module("alonet.detr.matcher")
    sys.modules["alonet.detr"].DetrCriterion = crit.DetrCriterion
    dcrit = importlib.import_module("alonet.deformable_detr.criterion")
    dmatch = importlib.import_module("alonet.deformable_detr.matcher")
    return aloscene, crit, match, dcrit, dmatch


def main():
    import torch

    aloscene, crit, match, dcrit, dmatch = load()
    torch.manual_seed(1313)
    num_classes, B, Q, stages = 5, 3, 12, 3   # decoder levels: 2 auxiliary + the last one
    names = [f"c{i}" for i in range(num_classes)]
    counts = [3, 0, 2]                         # image 1 has no ground truth
    frames, tgt_boxes, tgt_labels = [], [], []
    for n in counts:
        cxcy, wh = torch.rand(n, 2) * 0.6 + 0.2, torch.rand(n, 2) * 0.3 + 0.05
        box = torch.cat([cxcy, wh], 1)
        lab = torch.randint(0, num_classes, (n,)).float()
        labels = aloscene.Labels(lab, encoding="id", labels_names=names, names=("N",))
        boxes = aloscene.BoundingBoxes2D(box, boxes_format="xcyc", absolute=False, names=("N", None), labels=labels)
        frames.append(aloscene.Frame(torch.zeros(3, 16, 24), normalization="resnet", names=("C", "H", "W"), boxes2d=boxes))
        tgt_boxes.append(box.numpy())
        tgt_labels.append(lab.numpy())
    frames = aloscene.Frame.batch_list(frames)

    def outputs(n_logits, activation):
        def one():
            cxcy, wh = torch.rand(B, Q, 2) * 0.8 + 0.1, torch.rand(B, Q, 2) * 0.4 + 0.02
            return {"pred_logits": torch.randn(B, Q, n_logits) * 1.5, "pred_boxes": torch.cat([cxcy, wh], -1),
                    "activation_fn": activation}
        out = one()
        out["aux_outputs"] = [one() for _ in range(stages - 1)]
        return out

    save = dict(counts=np.array(counts), num_classes=np.array(num_classes))
    for i, (b, l) in enumerate(zip(tgt_boxes, tgt_labels)):
        save[f"tgt_boxes{i}"], save[f"tgt_labels{i}"] = b, l

    def run(tag, out, criterion):
        total, parts = criterion(out, frames)
        levels = [out] + out["aux_outputs"]
        for s, lvl in enumerate(levels):
            save[f"{tag}.logits{s}"], save[f"{tag}.boxes{s}"] = lvl["pred_logits"].numpy(), lvl["pred_boxes"].numpy()
            for bi, (pi, ti) in enumerate(criterion.matcher({k: v for k, v in lvl.items() if k != "aux_outputs"}, frames)):
                save[f"{tag}.match{s}.{bi}"] = np.stack([pi.numpy(), ti.numpy()])
        save[f"{tag}.total"] = np.array(float(total))
        for k, v in parts.items():
            save[f"{tag}.part.{k}"] = np.array(float(v))
        print(tag, float(total), {k: round(float(v), 5) for k, v in parts.items()})

    m1 = match.DetrHungarianMatcher(cost_class=1, cost_boxes=5, cost_giou=2)
    c1 = crit.DetrCriterion(matcher=m1, loss_ce_weight=1, loss_boxes_weight=5, loss_giou_weight=2, eos_coef=0.1,
                            aux_loss_stage=stages, losses=["labels", "boxes"])
    run("detr", outputs(num_classes + 1, "softmax"), c1)
    m2 = dmatch.DeformableDetrHungarianMatcher(cost_class=1, cost_boxes=5, cost_giou=2)
    c2 = dcrit.DeformableCriterion(matcher=m2, loss_label_weight=1, loss_boxes_weight=5, loss_giou_weight=2, eos_coef=0.1,
                                   aux_loss_stage=stages, losses=["labels", "boxes"], focal_alpha=0.25)
    run("deformable", outputs(num_classes, "sigmoid"), c2)
    np.savez_compressed(os.path.join(OUT, "g13_criterion.npz"), **save)
    print("wrote g13_criterion.npz")


if __name__ == "__main__":
    main()
