#!/usr/bin/env python
"""G14 / G15 / G16 — the reference's OWN model classes and I/O objects above the transformer (build container only).

What make_golden.py pins stops at the transformers (G5 / G10 / G12) and at RAFT (G7).  This generator runs the reference's
``DeformableDETR`` (forward, heads, ``inference()``), ``Detr``, ``PanopticHead`` (forward + ``inference()``) and the real
``aloscene.Frame`` (``norm_*``, ``batch_list``) and records inputs + outputs:

  g14_deformable_detr.npz        alonet/deformable_detr/deformable_detr.py:215-299 (forward: input_proj, stride-64 level, mask
                                 resize, positional encodings), :301-410 (heads, inverse_sigmoid refinement), :419-555
                                 (get_outs_labels / get_outs_filter / inference) — three configurations on one padded two-frame
                                 batch, fp64: ``plain`` (sigmoid, aux outputs, dec/enc/bb outputs), ``refine`` (with_box_refine),
                                 ``softmax`` (background class); inference at two thresholds each
  g14b_deformable_detr_d256.npz  same, at the DETR-family width (d_model 256, 8 heads, 4 levels, 4 points, 1 + 1 layers): the
                                 shape this repository's bf16 inference fast path is specialised for; stored as fp32
  g15_detr_panoptic.npz          alonet/detr/detr.py:126-240,315 (Detr forward / inference) and
                                 alonet/detr_panoptic/detr_panoptic.py:111-311 (PanopticHead forward / inference) over ``Detr`` AND
                                 over ``DeformableDETR`` (configs[4])
  g16_frame_io.npz               aloscene/frame.py:386-548 (norm01 / norm255 / norm_minmax_sym / norm_resnet from every state),
                                 aloscene/tensors/spatial_augmented_tensor.py:323-419 (batch_list: padded values + mask)

How the reference is made to run here (harness only; no reference source is edited or copied):
  * the reference's real ``aloscene`` and its ``alonet`` modules are imported as in make_golden_criterion.py (inert shells for
    torchvision / cv2 / pytorch_lightning, package shells so the heavy ``__init__`` files do not execute);
  * the ResNet body (torchvision) is replaced by ``helpers.stub_pyramid`` — the SAME seeded convolution stack the tests put under
    this repository's classes — sitting inside the reference's real ``BackboneBase`` / ``Joiner``;
  * ``torchvision.transforms.functional.resize`` (the one torchvision call on the path: the float padding mask,
    detr/backbone.py:127) is given its tensor semantics of the torchvision the reference pins (bilinear, align_corners=False,
    no antialias) = ``F.interpolate``; likewise ``pad`` (constant fill; ``Frame._pad`` of the non-resnet states) = ``F.pad``;
  * ``MSDeformAttnFunction.apply`` (CUDA only) is routed to the reference's own ``ms_deform_attn_core_pytorch`` — the two are
    the same function by the reference's ops/test.py — and the ``is_cuda`` assertion of ``DeformableDETR.forward`` is answered
    by a stand-in ``parameters()``.
Weights: ``helpers.tied_formula_state_dict`` (derived from tensor names), so no checkpoint is stored.

Usage:  python tests/golden/make_golden_models.py        (from the repo root; spawned by make_golden.py)
"""
import importlib
import os
import sys
import types

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
sys.path.insert(0, os.path.dirname(OUT))
import make_golden_criterion as G13  # noqa: E402  (its load() installs the shells and imports the reference's aloscene)

REF = G13.REF
WIDE_QUERIES = {"query_embed.weight": 8.0}   # G15: see helpers.tied_formula_state_dict


def _np(t):
    return t.detach().cpu().numpy()


def load():
    import torch.nn.functional as F

    aloscene = G13.load()[0]
    sys.modules["torchvision.transforms.functional"].resize = (
        lambda img, size, *a, **k: F.interpolate(img, size=list(size), mode="bilinear", align_corners=False))
    # torchvision's constant pad of a tensor, padding = [left, top, right, bottom] (Frame._pad of the "01" / "255" / "minmax_sym" states)
    sys.modules["torchvision.transforms.functional"].pad = (
        lambda img, padding, fill=0, padding_mode="constant": F.pad(img, [padding[0], padding[2], padding[1], padding[3]], value=fill))

    def shell(name, path):
        mod = types.ModuleType(name)
        mod.__path__ = [path]
        sys.modules[name] = mod
        return mod

    al = sys.modules["alonet"]
    al.ALONET_ROOT = REF + "/alonet"
    for sub in ("deformable_detr/ops", "deformable_detr/ops/functions", "deformable_detr/ops/modules", "transformers", "common",
                "detr_panoptic", "detr_panoptic/nn"):
        shell("alonet." + sub.replace("/", "."), REF + "/alonet/" + sub)
    al.common = sys.modules["alonet.common"]
    ns = types.SimpleNamespace(aloscene=aloscene)
    func = importlib.import_module("alonet.deformable_detr.ops.functions.ms_deform_attn_func")
    fpkg = sys.modules["alonet.deformable_detr.ops.functions"]
    fpkg.MSDeformAttnFunction = func.MSDeformAttnFunction
    fpkg.load_MultiScaleDeformableAttention = lambda: None
    mod = importlib.import_module("alonet.deformable_detr.ops.modules.ms_deform_attn")
    # the CUDA op is absent: its apply() becomes the reference's own torch formulation of the same function
    mod.MSDeformAttnFunction = types.SimpleNamespace(
        apply=lambda value, shapes, start, loc, attn, step: func.ms_deform_attn_core_pytorch(value, shapes, loc, attn))
    sys.modules["alonet.deformable_detr.ops.modules"].MSDeformAttn = mod.MSDeformAttn
    tpk = sys.modules["alonet.transformers"]
    tpk.MLP = importlib.import_module("alonet.transformers.mlp").MLP
    ns.pe = importlib.import_module("alonet.transformers.position_encoding")
    tpk.PositionEmbeddingSine = ns.pe.PositionEmbeddingSine
    ns.detr = importlib.import_module("alonet.detr.detr")
    dpk = sys.modules["alonet.detr"]
    dpk.Detr = ns.detr.Detr
    dpk.detr = ns.detr
    al.detr = dpk
    ns.detr_tr = importlib.import_module("alonet.detr.transformer")
    ns.detr_bb = importlib.import_module("alonet.detr.backbone")
    ns.ddetr = importlib.import_module("alonet.deformable_detr.deformable_detr")
    ns.dtr = importlib.import_module("alonet.deformable_detr.deformable_transformer")
    ns.dbb = importlib.import_module("alonet.deformable_detr.backbone")
    npk = sys.modules["alonet.detr_panoptic.nn"]
    npk.FPNstyleCNN = importlib.import_module("alonet.detr_panoptic.nn.FPNstyle").FPNstyleCNN
    npk.MHAttentionMap = importlib.import_module("alonet.detr_panoptic.nn.MHAttention").MHAttentionMap
    ns.pan = importlib.import_module("alonet.detr_panoptic.detr_panoptic")
    return ns


def make_frames(aloscene, sizes, seed, dtype):
    """Frames of different sizes, resnet-normalised by the reference's own Frame, batched by its batch_list."""
    import torch

    gen = torch.Generator().manual_seed(seed)
    raws = [(torch.rand(3, h, w, generator=gen) * 255).round() for h, w in sizes]
    frames = [aloscene.Frame(r.to(dtype), normalization="255", names=("C", "H", "W")).norm_resnet() for r in raws]
    return raws, aloscene.Frame.batch_list(frames)


def deformable_backbone(ns, channels, hidden):
    from helpers import stub_pyramid

    body = stub_pyramid(channels)
    bb = ns.dbb.BackboneBase(body, train_backbone=True, return_interm_layers=True)   # IntermediateLayerGetter is an inert shell
    bb.body = body
    bb.num_channels = list(channels)
    return ns.dbb.Joiner(bb, ns.pe.PositionEmbeddingSine(hidden // 2, normalize=True, center=True))


def on_cpu(model):
    """DeformableDETR.forward asserts its parameters are on cuda (the op has no CPU build); answered here, not edited there."""
    model.parameters = lambda: iter([types.SimpleNamespace(is_cuda=True)])
    return model


def record_forward(save, tag, out, f):
    for key, val in out.items():
        if key == "aux_outputs":
            for i, aux in enumerate(val):
                save[f"{tag}.aux{i}.pred_logits"], save[f"{tag}.aux{i}.pred_boxes"] = f(aux["pred_logits"]), f(aux["pred_boxes"])
        elif hasattr(val, "shape"):
            save[f"{tag}.{key}"] = _np(val) if val.dtype == __import__("torch").bool else f(val)


def record_inference(save, tag, model, out, thresholds, f, **kw):
    for t in thresholds:
        boxes = model.inference(out, threshold=t, **kw)
        name = "none" if t is None else str(t)
        for b, bx in enumerate(boxes):
            save[f"{tag}.inf{name}.boxes{b}"] = f(bx.as_tensor())
            save[f"{tag}.inf{name}.labels{b}"] = f(bx.labels.as_tensor())
            save[f"{tag}.inf{name}.scores{b}"] = f(bx.labels.scores)
            assert bx.boxes_format == "xcyc" and not bx.absolute and bx.names == ("N", None) and bx.labels.encoding == "id"


def g14(ns):
    import torch

    from helpers import tied_formula_state_dict

    f64 = lambda t: _np(t).astype(np.float64)  # noqa: E731
    raws, frames = make_frames(ns.aloscene, [(75, 100), (64, 90)], 1414, torch.float64)
    save = {f"raw{i}": _np(r).astype(np.float32) for i, r in enumerate(raws)}
    save["frames"], save["mask"] = f64(frames.as_tensor()), _np(frames.mask.as_tensor())
    channels, hidden = (8, 12, 16, 24), 64
    for tag, kw in (("plain", dict(with_box_refine=False, activation_fn="sigmoid")),
                    ("refine", dict(with_box_refine=True, activation_fn="sigmoid")),
                    ("softmax", dict(with_box_refine=False, activation_fn="softmax"))):
        tr = ns.dtr.DeformableTransformer(d_model=hidden, nhead=4, num_encoder_layers=2, num_decoder_layers=3,
                                          dim_feedforward=96, dropout=0.0, return_intermediate_dec=True,
                                          num_feature_levels=4, dec_n_points=2, enc_n_points=3)
        model = ns.ddetr.DeformableDETR(deformable_backbone(ns, channels, hidden), tr, num_classes=7, num_queries=12,
                                        num_feature_levels=4, aux_loss=True, return_dec_outputs=True, return_enc_outputs=True,
                                        return_bb_outputs=True, device=None, **kw).double().eval()
        res = model.load_state_dict(tied_formula_state_dict(model))
        assert not res.missing_keys and not res.unexpected_keys
        with torch.no_grad():
            out = on_cpu(model)(frames)
        record_forward(save, tag, out, f64)
        scores = (out["pred_logits"].softmax(-1) if kw["activation_fn"] == "softmax" else out["pred_logits"].sigmoid()).max(-1)[0]
        lo, hi = float(scores.quantile(0.35)), float(scores.quantile(0.7))
        thresholds = (round(lo, 3), round(hi, 3)) if tag != "softmax" else (None, round(lo, 3))
        save[f"{tag}.thresholds"] = np.array([np.nan if t is None else t for t in thresholds])
        record_inference(save, tag, model, out, thresholds, f64)
        kept = [int(v.shape[0]) for k, v in save.items() if k.startswith(f"{tag}.inf") and ".boxes" in k]
        print("g14", tag, "keys", len(out), "kept per (threshold, image):", kept)
    np.savez_compressed(os.path.join(OUT, "g14_deformable_detr.npz"), **save)


def g14b(ns):
    import torch

    from helpers import tied_formula_state_dict

    f32 = lambda t: _np(t).astype(np.float32)  # noqa: E731
    raws, frames = make_frames(ns.aloscene, [(200, 264), (168, 240)], 1415, torch.float64)
    save = {f"raw{i}": _np(r).astype(np.float32) for i, r in enumerate(raws)}
    channels, hidden = (32, 64, 128, 256), 256
    tr = ns.dtr.DeformableTransformer(d_model=hidden, nhead=8, num_encoder_layers=1, num_decoder_layers=2, dim_feedforward=1024,
                                      dropout=0.0, return_intermediate_dec=True, num_feature_levels=4, dec_n_points=4,
                                      enc_n_points=4)
    model = ns.ddetr.DeformableDETR(deformable_backbone(ns, channels, hidden), tr, num_classes=11, num_queries=20,
                                    num_feature_levels=4, aux_loss=True, device=None).double().eval()
    res = model.load_state_dict(tied_formula_state_dict(model))
    assert not res.missing_keys and not res.unexpected_keys
    with torch.no_grad():
        out = on_cpu(model)(frames)
    record_forward(save, "d256", out, f32)
    scores = out["pred_logits"].sigmoid().max(-1)[0]
    thresholds = (round(float(scores.quantile(0.4)), 3),)
    save["d256.thresholds"] = np.array(thresholds)
    record_inference(save, "d256", model, out, thresholds, f32)
    np.savez_compressed(os.path.join(OUT, "g14b_deformable_detr_d256.npz"), **save)
    print("g14b logits", tuple(out["pred_logits"].shape), "score range", float(scores.min()), float(scores.max()))


def g15(ns):
    import torch

    from helpers import stub_pyramid, tied_formula_state_dict

    f64 = lambda t: _np(t).astype(np.float64)  # noqa: E731
    raws, frames = make_frames(ns.aloscene, [(75, 100), (64, 90)], 1515, torch.float64)
    save = {f"raw{i}": _np(r).astype(np.float32) for i, r in enumerate(raws)}
    channels, hidden = (8, 12, 16, 24), 128   # FPNstyleCNN wants hidden / 16 divisible by 8

    # ---- Detr over the stub pyramid (detr/backbone.py BackboneBase / Joiner are the reference's) ---------------------------
    body = stub_pyramid(channels)
    bb = ns.detr_bb.BackboneBase(body, train_backbone=True, num_channels=channels[-1], return_interm_layers=True)
    bb.body = body
    joiner = ns.detr_bb.Joiner(bb, ns.pe.PositionEmbeddingSine(hidden // 2, normalize=True))
    joiner.num_channels = channels[-1]
    tr = ns.detr_tr.Transformer(d_model=hidden, nhead=8, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=96,
                                dropout=0.0, return_intermediate_dec=True)
    detr = ns.detr.Detr(joiner, tr, num_classes=7, num_queries=10, aux_loss=True, return_dec_outputs=True,
                        return_enc_outputs=True, return_bb_outputs=True).double().eval()
    res = detr.load_state_dict(tied_formula_state_dict(detr, scale=WIDE_QUERIES))
    assert not res.missing_keys and not res.unexpected_keys
    with torch.no_grad():
        out = detr(frames)
    record_forward(save, "detr", out, f64)
    scores = out["pred_logits"].softmax(-1).max(-1)[0]
    thresholds = (None, round(float(scores.quantile(0.5)), 3))
    save["detr.thresholds"] = np.array([np.nan, thresholds[1]])
    record_inference(save, "detr", detr, out, thresholds, f64)
    # background_class=-1: every query is kept (the plumbing call of configs[0])
    for b, bx in enumerate(detr.inference(out, background_class=-1)):
        save[f"detr.infall.boxes{b}"], save[f"detr.infall.labels{b}"] = f64(bx.as_tensor()), f64(bx.labels.as_tensor())

    def panoptic(tag, base, quantile):
        """PanopticHead over ``base``: weights of the WHOLE head (its ``detr.`` sub-module included) from the formula, then forward
        with a score threshold at ``quantile`` of the base model's own scores (so that the query filter keeps a proper subset)."""
        head = ns.pan.PanopticHead(base, fpn_list=[channels[2], channels[1], channels[0]]).double().eval()
        assert not head.load_state_dict(tied_formula_state_dict(head, scale=WIDE_QUERIES)).missing_keys
        deformable = tag == "pan_deformable"
        with torch.no_grad():
            logits = (on_cpu(base) if deformable else base)(frames)["pred_logits"]
            s = (logits.sigmoid() if deformable else logits.softmax(-1)).max(-1)[0]
            thr = round(float((s if deformable else s[0]).quantile(quantile)), 6)
            pout = (on_cpu(head) if deformable else head)(frames, threshold=thr)
        save[f"{tag}.threshold"] = np.array(thr)
        save[f"{tag}.pred_masks"] = f64(pout["pred_masks"])
        save[f"{tag}.pred_logits"], save[f"{tag}.pred_boxes"] = f64(pout["pred_logits"]), f64(pout["pred_boxes"])
        for b, flt in enumerate(pout["pred_masks_info"]["filters"]):
            save[f"{tag}.filter{b}"] = _np(flt)
        save[f"{tag}.frame_size"] = np.array(pout["pred_masks_info"]["frame_size"])
        boxes, masks = head.inference(pout, maskth=0.5, threshold=thr)
        for b, (bx, mk) in enumerate(zip(boxes, masks)):
            save[f"{tag}.inf.boxes{b}"], save[f"{tag}.inf.labels{b}"] = f64(bx.as_tensor()), f64(bx.labels.as_tensor())
            save[f"{tag}.inf.scores{b}"] = f64(bx.labels.scores)
            save[f"{tag}.inf.masks{b}"] = _np(mk.as_tensor()).astype(np.uint8)
            assert mk.names == ("N", "H", "W")
        print("g15", tag, "pred_masks", tuple(pout["pred_masks"].shape), "kept", [int(m.shape[0]) for m in masks],
              "mask pixels", [int(m.as_tensor().sum()) for m in masks])

    panoptic("pan_detr", detr, 0.5)
    # ---- PanopticHead over DeformableDETR (BASELINE configs[4]) -------------------------------------------------------------
    dtr = ns.dtr.DeformableTransformer(d_model=hidden, nhead=8, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=96,
                                       dropout=0.0, return_intermediate_dec=True, num_feature_levels=4, dec_n_points=2,
                                       enc_n_points=3)
    ddetr = ns.ddetr.DeformableDETR(deformable_backbone(ns, channels, hidden), dtr, num_classes=7, num_queries=12,
                                    num_feature_levels=4, aux_loss=False, device=None).double().eval()
    panoptic("pan_deformable", ddetr, 0.6)
    np.savez_compressed(os.path.join(OUT, "g15_detr_panoptic.npz"), **save)


def g16(ns):
    import torch

    A = ns.aloscene
    gen = torch.Generator().manual_seed(1616)
    raw = [(torch.rand(3, h, w, generator=gen) * 255).round() for h, w in ((20, 30), (12, 18), (20, 25))]
    save = {f"raw{i}": _np(r) for i, r in enumerate(raw)}
    f = A.Frame(raw[0], normalization="255", names=("C", "H", "W"))
    states = {"255": f, "01": f.norm01(), "minmax_sym": f.norm_minmax_sym(), "resnet": f.norm_resnet()}
    for src, fr in states.items():
        assert fr.normalization == src
        for dst, conv in (("01", fr.norm01), ("255", fr.norm255), ("minmax_sym", fr.norm_minmax_sym), ("resnet", fr.norm_resnet)):
            got = conv()
            assert got.normalization == dst and got.names == ("C", "H", "W")
            save[f"norm.{src}.{dst}"] = _np(got.as_tensor())
    res = states["resnet"]
    save["resnet.mean_std"] = np.array(res.mean_std)
    assert f.mean_std is None or True
    frames = [A.Frame(r, normalization="255", names=("C", "H", "W")).norm_resnet() for r in raw]
    batch = A.Frame.batch_list(frames)
    assert batch.names == ("B", "C", "H", "W") and batch.normalization == "resnet" and batch.mask.names == ("B", "C", "H", "W")
    save["batch.values"], save["batch.mask"] = _np(batch.as_tensor()), _np(batch.mask.as_tensor())
    save["batch.mask_dtype"] = np.array(str(batch.mask.as_tensor().dtype))
    save["batch.HW"] = np.array(batch.HW)
    one = frames[1].batch()
    assert one.names == ("B", "C", "H", "W")
    save["batch.single"] = _np(one.as_tensor())
    single = A.Frame.batch_list([frames[1]])
    save["batch_list.single.values"], save["batch_list.single.mask"] = _np(single.as_tensor()), _np(single.mask.as_tensor())
    # minmax_sym pairs, the RAFT input state (raft/raft.py:157-158)
    pair = A.Frame.batch_list([A.Frame(r, normalization="255", names=("C", "H", "W")).norm_minmax_sym() for r in raw[:2]])
    assert pair.normalization == "minmax_sym"
    save["pair.values"], save["pair.mask"] = _np(pair.as_tensor()), _np(pair.mask.as_tensor())
    # FrozenBatchNorm2d of the reference's backbone (detr/backbone.py:50-93: eps added before the rsqrt), formula buffers, and the
    # folded form this repository convolves with (conv followed by the frozen norm == conv with scaled weights + shift)
    from helpers import formula_state_dict

    fbn = ns.detr_bb.FrozenBatchNorm2d(6)
    fbn.load_state_dict(formula_state_dict(fbn.state_dict()))
    xg = torch.randn(2, 6, 5, 7, generator=gen)
    save["fbn.x"], save["fbn.out"] = _np(xg), _np(fbn(xg))
    np.savez_compressed(os.path.join(OUT, "g16_frame_io.npz"), **save)
    print("g16 batch", tuple(batch.shape), "mask sum per frame", [int(m.sum()) for m in batch.mask.as_tensor()])


def main():
    import torch

    if not os.path.isdir(REF):
        sys.exit("make_golden_models.py needs the reference checkout at /root/reference (build container only)")
    torch.set_num_threads(4)
    ns = load()
    todo = [fn for fn in (g14, g14b, g15, g16) if len(sys.argv) == 1 or fn.__name__ in sys.argv[1:]]
    for fn in todo:
        fn(ns)
        print("wrote", fn.__name__)


if __name__ == "__main__":
    main()
