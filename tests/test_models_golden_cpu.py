"""Model-level glue and I/O objects against the REFERENCE'S OWN classes (G14 / G14b / G15 / G16), on the CPU.

tests/golden/make_golden_models.py ran the reference's ``DeformableDETR`` / ``Detr`` / ``PanopticHead`` (forward, heads,
``inference()``) over ``helpers.stub_pyramid`` and the reference's real ``aloscene.Frame``; here this repository's classes get the
same convolution stack, the same name-derived weights and the same raw frames.  What is pinned: ``Frame.norm_resnet`` /
``batch_list`` (padding values + mask), the backbone's mask resize, positional encodings, ``input_proj`` + the stride-64 level,
the transformer call, the heads (``inverse_sigmoid`` refinement), the output dictionary and ``inference()``'s selection.
The deformable attention itself runs through the reference's explicit ``is_tracing`` escape hatch on the CPU (the product has no
CPU kernel); tests/test_models_gpu.py repeats G14 / G14b / G15 through the HIP op.
"""
import numpy as np
import pytest
import torch

import aloscene
from helpers import stub_pyramid, tied_formula_state_dict

t = torch.from_numpy


# ---- builders shared with the GPU tests ---------------------------------------------------------------------------------------
def batch_from_raw(g, n=2, dtype=torch.float64):
    """The fixture's raw 0..255 frames through THIS repository's Frame: norm_resnet + batch_list."""
    frames = [aloscene.Frame(t(g[f"raw{i}"]).to(dtype), normalization="255", names=("C", "H", "W")).norm_resnet() for i in range(n)]
    return aloscene.Frame.batch_list(frames)


def deformable_joiner(channels, hidden):
    from alonet.deformable_detr.backbone import BackboneBase, Joiner
    from alonet.transformers import PositionEmbeddingSine

    bb = BackboneBase(stub_pyramid(channels), train_backbone=True, return_interm_layers=True)
    bb.num_channels = list(channels)
    return Joiner(bb, PositionEmbeddingSine(hidden // 2, normalize=True, center=True))


def build_deformable(channels, hidden, nhead, enc, dec, ffn, dec_p, enc_p, num_classes, num_queries, scale=None, **kw):
    from alonet.deformable_detr import DeformableDETR, DeformableTransformer

    tr = DeformableTransformer(d_model=hidden, nhead=nhead, num_encoder_layers=enc, num_decoder_layers=dec, dim_feedforward=ffn,
                               dropout=0.0, return_intermediate_dec=True, num_feature_levels=4, dec_n_points=dec_p,
                               enc_n_points=enc_p)
    model = DeformableDETR(deformable_joiner(channels, hidden), tr, num_classes=num_classes, num_queries=num_queries,
                           num_feature_levels=4, device=None, **kw).double().eval()
    res = model.load_state_dict(tied_formula_state_dict(model))
    assert not res.missing_keys and not res.unexpected_keys
    return model


WIDE_QUERIES = {"query_embed.weight": 8.0}   # as in make_golden_models.py (G15)
G14_CONFIGS = {"plain": dict(with_box_refine=False, activation_fn="sigmoid"),
               "refine": dict(with_box_refine=True, activation_fn="sigmoid"),
               "softmax": dict(with_box_refine=False, activation_fn="softmax")}


def build_g14(tag):
    return build_deformable((8, 12, 16, 24), 64, 4, 2, 3, 96, 2, 3, 7, 12, aux_loss=True, return_dec_outputs=True,
                            return_enc_outputs=True, return_bb_outputs=True, **G14_CONFIGS[tag])


def build_g14b():
    return build_deformable((32, 64, 128, 256), 256, 8, 1, 2, 1024, 4, 4, 11, 20, aux_loss=True)


def build_g15_detr():
    from alonet.detr import Detr, Transformer
    from alonet.detr.backbone import BackboneBase, Joiner
    from alonet.transformers import PositionEmbeddingSine

    channels, hidden = (8, 12, 16, 24), 128
    bb = BackboneBase(stub_pyramid(channels), train_backbone=True, num_channels=channels[-1], return_interm_layers=True)
    joiner = Joiner(bb, PositionEmbeddingSine(hidden // 2, normalize=True))
    joiner.num_channels = channels[-1]
    tr = Transformer(d_model=hidden, nhead=8, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=96, dropout=0.0,
                     return_intermediate_dec=True)
    model = Detr(joiner, tr, num_classes=7, num_queries=10, aux_loss=True, return_dec_outputs=True, return_enc_outputs=True,
                 return_bb_outputs=True).double().eval()
    res = model.load_state_dict(tied_formula_state_dict(model, scale=WIDE_QUERIES))
    assert not res.missing_keys and not res.unexpected_keys
    return model


def build_g15_deformable():
    return build_deformable((8, 12, 16, 24), 128, 8, 2, 2, 96, 2, 3, 7, 12, aux_loss=False)


def build_g15_panoptic(base):
    from alonet.detr_panoptic import PanopticHead

    head = PanopticHead(base, fpn_list=[16, 12, 8]).double().eval()
    assert not head.load_state_dict(tied_formula_state_dict(head, scale=WIDE_QUERIES)).missing_keys   # re-loads the ``detr.`` sub-module too
    return head


def check_forward(out, g, tag, tol, skip=()):
    """Every tensor the reference's forward returned under ``tag`` (aux outputs included) against ``out``."""
    seen = 0
    for key in g.files:
        if not key.startswith(tag + ".") or ".inf" in key or key.endswith("thresholds"):
            continue
        name = key[len(tag) + 1:]
        if name in skip:
            continue
        if name.startswith("aux"):
            idx, field = name[3:].split(".", 1)
            got = out["aux_outputs"][int(idx)][field]
        else:
            got = out[name]
        want = g[key]
        if want.dtype == np.bool_:
            assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), want), key
        else:
            got = got.detach().double().cpu().numpy()
            assert got.shape == want.shape, (key, got.shape, want.shape)
            err = np.abs(got - want).max()
            assert err <= tol, (key, err)
        seen += 1
    return seen


def check_inference(model, out, g, tag, tol, **kw):
    thresholds = g[f"{tag}.thresholds"]
    for thr in thresholds:
        thr, name = (None, "none") if np.isnan(thr) else (float(thr), str(float(thr)))
        boxes = model.inference(out, threshold=thr, **kw)
        for b, bx in enumerate(boxes):
            assert isinstance(bx, aloscene.BoundingBoxes2D) and bx.boxes_format == "xcyc" and not bx.absolute
            assert bx.names == ("N", None) and isinstance(bx.labels, aloscene.Labels) and bx.labels.encoding == "id"
            want = g[f"{tag}.inf{name}.boxes{b}"]
            assert tuple(bx.shape) == want.shape, (tag, name, b, tuple(bx.shape), want.shape)   # the SAME queries were kept
            assert not bx.is_cuda and bx.labels.as_tensor().dtype == torch.float32
            np.testing.assert_array_equal(bx.labels.as_tensor().numpy(), g[f"{tag}.inf{name}.labels{b}"])
            np.testing.assert_allclose(bx.as_tensor().double().numpy(), want, rtol=0, atol=tol)
            np.testing.assert_allclose(bx.labels.scores.double().cpu().numpy(), g[f"{tag}.inf{name}.scores{b}"], rtol=0, atol=tol)


# ---- G16: Frame ---------------------------------------------------------------------------------------------------------------
def test_g16_frame_normalisations_match_the_reference_frame(golden):
    g = golden("g16_frame_io.npz")
    f = aloscene.Frame(t(g["raw0"]), normalization="255", names=("C", "H", "W"))
    states = {"255": f, "01": f.norm01(), "minmax_sym": f.norm_minmax_sym(), "resnet": f.norm_resnet()}
    for src, fr in states.items():
        assert fr.normalization == src
        for dst, conv in (("01", fr.norm01), ("255", fr.norm255), ("minmax_sym", fr.norm_minmax_sym), ("resnet", fr.norm_resnet)):
            got = conv()
            assert got.normalization == dst and got.names == ("C", "H", "W") and isinstance(got, aloscene.Frame)
            np.testing.assert_allclose(got.as_tensor().numpy(), g[f"norm.{src}.{dst}"], rtol=0, atol=2e-5 if dst == "255" else 2e-6,
                                       err_msg=f"{src} -> {dst}")
    assert np.allclose(np.array(states["resnet"].mean_std), g["resnet.mean_std"])


def test_g16_batch_list_pads_like_the_reference(golden):
    """spatial_augmented_tensor.py:323-419 + frame.py:555-600: a resnet-normalised frame is padded with the normalised value
    of a BLACK pixel (-mean / std), a minmax_sym frame with -1; the mask child is float32, 1 on the padding."""
    g = golden("g16_frame_io.npz")
    frames = [aloscene.Frame(t(g[f"raw{i}"]), normalization="255", names=("C", "H", "W")).norm_resnet() for i in range(3)]
    batch = aloscene.Frame.batch_list(frames)
    assert batch.names == ("B", "C", "H", "W") and batch.normalization == "resnet" and batch.mask.names == ("B", "C", "H", "W")
    assert tuple(batch.HW) == tuple(g["batch.HW"]) and str(batch.mask.as_tensor().dtype) == str(g["batch.mask_dtype"])
    np.testing.assert_array_equal(batch.mask.as_tensor().numpy(), g["batch.mask"])
    np.testing.assert_allclose(batch.as_tensor().numpy(), g["batch.values"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(frames[1].batch().as_tensor().numpy(), g["batch.single"], rtol=0, atol=2e-6)
    single = aloscene.Frame.batch_list([frames[1]])
    np.testing.assert_allclose(single.as_tensor().numpy(), g["batch_list.single.values"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(single.mask.as_tensor().numpy(), g["batch_list.single.mask"])
    pair = aloscene.Frame.batch_list([aloscene.Frame(t(g[f"raw{i}"]), normalization="255", names=("C", "H", "W")).norm_minmax_sym()
                                      for i in range(2)])
    assert pair.normalization == "minmax_sym"
    np.testing.assert_allclose(pair.as_tensor().numpy(), g["pair.values"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(pair.mask.as_tensor().numpy(), g["pair.mask"])


def test_g16_frozen_batch_norm_and_its_folded_form_match_the_reference(golden):
    """The reference's FrozenBatchNorm2d (detr/backbone.py:50-93) on formula buffers vs this repository's module, and vs the folded
    convolution the inference path actually runs (conv weights scaled, shift as bias)."""
    from alonet.detr.backbone import FrozenBatchNorm2d, folded_conv_bn
    from helpers import formula_state_dict

    g = golden("g16_frame_io.npz")
    fbn = FrozenBatchNorm2d(6)
    fbn.load_state_dict(formula_state_dict(fbn.state_dict()))
    x = t(g["fbn.x"])
    np.testing.assert_allclose(fbn(x).numpy(), g["fbn.out"], rtol=0, atol=2e-6)
    conv = torch.nn.Conv2d(6, 6, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.eye(6).view(6, 6, 1, 1))          # identity convolution: the folded pair must reproduce the norm alone
    with torch.no_grad():
        w, b = folded_conv_bn(conv.eval(), fbn)
        folded = torch.nn.functional.conv2d(x, w.float(), b.float())
    np.testing.assert_allclose(folded.numpy(), g["fbn.out"], rtol=0, atol=5e-6)


# ---- G14: DeformableDETR --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["plain", "refine", "softmax"])
def test_g14_deformable_detr_forward_and_inference_match_the_reference(golden, tag):
    g = golden("g14_deformable_detr.npz")
    frames = batch_from_raw(g)
    np.testing.assert_allclose(frames.as_tensor().numpy(), g["frames"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(frames.mask.as_tensor().numpy(), g["mask"])
    model = build_g14(tag)
    with torch.no_grad():
        out = model(frames, is_tracing=None)
    assert out["activation_fn"] == G14_CONFIGS[tag]["activation_fn"]
    assert set(k for k in out if not k.startswith("_")) == {"pred_logits", "pred_boxes", "activation_fn", "aux_outputs", "dec_outputs",
                                                            "enc_outputs"} | {f"bb_lvl{i}_{n}_outputs" for i in range(4)
                                                                              for n in ("src", "mask", "pos")}
    # fp64 end to end; the residue is the reference's own fp32 islands (positional encoding, reference points)
    assert check_forward(out, g, tag, 5e-6) >= 20
    check_inference(model, out, g, tag, 5e-6)


def test_g14b_deformable_detr_at_the_detr_family_width(golden):
    g = golden("g14b_deformable_detr_d256.npz")
    model = build_g14b()
    with torch.no_grad():
        out = model(batch_from_raw(g), is_tracing=None)
    assert check_forward(out, g, "d256", 2e-5) >= 4
    check_inference(model, out, g, "d256", 2e-5)


# ---- G15: Detr and PanopticHead ---------------------------------------------------------------------------------------------------
def test_g15_detr_forward_and_inference_match_the_reference(golden):
    g = golden("g15_detr_panoptic.npz")
    frames = batch_from_raw(g)
    model = build_g15_detr()
    with torch.no_grad():
        out = model(frames)
    assert check_forward(out, g, "detr", 5e-6) >= 18
    check_inference(model, out, g, "detr", 5e-6)
    every = model.inference(out, background_class=-1)
    for b, bx in enumerate(every):
        np.testing.assert_allclose(bx.as_tensor().double().numpy(), g[f"detr.infall.boxes{b}"], rtol=0, atol=5e-6)
        np.testing.assert_array_equal(bx.labels.as_tensor().numpy(), g[f"detr.infall.labels{b}"])


def check_panoptic(head, frames, g, tag, tol, mask_tol):
    thr = float(g[f"{tag}.threshold"])
    with torch.no_grad():
        out = head(frames, threshold=thr)
    for b, flt in enumerate(out["pred_masks_info"]["filters"]):
        np.testing.assert_array_equal(flt.cpu().numpy(), g[f"{tag}.filter{b}"])
    assert tuple(out["pred_masks_info"]["frame_size"]) == tuple(g[f"{tag}.frame_size"])
    for key in ("pred_logits", "pred_boxes"):
        assert np.abs(out[key].double().cpu().numpy() - g[f"{tag}.{key}"]).max() <= tol, key
    want = g[f"{tag}.pred_masks"]
    got = out["pred_masks"].double().cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() <= mask_tol * max(1.0, np.abs(want).max())
    boxes, masks = head.inference(out, maskth=0.5, threshold=thr)
    for b, (bx, mk) in enumerate(zip(boxes, masks)):
        assert isinstance(mk, aloscene.Mask) and mk.names == ("N", "H", "W")
        np.testing.assert_allclose(bx.as_tensor().double().numpy(), g[f"{tag}.inf.boxes{b}"], rtol=0, atol=tol)
        np.testing.assert_array_equal(bx.labels.as_tensor().numpy(), g[f"{tag}.inf.labels{b}"])
        assert_masks_equal_up_to_ties(mk, g, tag, b)


def assert_masks_equal_up_to_ties(mk, g, tag, b, gap=1e-6):
    """``aloscene.Mask`` of image ``b`` against the reference's.  The masks are the per-pixel arg-max over the kept queries of the
    up-sampled, sigmoid-ed, thresholded mask logits (detr_panoptic.py:262-283): a pixel may differ ONLY where the reference's own
    two best probabilities are closer than ``gap`` (an arithmetic tie: DETR's queries are near-copies of each other under random
    weights) or where the best one sits within ``gap`` of the 0.5 threshold."""
    import torch.nn.functional as F

    want_m = g[f"{tag}.inf.masks{b}"]
    got_m = mk.as_tensor().cpu().numpy()
    assert got_m.shape == want_m.shape and got_m.dtype == np.int64
    diff = (got_m != want_m).any(0)
    if not diff.any():
        return
    keep = t(g[f"{tag}.filter{b}"])
    logits = t(g[f"{tag}.pred_masks"])[b:b + 1, :int(keep.sum())]
    prob = F.interpolate(logits, size=want_m.shape[-2:], mode="bilinear", align_corners=False).sigmoid()[0]
    top = prob.topk(min(2, prob.shape[0]), dim=0)[0]
    tie = (top[0] - top[-1] < gap) | ((top[0] - 0.5).abs() < gap)
    assert bool(tie[t(diff)].all()) and diff.mean() < max(2e-3, 20 * gap), (tag, b, int(diff.sum()))


def test_g15_panoptic_head_over_detr_matches_the_reference(golden):
    g = golden("g15_detr_panoptic.npz")
    head = build_g15_panoptic(build_g15_detr())
    check_panoptic(head, batch_from_raw(g), g, "pan_detr", 5e-6, 1e-7)


def test_g15_panoptic_head_over_deformable_detr_matches_the_reference(golden):
    """BASELINE configs[4]'s composition.  On the CPU the deformable attention needs ``is_tracing``; the head passes its keyword
    arguments through to the detector as the reference does (detr_panoptic.py:170)."""
    g = golden("g15_detr_panoptic.npz")
    head = build_g15_panoptic(build_g15_deformable())
    thr = float(g["pan_deformable.threshold"])
    frames = batch_from_raw(g)
    from alonet.detr_panoptic.utils import get_mask_queries

    def queries(**kw):   # ``is_tracing`` is for the detector only; the reference's get_outs_filter would refuse it too
        kw.pop("is_tracing")
        return get_mask_queries(**kw)

    with torch.no_grad():
        out = head(frames, get_filter_fn=queries, threshold=thr, is_tracing=None)
    for b, flt in enumerate(out["pred_masks_info"]["filters"]):
        np.testing.assert_array_equal(flt.numpy(), g[f"pan_deformable.filter{b}"])
    want = g["pan_deformable.pred_masks"]
    assert np.abs(out["pred_masks"].numpy() - want).max() <= 5e-6 * max(1.0, np.abs(want).max())
    boxes, masks = head.inference(out, maskth=0.5, threshold=thr)
    for b, (bx, mk) in enumerate(zip(boxes, masks)):
        np.testing.assert_allclose(bx.as_tensor().double().numpy(), g[f"pan_deformable.inf.boxes{b}"], rtol=0, atol=5e-6)
        np.testing.assert_array_equal(bx.labels.as_tensor().numpy(), g[f"pan_deformable.inf.labels{b}"])
        assert_masks_equal_up_to_ties(mk, g, "pan_deformable", b)
