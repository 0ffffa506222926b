"""Model graphs (callers of the hot path) against the reference's outputs, on the CPU.

The gather / correlation ops have no CPU implementation in the product, so here the graphs run through the reference's
own explicit escape hatches: ``is_tracing`` (pure-torch MSDA branch) and RAFT's ``corr_block=`` hook (fed the oracle's
torch CorrBlock).  What is pinned is everything AROUND the kernels: module wiring, parameter names, arithmetic.
"""
import numpy as np
import pytest
import torch

import aloscene
import torch_ref as T
from alonet.deformable_detr import DeformableDetrR50, DeformableTransformer
from alonet.raft import RAFT
from alonet.transformers import PositionEmbeddingSine
from helpers import formula_state_dict

t = torch.from_numpy


def test_position_encoding_matches_reference(golden):
    g = golden("g9_posenc.npz")
    ft = torch.zeros(2, 4, 7, 9)
    out_c = PositionEmbeddingSine(16, normalize=True, center=True)((ft, t(g["mask"])))
    out_d = PositionEmbeddingSine(16, normalize=True)((ft, t(g["mask"])))
    np.testing.assert_allclose(out_c.numpy(), g["centered"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out_d.numpy(), g["default"], rtol=0, atol=1e-6)


def build_g5_transformer(g):
    d_model, nhead, enc, dec, ffn, L, dec_p, enc_p = (int(x) for x in g["cfg"])
    tr = DeformableTransformer(d_model=d_model, nhead=nhead, num_encoder_layers=enc, num_decoder_layers=dec,
                               dim_feedforward=ffn, dropout=0.0, return_intermediate_dec=True, num_feature_levels=L,
                               dec_n_points=dec_p, enc_n_points=enc_p).double().eval()
    res = tr.load_state_dict(formula_state_dict(tr.state_dict()))
    assert not res.missing_keys and not res.unexpected_keys
    return tr, L


@pytest.mark.parametrize("fixture", ["g5_deformable_transformer.npz", "g12_deformable_transformer_d256.npz"])
def test_deformable_transformer_graph_matches_reference(golden, fixture):
    g = golden(fixture)
    tr, L = build_g5_transformer(g)
    srcs = [t(g[f"src{i}"]).double() for i in range(L)]
    poss = [t(g[f"pos{i}"]).double() for i in range(L)]
    masks = [t(g[f"mask{i}"]) for i in range(L)]
    with torch.no_grad():
        out = tr(srcs, masks, poss, t(g["query_embed"]).double(), is_tracing=None)
    # G5 is stored in float64; G12 (the DETR-family width) holds the reference's fp64 outputs rounded to float32
    tol = dict(rtol=1e-9, atol=1e-10) if g["hs"].dtype == np.float64 else dict(rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["hs"].numpy(), g["hs"], **tol)
    np.testing.assert_allclose(out["inter_references_out"].numpy(), g["inter_references_out"], **tol)
    if "init_reference_out" in g.files:
        np.testing.assert_allclose(out["init_reference_out"].numpy(), g["init_reference_out"], **tol)
    for i in range(L):
        np.testing.assert_allclose(out["memory"][i].numpy(), g[f"memory{i}"], **tol)


def test_raft_graph_matches_reference(golden):
    g = golden("g7_raft.npz")
    model = RAFT(corr_block=T.CorrBlockRef).eval()
    res = model.load_state_dict(formula_state_dict(model.state_dict()))
    assert not res.missing_keys and not res.unexpected_keys
    f1 = aloscene.Frame(t(g["img1"]).float(), normalization="minmax_sym", names=("B", "C", "H", "W"))
    f2 = aloscene.Frame(t(g["img2"]).float(), normalization="minmax_sym", names=("B", "C", "H", "W"))
    with torch.no_grad():
        outs = model(f1, f2, iters=4)
    flows = np.stack([o["flow"].numpy() for o in outs])
    assert np.isfinite(g["flow"]).all() and np.isfinite(flows).all()
    np.testing.assert_allclose(flows, g["flow"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(outs[-1]["up_flow"].numpy(), g["up_flow_last"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(outs[0]["up_flow"].numpy(), g["up_flow_first"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(outs[-1]["hidden_state"].numpy(), g["hidden_last"], rtol=0, atol=2e-4)
    flow_up = model.inference(outs, only_last=True)
    assert isinstance(flow_up, aloscene.Flow) and flow_up.names == ("B", "C", "H", "W") and flow_up.shape == (2, 2, 128, 160)
    with torch.no_grad():
        last = model(f1, f2, iters=4, only_last=True)
    assert "up_flow" in last[-1] and "up_flow" not in last[0]


def test_state_dict_layout_is_the_reference_checkpoint_layout():
    """Key names a reference checkpoint carries; counts from the reference modules themselves."""
    det = DeformableDetrR50(device=None, aux_loss=False)
    keys = set(det.state_dict())
    for k in ("backbone.0.body.conv1.weight", "backbone.0.body.bn1.running_var",
              "backbone.0.body.layer1.0.downsample.0.weight", "backbone.0.body.layer4.2.conv3.weight",
              "input_proj.0.0.weight", "input_proj.3.1.bias", "query_embed.weight", "class_embed.5.bias",
              "bbox_embed.0.layers.2.weight", "transformer.level_embed", "transformer.reference_points.weight",
              "transformer.encoder.layers.5.self_attn.sampling_offsets.bias",
              "transformer.decoder.layers.0.cross_attn.value_proj.weight",
              "transformer.decoder.layers.3.self_attn.in_proj_weight", "transformer.decoder.layers.5.norm3.bias"):
        assert k in keys, k
    assert not any("num_batches_tracked" in k for k in keys)
    assert det.state_dict()["query_embed.weight"].shape == (300, 512)
    assert sum(1 for k in keys if k.startswith("transformer.")) == 231
    assert abs(sum(p.numel() for p in det.parameters() if p.requires_grad) - 39.85e6) < 0.05e6
    raft = RAFT()
    rk = set(raft.state_dict())
    assert len(rk) == 179 and sum(p.numel() for p in raft.parameters()) == 5257536
    for k in ("fnet.conv1.weight", "fnet.layer2.0.downsample.0.weight", "cnet.norm1.running_mean",
              "cnet.layer3.0.downsample.1.num_batches_tracked", "update_block.encoder.convc1.weight",
              "update_block.gru.convq2.bias", "update_block.flow_head.conv2.weight", "update_block.mask.2.weight"):
        assert k in rk, k
    assert not any(k.startswith("fnet.") and "norm" in k for k in rk)  # InstanceNorm2d: no affine, no stats
    assert raft.state_dict()["update_block.encoder.convc1.weight"].shape == (256, 324, 1, 1)


def test_frame_io_contract():
    img = torch.rand(3, 20, 30) * 255
    f = aloscene.Frame(img, normalization="255")
    r = f.norm_resnet()
    assert r.normalization == "resnet" and r.mean_std[0] == (0.485, 0.456, 0.406)
    np.testing.assert_allclose(r.norm255().as_tensor().numpy(), img.numpy(), atol=1e-4)  # unittest/test_frame.py bar
    np.testing.assert_allclose(f.norm_minmax_sym().norm01().norm255().as_tensor().numpy(), img.numpy(), atol=1e-4)
    small = aloscene.Frame(torch.rand(3, 12, 18) * 255).norm_resnet()
    batch = aloscene.Frame.batch_list([r, small])
    assert batch.names == ("B", "C", "H", "W") and batch.shape == (2, 3, 20, 30) and batch.normalization == "resnet"
    assert batch.mask.names == ("B", "C", "H", "W") and batch.mask.shape == (2, 1, 20, 30)
    m = batch.mask.as_tensor()
    assert m[0].sum() == 0 and m[1, 0, :12, :18].sum() == 0 and m[1].sum() == 20 * 30 - 12 * 18
    # the padding holds a black pixel in the frame's normalisation (reference frame.py:555-600; values pinned by G16)
    black = -torch.tensor(r.mean_std[0]) / torch.tensor(r.mean_std[1])
    assert torch.allclose(batch.as_tensor()[1, :, 12:, :], black.view(3, 1, 1).expand(3, 8, 30))
    assert type(batch.as_tensor()) is torch.Tensor


def test_detr_transformer_graph_matches_reference(golden):
    from alonet.detr import Transformer

    g = golden("g10_detr_transformer.npz")
    tr = Transformer(d_model=64, nhead=4, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=96, dropout=0.0,
                     return_intermediate_dec=True).double().eval()
    res = tr.load_state_dict(formula_state_dict(tr.state_dict()))
    assert not res.missing_keys and not res.unexpected_keys
    with torch.no_grad():
        out = tr(t(g["src"]), t(g["mask"]), t(g["query"]), t(g["pos"]))
    np.testing.assert_allclose(out["hs"].numpy(), g["hs"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(out["memory"].numpy(), g["memory"], rtol=1e-9, atol=1e-10)


def test_detr_r50_cpu_plumbing():
    """BASELINE configs[0]: DetrR50 inference on a Frame on the CPU (smaller frame here to keep the suite fast)."""
    from alonet.detr import DetrR50

    torch.manual_seed(0)
    model = DetrR50(num_classes=91, aux_loss=True).eval()  # default device: cpu
    keys = set(model.state_dict())
    for k in ("backbone.0.body.layer4.2.bn3.running_var", "input_proj.weight", "query_embed.weight",
              "transformer.encoder.layers.5.self_attn.in_proj_weight", "transformer.decoder.layers.0.multihead_attn.out_proj.bias",
              "transformer.decoder.norm.weight", "class_embed.weight", "bbox_embed.layers.2.bias"):
        assert k in keys, k
    assert model.state_dict()["class_embed.weight"].shape == (92, 256) and model.state_dict()["query_embed.weight"].shape == (100, 256)
    frame = aloscene.Frame(torch.rand(3, 160, 224) * 255, normalization="255").norm_resnet()
    frames = aloscene.Frame.batch_list([frame, aloscene.Frame(torch.rand(3, 128, 192) * 255).norm_resnet()])
    with torch.no_grad():
        out = model(frames)
    assert out["pred_logits"].shape == (2, 100, 92) and out["pred_boxes"].shape == (2, 100, 4)
    assert len(out["aux_outputs"]) == 5 and torch.isfinite(out["pred_logits"]).all()
    boxes = model.inference(out, threshold=0.0, background_class=-1)
    assert len(boxes) == 2 and isinstance(boxes[0], aloscene.BoundingBoxes2D) and boxes[0].shape[1] == 4


def test_detr_r50_config0_at_its_own_size():
    """BASELINE configs[0] as stated: alonet.detr.DetrR50 inference on ONE 640x480 aloscene.Frame through the PyTorch CPU path
    (reference: alonet/detr/detr_r50.py:55-75): forward + inference(), output surface of the reference."""
    from alonet.detr import DetrR50

    torch.manual_seed(0)
    model = DetrR50(num_classes=91, aux_loss=False).eval()
    frame = aloscene.Frame(torch.rand(3, 480, 640) * 255, normalization="255").norm_resnet()
    frames = aloscene.Frame.batch_list([frame])
    assert tuple(frames.shape) == (1, 3, 480, 640) and not bool(frames.mask.as_tensor().any())
    with torch.no_grad():
        out = model(frames)
        again = model(frames)
    assert out["pred_logits"].shape == (1, 100, 92) and out["pred_boxes"].shape == (1, 100, 4)
    assert torch.isfinite(out["pred_logits"]).all() and torch.equal(out["pred_logits"], again["pred_logits"])
    assert float(out["pred_boxes"].min()) >= 0.0 and float(out["pred_boxes"].max()) <= 1.0     # sigmoid boxes, xcyc-relative
    boxes = model.inference(out, threshold=0.0, background_class=-1)
    assert len(boxes) == 1 and isinstance(boxes[0], aloscene.BoundingBoxes2D) and boxes[0].shape == (100, 4)


def test_panoptic_head_blocks_match_reference(golden):
    from alonet.detr_panoptic import FPNstyleCNN, MHAttentionMap

    g = golden("g11_panoptic_nn.npz")
    att = MHAttentionMap(32, 32, 8, dropout=0.0).double().eval()
    assert not att.load_state_dict(formula_state_dict(att.state_dict())).missing_keys
    head = FPNstyleCNN(32 + 8, [48, 24, 16], 128).double().eval()
    assert not head.load_state_dict(formula_state_dict(head.state_dict())).missing_keys
    with torch.no_grad():
        w = att(t(g["q"]), t(g["k"]), mask=t(g["mask"]))
        seg = head(t(g["x"]), w, [t(g["fpn0"]), t(g["fpn1"]), t(g["fpn2"])])
    np.testing.assert_allclose(w.numpy(), g["weights"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(seg.numpy(), g["seg"], rtol=1e-8, atol=1e-9)
    assert abs(float(w[0, 0].sum()) - 1.0) < 1e-9 and float(w[1, :, :, :, 4:].abs().max()) == 0.0  # padded columns get 0


def test_panoptic_head_over_detr_r50_cpu():
    """PanopticHead wiring on the CPU-runnable detector (DetrR50): shapes, state-dict prefixes, inference products."""
    from alonet.detr import DetrR50
    from alonet.detr_panoptic import PanopticHead

    torch.manual_seed(0)
    model = PanopticHead(DetrR50(num_classes=91, aux_loss=False), fpn_list=[1024, 512, 256]).eval()
    keys = set(model.state_dict())
    assert {"detr.input_proj.weight", "bbox_attention.q_linear.weight", "mask_head.lay1.weight", "mask_head.adapter3.bias"} <= keys
    assert not any(p.requires_grad for p in model.detr.parameters())  # freeze_detr default
    frames = aloscene.Frame.batch_list([aloscene.Frame(torch.rand(3, 128, 160) * 255).norm_resnet(),
                                        aloscene.Frame(torch.rand(3, 96, 128) * 255).norm_resnet()])
    keep = [torch.zeros(100, dtype=torch.bool), torch.zeros(100, dtype=torch.bool)]
    keep[0][[3, 7, 50]] = True
    keep[1][[1]] = True
    with torch.no_grad():
        out = model(frames, filters=keep)
    assert out["pred_masks"].shape == (2, 3, 32, 40) and out["pred_logits"].shape == (2, 100, 92)
    boxes, masks = model.inference(out, filters=keep)
    assert [m.shape for m in masks] == [(3, 128, 160), (1, 128, 160)] and isinstance(masks[0], aloscene.Mask)
    assert boxes[0].shape == (3, 4) and int(masks[0].as_tensor().sum(0).max()) <= 1  # one-hot across queries


def test_finetune_variants_rehead_like_the_reference(tmp_path):
    """alonet/deformable_detr/deformable_detr_r50_finetune.py:10-136: base model with the checkpoint's 91 classes, then a new
    classification head (+1 background output under softmax, bias at the 0.01 prior) — ONE module shared by the six decoder
    layers for the plain model, six independent clones for the refinement model — and a fine-tuned checkpoint loaded on top."""
    import math

    from alonet.deformable_detr import DeformableDetrR50Finetune, DeformableDetrR50RefinementFinetune

    with pytest.raises(Exception, match="sigmoid"):
        DeformableDetrR50Finetune(num_classes=2, activation_fn="tanh", base_weights=None, device=None)
    with pytest.raises(FileNotFoundError, match="deformable-detr-r50"):   # the default base checkpoint is looked up, not downloaded
        DeformableDetrR50Finetune(num_classes=2, device=None)
    m = DeformableDetrR50Finetune(num_classes=2, base_weights=None, device=None, aux_loss=False)
    assert len(m.class_embed) == 6 and all(h is m.class_embed[0] for h in m.class_embed)
    assert m.class_embed[0].weight.shape == (2, 256) and m.background_class is None and m.activation_fn == "sigmoid"
    assert torch.allclose(m.class_embed[0].bias, torch.full((2,), -math.log(99.0)))
    sm = DeformableDetrR50Finetune(num_classes=2, activation_fn="softmax", base_weights=None, device=None, aux_loss=False)
    assert sm.class_embed[0].weight.shape == (3, 256) and sm.background_class == 2
    r = DeformableDetrR50RefinementFinetune(num_classes=5, base_weights=None, device=None, aux_loss=False)
    assert len({id(h) for h in r.class_embed}) == 6 and r.class_embed[3].weight.shape == (5, 256)
    assert r.transformer.decoder.bbox_embed is r.bbox_embed   # the decoder still refines with the model's box heads
    # a fine-tuned checkpoint (re-headed layout) loads on top of the base model
    with torch.no_grad():
        m.class_embed[0].weight.fill_(0.25)
    path = str(tmp_path / "finetuned.pth")
    torch.save({"model": m.state_dict()}, path)
    again = DeformableDetrR50Finetune(num_classes=2, base_weights=None, weights=path, device=None, aux_loss=False)
    assert torch.equal(again.class_embed[5].weight, torch.full((2, 256), 0.25))
    with pytest.raises(ValueError, match="Unknown weights"):
        DeformableDetrR50Finetune(num_classes=2, base_weights=None, weights="nonsense", device=None)
    # and the re-headed model runs: logits carry the new class count
    frames = aloscene.Frame.batch_list([aloscene.Frame(torch.rand(3, 64, 96) * 255, normalization="255").norm_resnet()])
    with torch.no_grad():
        out = m.eval()(frames, is_tracing=None)   # the reference's CPU-runnable branch (pure-torch op)
    assert out["pred_logits"].shape == (1, 300, 2)


def test_premade_detr_and_panoptic_variants_follow_the_reference():
    """alonet/detr/detr_r50_finetune.py:12-57, detr_panoptic/detr_r50_panoptic{,_finetune}.py, deformable_detr_panoptic/
    deformable_detr_r50_panoptic_finetune.py, deformable_detr/deformable_detr_r50_refinement.py: import paths, constructor
    arguments, head replacement, background ids, the GroupNorm -> BatchNorm switch of the mask head, checkpoint loading."""
    from alonet.deformable_detr.deformable_detr_r50_refinement import DeformableDetrR50Refinement as ByPath
    from alonet.deformable_detr import DeformableDetrR50Refinement
    from alonet.deformable_detr_panoptic import DeformableDetrR50PanopticFinetune
    from alonet.detr import DetrR50Finetune
    from alonet.detr_panoptic import DetrR50Panoptic, DetrR50PanopticFinetune

    assert ByPath is DeformableDetrR50Refinement
    with pytest.raises(FileNotFoundError, match="detr-r50"):
        DetrR50Finetune(num_classes=2)
    d = DetrR50Finetune(num_classes=2, base_weights=None, aux_loss=False)
    assert d.class_embed.weight.shape == (3, 256) and d.background_class == 2 and d.num_classes == 3
    assert DetrR50Finetune(num_classes=2, background_class=0, base_weights=None).background_class == 0
    p = DetrR50Panoptic(num_classes=7)
    assert p.detr.background_class == 7 and p.detr.class_embed.weight.shape == (8, 256)       # None -> the last id, not DetrR50's 91
    assert p.detr.return_dec_outputs and not any(q.requires_grad for q in p.detr.parameters())
    pf = DetrR50PanopticFinetune(num_classes=4, base_weights=None, use_bn_layers=True)
    assert pf.detr.class_embed.weight.shape == (5, 256) and pf.detr.background_class == 4
    assert all(isinstance(getattr(pf.mask_head, f"gn{i}"), torch.nn.BatchNorm2d) for i in range(1, 6))
    assert isinstance(DetrR50PanopticFinetune(num_classes=4, base_weights=None).mask_head.gn3, torch.nn.GroupNorm)
    dp = DeformableDetrR50PanopticFinetune(num_classes=3, base_weights=None, device=None)
    assert len(dp.detr.class_embed) == 6 and all(h is dp.detr.class_embed[0] for h in dp.detr.class_embed)
    assert dp.detr.class_embed[0].weight.shape == (3, 256) and dp.detr.background_class is None
    with pytest.raises(ValueError, match="Unknown weights"):
        DetrR50PanopticFinetune(num_classes=4, base_weights=None, weights="nonsense")
    # the BatchNorm variant runs (the mask head's fast paths only take GroupNorm layers)
    frames = aloscene.Frame.batch_list([aloscene.Frame(torch.rand(3, 64, 96) * 255, normalization="255").norm_resnet()])
    with torch.no_grad():
        out = pf.eval()(frames)
    assert out["pred_masks"].shape[:2] == (1, 100) and torch.isfinite(out["pred_masks"]).all()
