"""The re-stated callers (DeformableTransformer / DeformableDETR-R50 / RAFT) on the GPU, hot path on the HIP kernels."""
import numpy as np
import pytest
import torch

import aloscene
from alonet.deformable_detr import DeformableDetrR50
from alonet.raft import RAFT
from helpers import formula_state_dict
from test_models_cpu import build_g5_transformer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
t = torch.from_numpy


# fp64: the kernels agree with the oracle to 1e-11; what remains (3e-7) is the fp32 reference-point arithmetic that
# the reference itself does in float32 (deformable_transformer.py:381-396), evaluated by stock torch ops on GPU vs CPU
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 2e-6), (torch.float32, 1e-3)])
def test_deformable_transformer_on_hip_matches_reference(golden, dtype, tol):
    """North-star bar: <= 1e-3 max-abs deviation from the reference PyTorch path in fp32 (1e-9 in fp64)."""
    g = golden("g5_deformable_transformer.npz")
    tr, L = build_g5_transformer(g)
    tr = tr.to(DEV, dtype)
    srcs = [t(g[f"src{i}"]).to(DEV, dtype) for i in range(L)]
    poss = [t(g[f"pos{i}"]).to(DEV, dtype) for i in range(L)]
    masks = [t(g[f"mask{i}"]).to(DEV) for i in range(L)]
    with torch.no_grad():
        out = tr(srcs, masks, poss, t(g["query_embed"]).to(DEV, dtype))
    assert np.abs(out["hs"].double().cpu().numpy() - g["hs"]).max() <= tol
    assert np.abs(out["inter_references_out"].double().cpu().numpy() - g["inter_references_out"]).max() <= tol
    for i in range(L):
        assert np.abs(out["memory"][i].double().cpu().numpy() - g[f"memory{i}"]).max() <= tol


def test_raft_on_hip_matches_reference(golden):
    g = golden("g7_raft.npz")
    model = RAFT().eval()  # default corr_block = the HIP CorrBlock
    model.load_state_dict(formula_state_dict(model.state_dict()))
    model = model.to(DEV)
    f1 = aloscene.Frame(t(g["img1"]).float(), normalization="minmax_sym", names=("B", "C", "H", "W")).to(DEV)
    f2 = aloscene.Frame(t(g["img2"]).float(), normalization="minmax_sym", names=("B", "C", "H", "W")).to(DEV)
    with torch.no_grad():
        outs = model(f1, f2, iters=4)
    flows = np.stack([o["flow"].cpu().numpy() for o in outs])
    assert np.isfinite(g["flow"]).all() and np.isfinite(flows).all()
    assert np.abs(flows - g["flow"]).max() <= 1e-3  # 1/8-resolution flow, pixels
    assert np.abs(outs[-1]["up_flow"].cpu().numpy() - g["up_flow_last"]).max() <= 8e-3  # x8 up-sampled
    assert np.abs(outs[-1]["hidden_state"].cpu().numpy() - g["hidden_last"]).max() <= 1e-3
    flow = model.inference(outs, only_last=True)
    assert isinstance(flow, aloscene.Flow) and flow.shape == (2, 2, 128, 160)


def _frames(sizes, seed=0):
    gen = torch.Generator().manual_seed(seed)
    return [aloscene.Frame(torch.rand(3, h, w, generator=gen) * 255, normalization="255").norm_resnet() for h, w in sizes]


def test_deformable_detr_r50_end_to_end_hip_vs_torch_branch():
    """Full model, ragged batch (padding mask): HIP op vs the pure-torch export branch on the same weights, fp32."""
    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=91, aux_loss=True, device=torch.device(DEV)).eval()
    frames = aloscene.Frame.batch_list(_frames([(192, 256), (160, 224)])).to(DEV)
    with torch.no_grad():
        out = model(frames)
        ref = model(frames, is_tracing=None)
    assert out["pred_logits"].shape == (2, 300, 91) and out["pred_boxes"].shape == (2, 300, 4)
    assert len(out["aux_outputs"]) == 5 and out["activation_fn"] == "sigmoid"
    assert (out["pred_logits"] - ref["pred_logits"]).abs().max().item() <= 1e-3
    assert (out["pred_boxes"] - ref["pred_boxes"]).abs().max().item() <= 1e-3
    boxes = model.inference(out, threshold=0.0)
    assert len(boxes) == 2 and isinstance(boxes[0], aloscene.BoundingBoxes2D)
    assert boxes[0].boxes_format == "xcyc" and not boxes[0].absolute and boxes[0].shape == (300, 4)
    assert boxes[0].labels.scores.shape == (300,) and boxes[0].device.type == "cpu"
    # a list of Frames is accepted and batched by the forward decorator
    with torch.no_grad():
        out_list = model([f.to(DEV) for f in _frames([(192, 256), (160, 224)])])
    assert torch.allclose(out_list["pred_boxes"], out["pred_boxes"], atol=1e-5)


def _same_detections(a, b):
    return len(a) == len(b) and all(torch.equal(x.as_tensor(), y.as_tensor()) and torch.equal(x.labels.as_tensor(), y.labels.as_tensor())
                                    and torch.equal(x.labels.scores, y.labels.scores) for x, y in zip(a, b))


def _without_pack(out):
    """The same forward outputs under fresh tensor objects (same storage): nothing rides on them, inference() takes its chain."""
    return {k: (v.detach() if k in ("pred_logits", "pred_boxes") else v) for k, v in out.items()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_inference_from_the_packed_detections_equals_the_step_by_step_path(dtype):
    """The forward leaves (score, label, box) of its last level packed in one fp32 tensor so that inference() needs one
    device-to-host copy; the products must be those of the reference's chain (softmax / sigmoid, max, threshold,
    deformable_detr.py:508-555) on `pred_logits` / `pred_boxes` — and a caller who edits those tensors gets the chain, not the pack.
    The pack rides on the `pred_logits` tensor object: the output dictionary has the reference's keys and nothing else."""
    torch.manual_seed(1)
    model = DeformableDetrR50(num_classes=91, aux_loss=False, device=torch.device(DEV)).eval().to(dtype)
    frames = aloscene.Frame.batch_list(_frames([(192, 256), (160, 224)], seed=2)).to(DEV).to(dtype)
    with torch.no_grad():
        out = model(frames)
    assert set(out) == {"pred_logits", "pred_boxes", "activation_fn"}          # reference deformable_detr.py:385-389
    assert model._packed_detections(out["pred_logits"], out["pred_boxes"], "sigmoid") is not None
    plain = _without_pack(out)
    assert model._packed_detections(plain["pred_logits"], plain["pred_boxes"], "sigmoid") is None
    for kwargs in ({}, {"threshold": 0.0}, {"threshold": 0.05}):
        fast, slow = model.inference(out, **kwargs), model.inference(plain, **kwargs)
        assert _same_detections(fast, slow), kwargs
    assert sum(len(b) for b in model.inference(out, threshold=0.0)) == 600
    keep = [torch.arange(300, device=DEV) % 7 == i for i in range(2)]          # caller-made filters, on the device
    assert _same_detections(model.inference(out, filters=keep), model.inference(plain, filters=keep))
    # the same through a HIP graph: the pack is part of the replayed forward
    from alonet.common import GraphedForward
    graphed = GraphedForward(model)
    g_out = graphed(frames)
    assert set(g_out) == {"pred_logits", "pred_boxes", "activation_fn"}
    assert model._packed_detections(g_out["pred_logits"], g_out["pred_boxes"], "sigmoid") is not None
    assert _same_detections(model.inference(g_out, threshold=0.05), model.inference(_without_pack(g_out), threshold=0.05))
    g_out = graphed(frames)   # a replay refills the same buffers, the pack included
    assert _same_detections(model.inference(g_out), model.inference(_without_pack(g_out)))
    # edited logits (in place, or another tensor under the same key): the pack no longer describes them
    edited = dict(out)
    edited["pred_logits"] = out["pred_logits"] + 3.0
    assert _same_detections(model.inference(edited), model.inference(_without_pack(edited)))
    out["pred_logits"].add_(3.0)
    assert model._packed_detections(out["pred_logits"], out["pred_boxes"], "sigmoid") is None
    assert _same_detections(model.inference(out), model.inference(edited))


def test_forward_and_inference_under_inference_mode():
    """`torch.inference_mode()` (Lightning's default for validate / predict): its tensors have no version counter, so nothing
    may read `_version` on them (round-3 advisor finding); same detections as under `no_grad`, HIP graph replay included."""
    torch.manual_seed(1)
    model = DeformableDetrR50(num_classes=91, aux_loss=False, device=torch.device(DEV)).eval()
    frames = aloscene.Frame.batch_list(_frames([(192, 256), (160, 224)], seed=2)).to(DEV)
    def close(a, b):   # two forwards (not one forward read twice): stock GEMM / attention kernels may pick another algorithm
        return len(a) == len(b) and all(x.shape == y.shape and torch.allclose(x.as_tensor(), y.as_tensor(), atol=2e-5)
                                        and torch.equal(x.labels.as_tensor(), y.labels.as_tensor())
                                        and torch.allclose(x.labels.scores, y.labels.scores, atol=2e-5) for x, y in zip(a, b))

    with torch.no_grad():
        model(frames)                                                         # derived weights, solver choices
        want = model.inference(model(frames), threshold=0.05)
    with torch.inference_mode():
        out = model(frames)
        assert out["pred_logits"].is_inference() and set(out) == {"pred_logits", "pred_boxes", "activation_fn"}
        got = model.inference(out, threshold=0.05)
        by_filter = model.inference(out, filters=model.get_outs_filter(m_outputs=out, threshold=0.05))
    assert sum(len(b) for b in want) >= 2 and close(got, want), (got, want)
    assert _same_detections(by_filter, got)
    assert _same_detections(model.inference(out, threshold=0.05), got)        # inference tensors consumed outside the mode
    from alonet.common import GraphedForward
    with torch.inference_mode():
        g_out = GraphedForward(model)(frames)
        assert close(model.inference(g_out, threshold=0.05), want)
    # the panoptic model built on it
    from alonet.deformable_detr_panoptic import DeformableDetrR50Panoptic
    pan = DeformableDetrR50Panoptic(num_classes=91, device=torch.device(DEV)).eval()
    with torch.inference_mode():
        p_out = pan(frames, threshold=0.3)
        boxes, masks = pan.inference(p_out, threshold=0.3)
    assert len(boxes) == 2 and all(m.shape[0] == b.shape[0] for m, b in zip(masks, boxes))


def test_deformable_detr_r50_bf16_runs_and_tracks_fp32():
    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=91, aux_loss=False, device=torch.device(DEV)).eval()
    frames = aloscene.Frame.batch_list(_frames([(256, 320), (256, 320)], seed=3)).to(DEV)
    with torch.no_grad():
        ref = model(frames)
        model_bf16 = model.bfloat16()
        out = model_bf16(frames.to(torch.bfloat16))
    assert out["pred_boxes"].dtype == torch.bfloat16 and torch.isfinite(out["pred_logits"].float()).all()
    # reduced precision end to end (backbone included): boxes are sigmoids in [0,1]; a loose sanity band only
    assert (out["pred_boxes"].float() - ref["pred_boxes"]).abs().mean().item() < 0.05


def test_training_step_end_to_end_on_hip():
    """BASELINE configs[3] in miniature: forward, Hungarian matching, set loss, backward through alo_msda_backward,
    gradient clipping, AdamW with the reference's parameter groups."""
    from alonet.deformable_detr.training import build_criterion, configure_optimizers, training_step
    import alo_hip

    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=5, aux_loss=True, device=torch.device(DEV)).train()
    names = [f"c{i}" for i in range(5)]
    frs = []
    for i, (h, w) in enumerate([(192, 256), (160, 224)]):
        lab = aloscene.Labels(torch.tensor([1.0, 3.0][: i + 1]), encoding="id", labels_names=names)
        bx = aloscene.BoundingBoxes2D(torch.tensor([[0.3, 0.4, 0.2, 0.3], [0.7, 0.6, 0.2, 0.2]][: i + 1]), "xcyc", False, labels=lab)
        frs.append(aloscene.Frame(torch.rand(3, h, w) * 255, normalization="255", boxes2d=bx).norm_resnet())
    frames = aloscene.Frame.batch_list(frs).to(DEV)
    crit, opt = build_criterion(), configure_optimizers(model)
    before = model.transformer.encoder.layers[0].self_attn.value_proj.weight.detach().clone()
    with alo_hip.LaunchTimer() as t:
        loss0, parts = training_step(model, crit, opt, frames)
        loss1, _ = training_step(model, crit, opt, frames)
    tags = t.summary()
    assert any(k.startswith("msda_bwd") for k in tags) and any(k.startswith("msda_fwd/") for k in tags)
    assert torch.isfinite(loss0) and torch.isfinite(loss1)
    assert {"loss_focal_label", "loss_bbox", "loss_giou", "loss_bbox_4"} <= set(parts)
    after = model.transformer.encoder.layers[0].self_attn.value_proj.weight.detach()
    assert not torch.equal(before, after)  # gradients reached the attention's value projection through the HIP backward


def test_panoptic_head_on_deformable_detr_hip_vs_torch_branch():
    """BASELINE configs[4] in miniature: PanopticHead over Deformable-DETR R50; detector attention on the HIP op vs the
    pure-torch branch, same weights and same kept queries, then masks through ``inference``."""
    from alonet.deformable_detr_panoptic import DeformableDetrR50Panoptic

    torch.manual_seed(0)
    model = DeformableDetrR50Panoptic(num_classes=250, device=torch.device(DEV)).eval()
    assert {"detr.transformer.encoder.layers.0.self_attn.value_proj.weight", "bbox_attention.k_linear.weight",
            "mask_head.out_lay.bias"} <= set(model.state_dict())
    frames = aloscene.Frame.batch_list(_frames([(192, 256), (160, 224)], seed=5)).to(DEV)
    keep = [torch.zeros(300, dtype=torch.bool, device=DEV) for _ in range(2)]
    keep[0][[0, 17, 120, 299]] = True
    keep[1][[5, 6]] = True
    with torch.no_grad():
        out = model(frames, filters=keep)
        ref = model(frames, filters=keep, is_tracing=None)
    assert out["pred_masks"].shape == (2, 4, 48, 64)
    scale = ref["pred_masks"].abs().max().item()
    assert (out["pred_masks"] - ref["pred_masks"]).abs().max().item() <= 1e-3 * max(scale, 1.0)
    boxes, masks = model.inference(out, filters=keep)
    assert [tuple(m.shape) for m in masks] == [(4, 192, 256), (2, 192, 256)]
    assert isinstance(masks[0], aloscene.Mask) and boxes[1].shape == (2, 4)
    assert int(masks[0].as_tensor().sum(0).max()) <= 1
    # default query selection: the detector's own score filter
    with torch.no_grad():
        out2 = model(frames, threshold=0.0)
    assert out2["pred_masks"].shape[:2] == (2, 300)


def test_graphed_forward_replays_the_eager_forward_bit_for_bit():
    """alonet.common.GraphedForward: a HIP graph of DeformableDETR-R50's forward gives the eager outputs exactly, for the captured
    batch and for new batches (data AND padding mask) copied into the captured input."""
    from alonet.common import GraphedForward

    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=91, aux_loss=False, device=torch.device(DEV)).eval().to(torch.bfloat16)
    gen = torch.Generator().manual_seed(5)

    def batch(pad):
        fr = [aloscene.Frame(torch.rand(3, 256 - pad * i, 320, generator=gen) * 255, normalization="255").norm_resnet()
              for i in range(2)]
        return aloscene.Frame.batch_list(fr).to(DEV).to(torch.bfloat16)

    first, second = batch(0), batch(32)   # same padded shape (the second batch has a padded frame)
    assert first.shape == second.shape and bool(second.mask.as_tensor().any()) and not bool(first.mask.as_tensor().any())
    graphed = GraphedForward(model)
    with torch.no_grad():
        for frames in (first, second, first):
            want = model(frames)
            got = graphed(frames)
            for key in ("pred_logits", "pred_boxes"):
                assert torch.equal(got[key], want[key]), key
    assert len(graphed._graphs) == 1


def test_two_graphed_wrappers_of_one_model_do_not_invalidate_each_other():
    """Round-3 advisor finding: the cache epoch is per model but "no graph alive" was judged per wrapper, so the first capture of a
    second GraphedForward re-derived (freed) the tensors baked into the first one's live graphs, which then re-captured and
    invalidated the second in turn — a capture on every alternating call.  Both wrappers keep their graphs now."""
    from alonet.common import GraphedForward

    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=91, aux_loss=False, device=torch.device(DEV)).eval().to(torch.bfloat16)
    frames = aloscene.Frame.batch_list(_frames([(192, 256), (192, 256)], seed=4)).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        want = {k: v.clone() for k, v in model(frames).items() if k in ("pred_logits", "pred_boxes")}
    a, b = GraphedForward(model), GraphedForward(model)
    a(frames)
    graph_a = next(iter(a._graphs.values()))[1]
    b(frames)
    graph_b = next(iter(b._graphs.values()))[1]
    for _ in range(3):
        for w, g in ((a, graph_a), (b, graph_b)):
            got = w(frames)
            assert next(iter(w._graphs.values()))[1] is g          # replayed, not captured again
            for key in want:
                assert torch.equal(got[key], want[key]), key


def test_graphed_forward_two_shapes_and_weight_surgery_keep_every_graph_valid():
    """Round-2 advisor finding: capturing a second key used to drop the derived weight tensors (packed / folded / merged copies)
    the FIRST graph had baked in.  Two input shapes are captured, memory is churned, both graphs must still replay the eager
    outputs bit for bit; then the weights change through load_weights-style surgery (invalidate_caches) and every graph is
    captured again instead of replaying on freed tensors."""
    import alo_hip
    from alonet.common import GraphedForward

    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=91, aux_loss=False, device=torch.device(DEV)).eval().to(torch.bfloat16)
    gen = torch.Generator().manual_seed(7)

    def batch(h, w):
        fr = [aloscene.Frame(torch.rand(3, h, w, generator=gen) * 255, normalization="255").norm_resnet() for _ in range(2)]
        return aloscene.Frame.batch_list(fr).to(DEV).to(torch.bfloat16)

    small, large = batch(192, 256), batch(256, 320)
    graphed = GraphedForward(model)
    with torch.no_grad():
        want_small, want_large = model(small), model(large)
        want_small = {k: want_small[k].clone() for k in ("pred_logits", "pred_boxes")}
        want_large = {k: want_large[k].clone() for k in ("pred_logits", "pred_boxes")}
        graphed(small)
        graphed(large)            # second key: must not free what the first graph reads
        assert len(graphed._graphs) == 2
        junk = [torch.randn(1 << 20, device=DEV) for _ in range(64)]   # re-use whatever the allocator got back
        del junk
        for frames, want in ((small, want_small), (large, want_large), (small, want_small)):
            got = graphed(frames)
            for key in want:
                assert torch.equal(got[key], want[key]), key
        # weight surgery through .data (no version bump), the way load_weights does it
        for p_ in model.parameters():
            p_.data.mul_(1.01)
        alo_hip.invalidate_caches(model)
        want2 = {k: v.clone() for k, v in model(small).items() if k in ("pred_logits", "pred_boxes")}
        got2 = graphed(small)     # epoch changed: old graphs dropped, captured again on the new derived tensors
        assert len(graphed._graphs) == 1
        for key in want2:
            assert torch.equal(got2[key], want2[key]), key
        assert not torch.equal(want2["pred_logits"], want_small["pred_logits"])


def test_graphed_forward_replays_raft(golden):
    """Two frame inputs + keyword options through one HIP graph: RAFT's iterations replay to the eager flow."""
    from alonet.common import GraphedForward

    g = golden("g7_raft.npz")
    model = RAFT().eval()
    model.load_state_dict(formula_state_dict(model.state_dict()))
    model = model.to(DEV)
    mk = lambda a: aloscene.Frame(t(a).float(), normalization="minmax_sym", names=("B", "C", "H", "W")).to(DEV)  # noqa: E731
    f1, f2 = mk(g["img1"]), mk(g["img2"])
    graphed = GraphedForward(model)
    flows = []
    with torch.no_grad():
        for a, b in ((f1, f2), (f2, f1)):
            want = model(a, b, iters=3, only_last=True)
            got = graphed(a, b, iters=3, only_last=True)
            # fp32 MIOpen convolutions: eager and replayed launches agree to float noise, not to the bit
            assert (got[-1]["up_flow"] - want[-1]["up_flow"]).abs().max().item() <= 1e-3
            flows.append(got[-1]["up_flow"].clone())
    assert (flows[0] - flows[1]).abs().max().item() > 0.1   # the swapped pair was really copied into the captured inputs
    assert len(graphed._graphs) == 1


# ---- stated bf16 end-to-end tolerances -------------------------------------------------------------------
# measured on MI355X (round 2): G5 0.077 / G12 0.042 on hs; full model 0.039 on logits, 0.004 on boxes
BF16_TRANSFORMER_TOL = 0.1    # max-abs on hs / memory (LayerNorm-ed, O(1) values) through the bf16 layers vs the reference's fp64 outputs
BF16_MODEL_LOGIT_TOL = 0.08   # max-abs on the raw class logits of the full R50 model, bf16 vs fp32 (same weights, same frames): 2 x measured
BF16_MODEL_BOX_TOL = 0.01     # max-abs on the (sigmoid) boxes in [0, 1]: 2.5 x measured


@pytest.mark.parametrize("fixture", ["g5_deformable_transformer.npz", "g12_deformable_transformer_d256.npz"])
def test_deformable_transformer_bf16_fast_path_vs_reference_golden(golden, fixture):
    """The reference's own transformer outputs (G5: small config; G12: the DETR-family width d_model 256 / 8 heads / 4 levels /
    4 points, i.e. the shape bench.py runs: merged projections, value_proj_head_major, fused head-major MSDA, add_layernorm,
    ffn256 kernels) against the bf16 inference path, with a stated max-abs tolerance."""
    import alo_hip

    g = golden(fixture)
    tr, L = build_g5_transformer(g)
    tr = tr.to(DEV, torch.bfloat16).eval()
    bf = lambda a: t(a).to(DEV, torch.bfloat16)  # noqa: E731
    srcs, poss = [bf(g[f"src{i}"]) for i in range(L)], [bf(g[f"pos{i}"]) for i in range(L)]
    masks = [t(g[f"mask{i}"]).to(DEV) for i in range(L)]
    with alo_hip.LaunchTimer() as timer, torch.no_grad():
        out = tr(srcs, masks, poss, bf(g["query_embed"]))
    tags = timer.summary()
    assert any(k.startswith("msda_fwd_fused") for k in tags) and any(k.startswith("add_layernorm") for k in tags), tags.keys()
    if "d256" in fixture:
        assert any(k.startswith("value_proj_hm") for k in tags) and any(k.startswith("ffn256") for k in tags), tags.keys()
    errs = {"hs": np.abs(out["hs"].double().cpu().numpy() - g["hs"]).max(),
            "ref": np.abs(out["inter_references_out"].double().cpu().numpy() - g["inter_references_out"]).max()}
    for i in range(L):
        errs[f"memory{i}"] = np.abs(out["memory"][i].double().cpu().numpy() - g[f"memory{i}"]).max()
    print("bf16 fast path vs G5 golden, max-abs:", {k: float(v) for k, v in errs.items()})
    assert max(errs.values()) <= BF16_TRANSFORMER_TOL, errs


def test_deformable_detr_r50_bf16_vs_fp32_max_abs():
    """The configuration of the headline number (bf16 end to end, backbone included) against the fp32 run of the same model:
    max-abs bounds on logits and boxes, not a mean."""
    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=91, aux_loss=False, device=torch.device(DEV)).eval()
    frames = aloscene.Frame.batch_list(_frames([(384, 512), (352, 480)], seed=7)).to(DEV)
    with torch.no_grad():
        ref = model(frames)
        out = model.bfloat16()(frames.to(torch.bfloat16))
    dl = (out["pred_logits"].float() - ref["pred_logits"]).abs().max().item()
    db = (out["pred_boxes"].float() - ref["pred_boxes"]).abs().max().item()
    print("bf16 vs fp32 full model: max-abs logits", dl, "boxes", db)
    assert dl <= BF16_MODEL_LOGIT_TOL and db <= BF16_MODEL_BOX_TOL


def test_detector_runs_the_resident_forward_and_it_changes_no_bit():
    """DeformableDETR-R50, bf16 inference: the encoder's self-attention takes alo_msda_forward_fused_hm_resident (coarse pyramid levels
    in LDS) — and the model's outputs are bit-identical to a run with the plain head-major kernel forced."""
    import alo_hip

    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=91, aux_loss=False, device=torch.device(DEV)).eval().to(torch.bfloat16)
    frames = aloscene.Frame.batch_list(_frames([(512, 672), (480, 640)], seed=3)).to(DEV).to(torch.bfloat16)
    with alo_hip.LaunchTimer() as timer, torch.no_grad():
        out = model(frames)
    tags = timer.summary()
    assert any(k.startswith("msda_fwd_fused_resident") for k in tags), tags.keys()      # the encoder (Lq = S)
    assert any(k.startswith("msda_fwd_fused/Lq=300") for k in tags), tags.keys()        # the decoder stays on the plain kernel
    plain = alo_hip.msda_forward_fused_hm
    alo_hip.msda_forward_fused_hm = lambda *a, **k: plain(*a, **dict(k, resident=False))
    try:
        with alo_hip.LaunchTimer() as timer2, torch.no_grad():
            ref = model(frames)
    finally:
        alo_hip.msda_forward_fused_hm = plain
    assert not any(k.startswith("msda_fwd_fused_resident") for k in timer2.summary())
    assert torch.equal(out["pred_logits"], ref["pred_logits"]) and torch.equal(out["pred_boxes"], ref["pred_boxes"])


# ---- BASELINE configs[3] / configs[4] at their per-GPU size -------------------------------------------------------------------
def test_config4_training_step_at_per_gpu_size():
    """configs[3]: global batch 32 on 8 GPUs = 4 frames of 1333x800 per GPU, fp32: three full training steps
    (forward, Hungarian match, set loss, alo_msda_backward at Lq = S = 22223, clip, AdamW)."""
    from alonet.deformable_detr.training import build_criterion, configure_optimizers, training_step
    import alo_hip

    torch.manual_seed(0)
    model = DeformableDetrR50(num_classes=91, aux_loss=True, device=torch.device(DEV)).train()
    gen = torch.Generator().manual_seed(11)
    names = [f"class_{i}" for i in range(91)]
    frs = []
    for _ in range(4):
        lab = aloscene.Labels(torch.randint(0, 91, (10,), generator=gen).float(), encoding="id", labels_names=names)
        cxcy, wh = torch.rand(10, 2, generator=gen) * 0.6 + 0.2, torch.rand(10, 2, generator=gen) * 0.3 + 0.05
        bx = aloscene.BoundingBoxes2D(torch.cat([cxcy, wh], 1), "xcyc", False, labels=lab)
        frs.append(aloscene.Frame(torch.rand(3, 800, 1333, generator=gen) * 255, normalization="255", boxes2d=bx).norm_resnet())
    frames = aloscene.Frame.batch_list(frs).to(DEV)
    crit, opt = build_criterion(), configure_optimizers(model)
    with alo_hip.LaunchTimer(only="msda_bwd") as timer:
        losses = [training_step(model, crit, opt, frames)[0] for _ in range(3)]
    tags = timer.summary()
    assert any(k == "msda_bwd/Lq=22223" for k in tags), tags.keys()
    assert all(torch.isfinite(x) for x in losses)
    got_grad = 0
    for p in model.transformer.parameters():
        assert p.grad is None or torch.isfinite(p.grad).all()
        got_grad += int(p.grad is not None and bool(p.grad.abs().sum() > 0))
    assert got_grad >= 0.9 * sum(1 for p in model.transformer.parameters() if p.requires_grad)
    # three AdamW steps on ONE batch: the set loss must go down (what DESIGN.md section 3 states for this row)
    losses = [float(x) for x in losses]
    print("config-4 size training losses:", losses)
    assert losses[2] < losses[0], losses


def test_config5_panoptic_at_per_gpu_size():
    """configs[4]: batch 64 on 8 GPUs = 8 frames of 1333x800 per GPU, bf16, 16 kept queries per frame: forward + inference()."""
    from alonet.deformable_detr_panoptic import DeformableDetrR50Panoptic

    torch.manual_seed(0)
    model = DeformableDetrR50Panoptic(num_classes=250, device=torch.device(DEV)).eval().to(torch.bfloat16)
    model = model.to(memory_format=torch.channels_last)
    frames = aloscene.Frame.batch_list(_frames([(800, 1333)] * 8, seed=9)).to(DEV).to(torch.bfloat16)
    keep = [torch.zeros(300, dtype=torch.bool, device=DEV) for _ in range(8)]
    for k in keep:
        k[torch.arange(16, device=DEV) * 18] = True
    with torch.no_grad():
        out = model(frames, filters=keep)
        boxes, masks = model.inference(out, filters=keep)
    assert out["pred_masks"].shape[:2] == (8, 16) and torch.isfinite(out["pred_masks"].float()).all()
    assert len(masks) == 8 and tuple(masks[0].shape) == (16, 800, 1333) and int(masks[0].as_tensor().sum(0).max()) <= 1
    assert boxes[0].shape == (16, 4)


def test_bench_gpu_branch_with_two_ranks(tmp_path):
    """The N > 1 branch of bench.py ON THE GPU (rank-seeded shards, barrier + synchronize fences, max over ranks, one JSON line
    from rank 0): two ranks share the one GPU of this box (`--share-gpu`: control collectives on gloo, since RCCL wants one
    device per rank).  The whole-job value must count both ranks' frames."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # started PLAINLY, the way the driver starts it (no torch.distributed environment): bench.py spawns its own two ranks
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2",
           "--raft-steps", "1", "--raft-warmup", "1", "--raft-batch", "1", "--train-steps", "0", "--panoptic-steps", "1",
           "--eager-steps", "2", "--fp32-steps", "0", "--micro-reps", "0", "--no-cpu-baseline", "--no-pmc", "--share-gpu",
           "--detail-out", str(tmp_path / "detail.json")]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(env, OMP_NUM_THREADS="4"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 4
    assert abs(line["value"] - 2 * 2 * 3 / (line["ms_per_step"] * 3 / 1e3)) / line["value"] < 1e-3
    assert line["raft"]["value"] > 0 and line["panoptic"]["value"] > 0 and "cpu_baseline" not in line
    assert len(lines[0]) < 4096 and line["raft"]["corr_build_ms"] > 0 and line["per_rank"]["ms_per_step"]
    # the full record (kernel tables, the eager leg, per-leg configuration) is the sidecar file, not the line
    detail = json.loads((tmp_path / "detail.json").read_text())
    assert detail["value"] == line["value"] and detail["eager"]["value"] > 0 and detail["raft"]["hot_path_ms_per_step"] > 0


def test_bench_rccl_code_path_on_one_rank(tmp_path):
    """What a multi-GPU run adds to bench.py, exercised on the one GPU of this box with a ONE-rank RCCL group (`--force-dist`):
    `init_process_group("nccl", device_id=...)`, barriers and the timing all-reduce on device tensors, HIP-graph capture of the
    forward while the communicator exists, and the training step under DistributedDataParallel with RCCL buckets."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "2",
           "--no-raft", "--train-steps", "2", "--train-batch", "1", "--panoptic-steps", "0", "--no-cpu-baseline", "--no-pmc", "--force-dist",
           "--eager-steps", "0", "--fp32-steps", "2", "--micro-reps", "0", "--detail-out", str(tmp_path / "detail.json")]
    env = dict(os.environ, OMP_NUM_THREADS="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "error" not in line["train"], line["train"]
    detail = json.loads((tmp_path / "detail.json").read_text())
    assert detail["train"]["config"]["parallelism"] == "DDP over RCCL" and line["train"]["value"] > 0
    assert "hip-graph" in line["config"]["launch"], line["config"]["launch"]   # capture worked beside the communicator
    assert line["fp32"]["value"] > 0 and line["fp32"]["frac"] > 0, line["fp32"]


# ---- the reference's golden vectors ON THE DEVICE ---------------------------------------------------------------------------------
PANOPTIC_FP32_TOL = 2e-5      # relative to the largest reference entry: fp32 MIOpen / hipBLASLt kernels vs the reference's float64
PANOPTIC_BF16_ATTN_TOL = 0.02  # max-abs on the attention maps (softmax outputs <= 0.17 in G11), bf16 storage of q / k / logits
PANOPTIC_BF16_SEG_TOL = 0.12   # max-abs on the mask logits (|seg| <= 2.5 in G11): five bf16 conv + GroupNorm stages deep


@pytest.mark.parametrize("mode", ["fp32", "bf16_channels_last"])
def test_g11_panoptic_blocks_on_the_device(golden, mode):
    """G11 = the outputs of the reference's OWN MHAttentionMap / FPNstyleCNN (alonet/detr_panoptic/nn/MHAttention.py:30-47,
    nn/FPNstyle.py:49-84) on seeded inputs, fed to the DEVICE path: fp32, and the configuration BASELINE configs[4] runs —
    bf16, channels_last, the two widest convolutions on alo_conv3x3_nhwc with zero-padded channels (40 -> 64)."""
    import alo_hip
    from alonet.detr_panoptic import FPNstyleCNN, MHAttentionMap

    g = golden("g11_panoptic_nn.npz")
    att = MHAttentionMap(32, 32, 8, dropout=0.0).double().eval()
    att.load_state_dict(formula_state_dict(att.state_dict()))
    head = FPNstyleCNN(32 + 8, [48, 24, 16], 128).double().eval()
    head.load_state_dict(formula_state_dict(head.state_dict()))
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16
    att, head = att.to(DEV, dtype), head.to(DEV, dtype)
    cast = lambda a: t(a).to(DEV, dtype)  # noqa: E731
    x, fpns = cast(g["x"]), [cast(g[f"fpn{i}"]) for i in range(3)]
    if mode != "fp32":
        head = head.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
        fpns = [f.contiguous(memory_format=torch.channels_last) for f in fpns]
    with alo_hip.LaunchTimer() as timer, torch.no_grad():
        w = att(cast(g["q"]), cast(g["k"]), mask=t(g["mask"]).to(DEV))
        seg = head(x, w, fpns)
    w64, seg64 = w.double().cpu().numpy(), seg.double().cpu().numpy()
    assert w64.shape == g["weights"].shape and seg64.shape == g["seg"].shape
    ew, es = np.abs(w64 - g["weights"]).max(), np.abs(seg64 - g["seg"]).max()
    print(f"G11 on the device ({mode}): max-abs attention maps {ew:.3e}, mask logits {es:.3e}")
    assert float(np.abs(w64[1, :, :, :, 4:]).max()) == 0.0          # padded columns get exactly 0 on the device too
    if mode == "fp32":
        assert ew <= PANOPTIC_FP32_TOL * np.abs(g["weights"]).max() and es <= PANOPTIC_FP32_TOL * 10 * np.abs(g["seg"]).max()
    else:
        assert any(k.startswith("conv3x3") for k in timer.summary()), timer.summary().keys()   # the HIP convolution really ran
        assert ew <= PANOPTIC_BF16_ATTN_TOL and es <= PANOPTIC_BF16_SEG_TOL


@pytest.mark.parametrize("tag", ["detr", "deformable"])
def test_g13_criterion_and_matcher_on_cuda_tensors(golden, tag):
    """G13 (the reference's own criterion + Hungarian matchers) with predictions and targets living ON THE DEVICE, as in the
    training step: the cost matrix is built on the GPU, the assignment on the host; indices equal, every loss term to 1e-5."""
    from alonet.deformable_detr.criterion import DeformableCriterion
    from alonet.deformable_detr.matcher import DeformableDetrHungarianMatcher
    from alonet.detr.criterion import DetrCriterion
    from alonet.detr.matcher import DetrHungarianMatcher
    from test_training_cpu import _g13_frames, _g13_outputs

    g = golden("g13_criterion.npz")
    frames = _g13_frames(g).to(DEV)
    out, levels = _g13_outputs(g, tag, "softmax" if tag == "detr" else "sigmoid")
    mv = lambda d: {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}  # noqa: E731
    levels = [mv(lv) for lv in levels]
    out = dict(levels[0])
    out["aux_outputs"] = levels[1:]
    if tag == "detr":
        crit = DetrCriterion(matcher=DetrHungarianMatcher(1, 5, 2), loss_ce_weight=1, loss_boxes_weight=5, loss_giou_weight=2,
                             eos_coef=0.1, aux_loss_stage=len(levels), losses=["labels", "boxes"])
    else:
        crit = DeformableCriterion(matcher=DeformableDetrHungarianMatcher(1, 5, 2), loss_label_weight=1, loss_boxes_weight=5,
                                   loss_giou_weight=2, eos_coef=0.1, aux_loss_stage=len(levels), losses=["labels", "boxes"],
                                   focal_alpha=0.25)
    crit = crit.to(DEV)
    for s_, lvl in enumerate(levels):
        for bi, (pi, ti) in enumerate(crit.matcher(lvl, frames)):
            want = g[f"{tag}.match{s_}.{bi}"]
            assert pi.tolist() == want[0].tolist() and ti.tolist() == want[1].tolist(), (s_, bi)
    total, parts = crit(out, frames)
    assert total.is_cuda
    assert abs(float(total) - float(g[f"{tag}.total"])) <= 1e-5 * abs(float(g[f"{tag}.total"]))
    ref_parts = {k[len(tag) + 6:]: float(g[k]) for k in g.files if k.startswith(f"{tag}.part.")}
    assert set(ref_parts) == set(parts)
    for k, v in ref_parts.items():
        assert abs(float(parts[k]) - v) <= 1e-5 * max(1.0, abs(v)), (k, float(parts[k]), v)


def test_config3_raft_32_iterations_batch4_720p():
    """BASELINE configs[2] as bench.py runs it — RAFT, 32 iterations, 4 pairs of 1280 x 720, fp32 — with a check: finite flows
    of the right shape, and pair 2 of the batch equals the same pair run alone to <= 1e-3 px at 1/8 resolution (every kernel is
    independent across the batch index: the correlation build's per-item scaling and block mapping, the lookup's slab offsets
    and MIOpen's batched convolutions).  Reference call pattern: alonet/raft/raft.py:157-193."""
    torch.manual_seed(0)
    model = RAFT().eval().to(DEV)
    gen = torch.Generator().manual_seed(4321)
    f1 = torch.rand(4, 3, 720, 1280, generator=gen) * 2 - 1
    f2 = torch.roll(f1, shifts=(3, -5), dims=(2, 3)) + 0.01 * torch.randn(f1.shape, generator=gen)
    mk = lambda x: aloscene.Frame(x, normalization="minmax_sym", names=("B", "C", "H", "W")).to(DEV)  # noqa: E731
    with torch.no_grad():
        outs = model(mk(f1), mk(f2), iters=32, only_last=True)
        flow4, up4 = outs[-1]["flow"].clone(), outs[-1]["up_flow"].clone()
        solo = model(mk(f1[2:3]), mk(f2[2:3]), iters=32, only_last=True)
    assert flow4.shape == (4, 2, 90, 160) and up4.shape == (4, 2, 720, 1280)
    assert torch.isfinite(flow4).all() and torch.isfinite(up4).all()
    d = (flow4[2] - solo[-1]["flow"][0]).abs().max().item()
    du = (up4[2] - solo[-1]["up_flow"][0]).abs().max().item()
    print("RAFT 32 iters, batch of 4 vs solo pair: max-abs flow", d, "up_flow", du)
    assert d <= 1e-3 and du <= 8e-3     # up_flow = 8 x the 1/8-resolution flow
    flows = model.inference(outs, only_last=True)
    assert isinstance(flows, aloscene.Flow) and tuple(flows.shape) == (4, 2, 720, 1280)


# ---- G14 / G14b / G15: the reference's own DeformableDETR / PanopticHead outputs, through the HIP op -----------------------------
def _golden_builders():
    import test_models_golden_cpu as M

    return M


@pytest.mark.parametrize("tag", ["plain", "refine", "softmax"])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 5e-6), (torch.float32, 1e-3)])
def test_g14_deformable_detr_on_hip_matches_the_reference_model(golden, tag, dtype, tol):
    """The reference's DeformableDETR (forward, heads with / without box refinement, softmax / sigmoid, aux + dec + enc + backbone
    outputs, inference() at two thresholds) vs this repository's class with the deformable attention on the HIP kernels.
    North-star bar: <= 1e-3 max-abs in fp32."""
    M = _golden_builders()
    g = golden("g14_deformable_detr.npz")
    model = M.build_g14(tag).to(DEV, dtype)
    frames = M.batch_from_raw(g, dtype=dtype).to(DEV)
    with torch.no_grad():
        out = model(frames)
    assert M.check_forward(out, g, tag, tol) >= 20
    if dtype == torch.float64:      # the kept sets are defined by thresholds on scores: compare them where rounding cannot flip one
        M.check_inference(model, out, g, tag, tol)


@pytest.mark.parametrize("dtype,tol_logits,tol_boxes,channels_last", [(torch.float32, 1e-3, 1e-3, False), (torch.bfloat16, 0.08, 0.01, False),
                                                                      (torch.bfloat16, 0.08, 0.01, True)])
def test_g14b_deformable_detr_d256_on_hip_matches_the_reference_model(golden, dtype, tol_logits, tol_boxes, channels_last):
    """DETR-family width (d_model 256, 8 heads x 32 channels, 4 levels x 4 points): in bf16 this is the configuration of the
    headline number — lazily fused positional encodings, projections normalised straight into the flattened source, the mask
    pyramid kernel, merged projections, head-major fused MSDA, add_layernorm, ffn256 — against the REFERENCE model's outputs.
    bf16 bars as stated in DESIGN.md section 3 (logits 0.08, boxes 0.01: twice what was measured); fp32 at the north-star 1e-3."""
    M = _golden_builders()
    g = golden("g14b_deformable_detr_d256.npz")
    model = M.build_g14b().to(DEV, dtype)
    frames = M.batch_from_raw(g, dtype=torch.float32).to(DEV).to(dtype)
    if channels_last:
        # the layout bench.py's backbone hands over: the projections then run as GEMMs / the implicit-GEMM kernel over NHWC rows,
        # GroupNorm writes straight into the flattened encoder source, the level masks come from the mask-pyramid kernel
        model = model.to(memory_format=torch.channels_last)
    import alo_hip
    with torch.no_grad(), alo_hip.LaunchTimer() as timer:
        out = model(frames)
    launched = set(k.split("/")[0] for k in timer.summary())
    if dtype == torch.bfloat16:
        assert {"msda_fwd_fused", "add_layernorm", "ffn256", "pos_sine_flat"} <= launched, launched
    if channels_last:
        assert {"groupnorm_rows", "mask_pyramid", "conv3x3"} <= launched, launched      # the flat-source path of the headline configuration
    levels = [out] + out["aux_outputs"]
    wants = [("d256.pred_logits", "d256.pred_boxes")] + [(f"d256.aux{i}.pred_logits", f"d256.aux{i}.pred_boxes")
                                                         for i in range(len(out["aux_outputs"]))]
    for lvl, (kl, kb) in zip(levels, wants):
        assert np.abs(lvl["pred_logits"].double().cpu().numpy() - g[kl]).max() <= tol_logits, kl
        assert np.abs(lvl["pred_boxes"].double().cpu().numpy() - g[kb]).max() <= tol_boxes, kb
    thr = float(g["d256.thresholds"][0])
    boxes = model.inference(out, threshold=thr)
    for b, bx in enumerate(boxes):
        want_scores = g[f"d256.inf{thr}.scores{b}"]
        margin = 4 * tol_logits                                  # a score this close to the threshold may fall either side
        scores = out["pred_logits"][b].float().sigmoid().max(-1)[0].cpu().numpy()
        sure = int((scores > thr + margin).sum())
        assert sure <= bx.shape[0] <= int((scores > thr - margin).sum())
        assert abs(bx.shape[0] - len(want_scores)) <= int((np.abs(scores - thr) <= margin).sum())


def test_g15_panoptic_head_over_deformable_detr_on_hip_matches_the_reference(golden):
    """BASELINE configs[4]'s composition (PanopticHead over DeformableDETR) against the reference's, attention on the HIP op, fp32."""
    M = _golden_builders()
    g = golden("g15_detr_panoptic.npz")
    head = M.build_g15_panoptic(M.build_g15_deformable()).to(DEV, torch.float32)
    frames = M.batch_from_raw(g, dtype=torch.float32).to(DEV)
    thr = float(g["pan_deformable.threshold"])
    with torch.no_grad():
        out = head(frames, threshold=thr)
    for key in ("pred_logits", "pred_boxes"):
        assert np.abs(out[key].double().cpu().numpy() - g[f"pan_deformable.{key}"]).max() <= 1e-3, key
    scores = torch.from_numpy(g["pan_deformable.pred_logits"]).sigmoid().max(-1)[0].numpy()
    if np.abs(scores - thr).min() > 1e-3:                       # no score within fp32 noise of the threshold: same kept queries
        for b, flt in enumerate(out["pred_masks_info"]["filters"]):
            np.testing.assert_array_equal(flt.cpu().numpy(), g[f"pan_deformable.filter{b}"])
        want = g["pan_deformable.pred_masks"]
        assert np.abs(out["pred_masks"].double().cpu().numpy() - want).max() <= 1e-3 * max(1.0, np.abs(want).max())
        boxes, masks = head.inference(out, maskth=0.5, threshold=thr)
        for b, (bx, mk) in enumerate(zip(boxes, masks)):
            assert np.abs(bx.as_tensor().double().cpu().numpy() - g[f"pan_deformable.inf.boxes{b}"]).max() <= 1e-3
            M.assert_masks_equal_up_to_ties(mk, g, "pan_deformable", b, gap=2e-3)


def test_get_mask_queries_single_transfer_form_equals_the_per_image_form():
    """detr_panoptic/utils.py: on the device the kept decoder rows are gathered after ONE host transfer of the filters; rows, zero
    padding and the returned filters must be those of the reference's per-image form (utils.py:7-50), NaN in a query that is not
    kept included."""
    from alonet.detr_panoptic.utils import get_mask_queries

    gen = torch.Generator(device=DEV).manual_seed(3)
    dec = torch.randn(2, 4, 9, 16, generator=gen, device=DEV)          # (stages, B, Q, C)
    dec[-1, 0, 0] = float("nan")                                       # query 0 of image 0 is not kept below
    filters = [torch.tensor([0, 1, 1, 0, 0, 1, 0, 0, 1], dtype=torch.bool, device=DEV),
               torch.zeros(9, dtype=torch.bool, device=DEV),
               torch.ones(9, dtype=torch.bool, device=DEV),
               torch.tensor([1, 0, 0, 0, 0, 0, 0, 0, 0], dtype=torch.bool, device=DEV)]
    rows, kept = get_mask_queries(frames=None, m_outputs={"dec_outputs": dec}, model=None, filters=filters)
    assert kept is filters and rows.shape == (4, 9, 16)
    last = dec[-1]
    for b, f in enumerate(filters):
        n = int(f.sum())
        assert torch.equal(rows[b, :n], last[b, f]) and bool((rows[b, n:] == 0).all())
    none_kept = [torch.zeros(9, dtype=torch.bool, device=DEV) for _ in range(4)]
    rows0, _ = get_mask_queries(frames=None, m_outputs={"dec_outputs": dec}, model=None, filters=none_kept)
    assert rows0.shape == (4, 0, 16)


def test_finetune_variants_run_on_the_hip_path():
    """The re-headed pre-made models of round 5 (alonet/deformable_detr/deformable_detr_r50_finetune.py and its panoptic
    counterpart, reference lines in their docstrings) through the HIP op: new class counts in the logits, the bf16 fast path sees
    the NEW head (its merged / packed copies are invalidated on re-heading), and the mask head with BatchNorm layers (the fast
    GroupNorm kernel must not be handed a BatchNorm) gives finite masks."""
    from alonet.deformable_detr import DeformableDetrR50Finetune
    from alonet.deformable_detr_panoptic import DeformableDetrR50PanopticFinetune

    torch.manual_seed(3)
    frames = aloscene.Frame.batch_list([aloscene.Frame(torch.rand(3, 256, 320) * 255, normalization="255").norm_resnet()
                                        for _ in range(2)]).to(DEV)
    m = DeformableDetrR50Finetune(num_classes=5, base_weights=None, aux_loss=False, device=torch.device(DEV)).eval()
    with torch.no_grad():
        out32 = m(frames)
        hip = out32["pred_logits"]
        ref = m(frames, is_tracing=None)["pred_logits"]          # the reference's pure-torch branch, same weights
        out16 = m.to(torch.bfloat16)(frames.to(torch.bfloat16))
    assert hip.shape == (2, 300, 5) and (hip - ref).abs().max().item() <= 1e-3
    assert out16["pred_logits"].shape == (2, 300, 5) and (out16["pred_logits"].float() - hip).abs().max().item() <= 0.08
    assert len(m.inference(out32, threshold=0.0)) == 2
    p = DeformableDetrR50PanopticFinetune(num_classes=4, base_weights=None, use_bn_layers=True, device=torch.device(DEV)).eval()
    with torch.no_grad():
        po = p.to(torch.bfloat16).to(memory_format=torch.channels_last)(frames.to(torch.bfloat16))
    assert po["pred_logits"].shape == (2, 300, 4) and torch.isfinite(po["pred_masks"].float()).all()
