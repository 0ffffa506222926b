#!/usr/bin/env python
"""Generate the golden input/output vectors under tests/golden/ from the *reference itself*.

Runs ONLY in the build container (needs /root/reference, CPU torch).  Nothing here travels to the
GPU box except the .npz files it writes: they hold inputs and the reference's outputs, no source.

The reference's packages cannot be imported wholesale (torchvision / pytorch_lightning / cv2 are
absent), so the hot-path modules are imported one by one underneath empty package shells, which
keeps the heavy ``__init__.py`` files from executing.

Fixtures written (float64 or float32 numpy arrays, about 5 MB in total):

  g1_msda_optest.npz     reference ops/test.py case (seed 3; N1 M2 D2 Lq2 L2 P2; shapes (6,4),(3,2))
                         fwd in fp64 and fp32 through ms_deform_attn_core_pytorch
  g2_msda_grad_D{30,32,64,71}.npz   same generator, D in the reference's gradcheck set; fwd + autograd
                         grads (value / loc / attn) in fp64
  g3_msda_medium.npz     N2 M8 D32 Lq64 L4 P4, loc in [-0.25,1.25] (border + out-of-range), fwd+bwd fp64
  g4_msda_module.npz     MSDeformAttn module (d_model 64, 4 heads): state-dict + inputs + outputs for 2-d
                         and 4-d reference points with a padding mask
  g6_corr.npz            CorrBlock: fmaps (1,256,16,20) -> 4 pyramid levels + lookups (in-range, integer and
                         far out-of-frame coords); plus an odd-sized batched case (2,32,17,18), radius 3
  g5_deformable_transformer.npz   DeformableTransformer (2 enc + 2 dec layers, d_model 64) on padded inputs, fp64:
                         hs, inter_references_out, memory per level.  Weights come from helpers.formula_state_dict.
  g7_raft.npz            full RAFT forward on a 128x160 pair (images stored as fp16, used as such), 4 iterations, fp32: per-iteration flow, up_flow
  g8_known_answers.npz   hand-checkable micro cases (pixel-centre sample, corner sample)
  g10_detr_transformer.npz   the reference's vanilla DETR Transformer (2 enc + 2 dec layers, d_model 64), fp64
  g11_panoptic_nn.npz    MHAttentionMap + FPNstyleCNN of the reference's PanopticHead (fp64)
  g12_deformable_transformer_d256.npz   DeformableTransformer at the DETR-family width (d_model 256, 8 heads, 4 levels,
                         4 points, 1 + 1 layers), fp64 run stored as fp32: pins the bf16 inference fast path
  g13_criterion.npz      (make_golden_criterion.py) the reference's DetrCriterion / DeformableCriterion + Hungarian matchers on a
                         seeded batch: matched indices per decoder level, every loss term, totals, monitoring metrics
  g9_posenc.npz          PositionEmbeddingSine on a partly padded map (centred and default variants)
  g17_msda_trt_plugin_case.npz   the case the reference's TensorRT-plugin test feeds the same kernel (torch2trt/plugins/ms_deform_im2col/
                         test.py:103-121: N, M, D = 1, 8, 32; Lq = 12000; levels 64^2 .. 8^2): seed + sha256 of the 24 MB of inputs,
                         the reference's outputs for every 32nd query + the last 16 (float32 and float64)
  g18_corr_grad.npz      the reference's CorrBlock under autograd: gradients w.r.t. both feature maps through three lookups, and
                         w.r.t. attached coordinates
  g14 / g14b / g15 / g16 (make_golden_models.py) the reference's DeformableDETR / Detr / PanopticHead forward + inference() over a
                         stub convolution pyramid, and the real aloscene.Frame's norm_* / batch_list

Usage:  python tests/golden/make_golden.py            (from the repo root)
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _shell(name, path=None, **attrs):
    mod = types.ModuleType(name)
    if path is not None:
        mod.__path__ = [path]
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def load_reference():
    """Import the reference's pure-torch hot-path modules; returns a namespace of them."""
    root = _shell("alonet", REF + "/alonet", ALONET_ROOT=REF + "/alonet")
    for sub in (
        "deformable_detr",
        "deformable_detr/ops",
        "deformable_detr/ops/functions",
        "deformable_detr/ops/modules",
        "raft",
        "raft/utils",
        "common",
        "transformers",
    ):
        _shell("alonet." + sub.replace("/", "."), REF + "/alonet/" + sub)
    root.common = sys.modules["alonet.common"]

    class _Frame:  # the five attributes RAFTBase.forward touches
        def __init__(self, t, normalization):
            self.t, self.normalization, self.shape = t, normalization, t.shape

        def as_tensor(self):
            return self.t

    _shell("aloscene", None, Frame=_Frame, Flow=lambda x, names=None: x)

    ns = types.SimpleNamespace()
    ns.func = importlib.import_module("alonet.deformable_detr.ops.functions.ms_deform_attn_func")
    fpkg = sys.modules["alonet.deformable_detr.ops.functions"]
    fpkg.MSDeformAttnFunction = ns.func.MSDeformAttnFunction
    fpkg.load_MultiScaleDeformableAttention = lambda: None
    ns.mod = importlib.import_module("alonet.deformable_detr.ops.modules.ms_deform_attn")
    sys.modules["alonet.deformable_detr.ops.modules"].MSDeformAttn = ns.mod.MSDeformAttn
    ns.corr = importlib.import_module("alonet.raft.corr")
    ns.rutils = importlib.import_module("alonet.raft.utils.utils")
    ns.Frame = _Frame
    return ns


def _np(t):
    return t.detach().cpu().numpy()


def _level_start(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1])).to(torch.int32)


def _optest_inputs(N, M, D, Lq, L, P, shapes):
    """Input generator of the reference's ops/test.py (same call order after the seed)."""
    S = int(shapes.prod(1).sum())
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, loc, attn


def g1(ref):
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.int32)
    torch.manual_seed(3)
    value, loc, attn = _optest_inputs(N, M, D, Lq, L, P, shapes)
    out64 = ref.func.ms_deform_attn_core_pytorch(value.double(), shapes, loc.double(), attn.double())
    value2, loc2, attn2 = _optest_inputs(N, M, D, Lq, L, P, shapes)  # second draw, as check_..._float does
    out32 = ref.func.ms_deform_attn_core_pytorch(value2, shapes, loc2, attn2)
    np.savez_compressed(
        os.path.join(OUT, "g1_msda_optest.npz"),
        shapes=_np(shapes), level_start=_np(_level_start(shapes)),
        value=_np(value), loc=_np(loc), attn=_np(attn), out_f64=_np(out64),
        value_b=_np(value2), loc_b=_np(loc2), attn_b=_np(attn2), out_f32=_np(out32),
    )


def _fwd_bwd_f64(ref, value, shapes, loc, attn, gout):
    value = value.double().requires_grad_(True)
    loc = loc.double().requires_grad_(True)
    attn = attn.double().requires_grad_(True)
    out = ref.func.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
    out.backward(gout.double())
    return out, value.grad, loc.grad, attn.grad


def g2(ref):
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.int32)
    for D in (30, 32, 64, 71):
        torch.manual_seed(3 + D)
        value, loc, attn = _optest_inputs(N, M, D, Lq, L, P, shapes)
        gout = torch.randn(N, Lq, M * D)
        out, gv, gl, ga = _fwd_bwd_f64(ref, value, shapes, loc, attn, gout)
        np.savez_compressed(
            os.path.join(OUT, f"g2_msda_grad_D{D}.npz"),
            shapes=_np(shapes), level_start=_np(_level_start(shapes)),
            value=_np(value), loc=_np(loc), attn=_np(attn), grad_out=_np(gout),
            out=_np(out), grad_value=_np(gv), grad_loc=_np(gl), grad_attn=_np(ga),
        )


def g3(ref):
    N, M, D, Lq, L, P = 2, 8, 32, 64, 4, 4
    shapes = torch.as_tensor([(16, 21), (8, 11), (4, 6), (2, 3)], dtype=torch.int32)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(33)
    value = torch.randn(N, S, M, D)
    loc = torch.rand(N, Lq, M, L, P, 2) * 1.5 - 0.25
    # a few exact special positions: pixel centres, the -1 / H boundaries of the validity test
    loc[0, 0, 0, 0, 0] = torch.tensor([0.5 / 21, 0.5 / 16])  # centre of pixel (0,0) of level 0
    loc[0, 0, 0, 0, 1] = torch.tensor([0.0, 0.0])  # corner of the map
    loc[0, 0, 0, 0, 2] = torch.tensor([1.0, 1.0])
    loc[0, 0, 0, 0, 3] = torch.tensor([-0.5 / 21, 0.25])  # w_im == -1 exactly -> skipped
    loc[0, 1, 0, 0, 0] = torch.tensor([1.0 + 0.5 / 21, 0.25])  # w_im == W exactly -> skipped
    attn = torch.softmax(torch.randn(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    gout = torch.randn(N, Lq, M * D)
    out, gv, gl, ga = _fwd_bwd_f64(ref, value, shapes, loc, attn, gout)
    np.savez_compressed(
        os.path.join(OUT, "g3_msda_medium.npz"),
        shapes=_np(shapes), level_start=_np(_level_start(shapes)),
        value=_np(value), loc=_np(loc), attn=_np(attn), grad_out=_np(gout),
        out=_np(out).astype(np.float64), grad_value=_np(gv).astype(np.float32),
        grad_loc=_np(gl).astype(np.float32), grad_attn=_np(ga).astype(np.float32),
    )


def g4(ref):
    d_model, n_levels, n_heads, n_points = 64, 3, 4, 2
    shapes = torch.as_tensor([(7, 9), (4, 5), (2, 3)], dtype=torch.int64)
    S = int(shapes.prod(1).sum())
    N, Lq = 2, 11
    torch.manual_seed(44)
    m = ref.mod.MSDeformAttn(d_model, n_levels, n_heads, n_points).double()
    with torch.no_grad():  # the default init zeroes two of the four linears: perturb them
        m.sampling_offsets.weight.normal_(0, 0.05)
        m.attention_weights.weight.normal_(0, 0.3)
        m.attention_weights.bias.normal_(0, 0.3)
    query = torch.randn(N, Lq, d_model).double()
    src = torch.randn(N, S, d_model).double()
    ref2 = torch.rand(N, Lq, n_levels, 2).double()
    ref4 = torch.cat([ref2, torch.rand(N, Lq, n_levels, 2).double() * 0.5], -1)
    mask = torch.zeros(N, S, dtype=torch.bool)
    mask[1, -7:] = True
    mask[0, 5:9] = True
    lsi = _level_start(shapes.to(torch.int32))
    with torch.no_grad():
        out2 = m(query, ref2, src, shapes, lsi, mask, is_tracing=None)
        out4 = m(query, ref4, src, shapes, lsi, mask, is_tracing=None)
        out2_nomask = m(query, ref2, src, shapes, lsi, None, is_tracing=None)
    sd = {"sd." + k: _np(v) for k, v in m.state_dict().items()}
    np.savez_compressed(
        os.path.join(OUT, "g4_msda_module.npz"),
        cfg=np.array([d_model, n_levels, n_heads, n_points]), shapes=_np(shapes), level_start=_np(lsi),
        query=_np(query), src=_np(src), ref2=_np(ref2), ref4=_np(ref4), mask=_np(mask),
        out2=_np(out2), out4=_np(out4), out2_nomask=_np(out2_nomask), **sd,
    )


def g6(ref):
    # every pyramid level must keep h,w >= 2: with h == 1 the reference divides by (h-1) == 0 and
    # returns NaN (utils.py:8-9) -- a degenerate case we do not pin.
    B, C, H, W = 1, 256, 16, 20
    torch.manual_seed(66)
    f1 = torch.randn(B, C, H, W)
    f2 = torch.randn(B, C, H, W)
    blk = ref.corr.CorrBlock(f1, f2, radius=4)
    grid = ref.rutils.coords_grid(B, H, W)
    coords_a = grid + torch.randn(B, 2, H, W) * 2.5  # typical flow magnitudes at 1/8 res
    coords_b = grid.clone()  # first RAFT iteration: exact integer coordinates
    coords_c = grid + torch.randn(B, 2, H, W) * 40.0  # mostly out of frame
    out = {k: _np(blk(c)) for k, c in (("a", coords_a), ("b", coords_b), ("c", coords_c))}
    assert all(np.isfinite(v).all() for v in out.values())
    # an odd-sized, batched map exercises avg_pool2d's floor (dropped last row/col), rows that are not
    # 16-byte aligned, and a different radius
    f1o = torch.randn(2, 32, 17, 18)
    f2o = torch.randn(2, 32, 17, 18)
    blko = ref.corr.CorrBlock(f1o, f2o, radius=3)
    coords_o = ref.rutils.coords_grid(2, 17, 18) + torch.randn(2, 2, 17, 18) * 2.0
    out_o = _np(blko(coords_o))
    assert np.isfinite(out_o).all()
    np.savez_compressed(
        os.path.join(OUT, "g6_corr.npz"),
        f1=_np(f1), f2=_np(f2),
        lvl0=_np(blk.corr_pyramid[0]), lvl1=_np(blk.corr_pyramid[1]),
        lvl2=_np(blk.corr_pyramid[2]), lvl3=_np(blk.corr_pyramid[3]),
        coords_a=_np(coords_a), coords_b=_np(coords_b), coords_c=_np(coords_c),
        out_a=out["a"], out_b=out["b"], out_c=out["c"],
        f1o=_np(f1o), f2o=_np(f2o), coords_o=_np(coords_o), out_o=out_o,
        lvl0o=_np(blko.corr_pyramid[0]), lvl1o=_np(blko.corr_pyramid[1]),
        lvl2o=_np(blko.corr_pyramid[2]), lvl3o=_np(blko.corr_pyramid[3]),
    )


def g8(ref):
    """Known-answer micro cases; expected values are also derivable by hand (see tests)."""
    shapes = torch.as_tensor([(2, 3)], dtype=torch.int32)
    value = torch.arange(1.0, 7.0, dtype=torch.float64).view(1, 6, 1, 1)  # pixel (y,x) holds 1 + 3y + x
    loc = torch.tensor(
        [
            [(0.5 / 3, 0.5 / 2)],  # centre of pixel (0,0)  -> 1
            [(2.5 / 3, 1.5 / 2)],  # centre of pixel (1,2)  -> 6
            [(0.0, 0.0)],  # map corner: 1/4 of pixel (0,0) -> 0.25
            [(1.0 / 3, 0.5)],  # between four pixels (0,0),(0,1),(1,0),(1,1) -> (1+2+4+5)/4 = 3
        ],
        dtype=torch.float64,
    ).view(1, 4, 1, 1, 1, 2)
    attn = torch.ones(1, 4, 1, 1, 1, dtype=torch.float64)
    out = ref.func.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
    np.savez_compressed(
        os.path.join(OUT, "g8_known_answers.npz"),
        shapes=_np(shapes), level_start=np.zeros(1, np.int32), value=_np(value), loc=_np(loc), attn=_np(attn),
        out=_np(out), expected=np.array([1.0, 6.0, 0.25, 3.0]),
    )


def g5(ref):
    """DeformableTransformer (small config) through the reference's CPU/tracing branch, fp64."""
    sys.path.insert(0, os.path.dirname(OUT))
    from helpers import formula_state_dict

    DT = importlib.import_module("alonet.deformable_detr.deformable_transformer")
    torch.manual_seed(55)
    d_model, nhead, L = 64, 4, 3
    tr = DT.DeformableTransformer(d_model=d_model, nhead=nhead, num_encoder_layers=2, num_decoder_layers=2,
                                  dim_feedforward=96, dropout=0.0, return_intermediate_dec=True,
                                  num_feature_levels=L, dec_n_points=2, enc_n_points=3).double().eval()
    tr.load_state_dict(formula_state_dict(tr.state_dict()))
    B, sizes = 2, [(9, 12), (5, 6), (3, 3)]
    srcs = [torch.randn(B, d_model, h, w, dtype=torch.float64) for h, w in sizes]
    poss = [torch.randn(B, d_model, h, w, dtype=torch.float64) * 0.5 for h, w in sizes]
    masks = []
    for h, w in sizes:  # image 1 is padded on its right/bottom quarter
        m = torch.zeros(B, h, w, dtype=torch.bool)
        m[1, :, w - max(1, w // 4):] = True
        m[1, h - max(1, h // 4):, :] = True
        masks.append(m)
    query_embed = torch.randn(10, 2 * d_model, dtype=torch.float64)
    with torch.no_grad():
        out = tr(srcs, masks, poss, query_embed, is_tracing=None)
    save = dict(cfg=np.array([d_model, nhead, 2, 2, 96, L, 2, 3]), query_embed=_np(query_embed),
                hs=_np(out["hs"]), inter_references_out=_np(out["inter_references_out"]),
                init_reference_out=_np(out["init_reference_out"]))
    for i in range(L):
        save[f"src{i}"], save[f"pos{i}"], save[f"mask{i}"] = _np(srcs[i]), _np(poss[i]), _np(masks[i])
        save[f"memory{i}"] = _np(out["memory"][i])
    np.savez_compressed(os.path.join(OUT, "g5_deformable_transformer.npz"), **save)


def g12(ref):
    """DeformableTransformer at the DETR-family width (d_model 256, 8 heads of 32 channels, 4 levels, 4 points; 1 encoder +
    1 decoder layer, ffn 1024): the shape the bf16 inference fast path of this repository is specialised for.  Inputs are
    float32-representable (stored as float32), the reference runs them in fp64."""
    sys.path.insert(0, os.path.dirname(OUT))
    from helpers import formula_state_dict

    DT = importlib.import_module("alonet.deformable_detr.deformable_transformer")
    torch.manual_seed(1212)
    d_model, nhead, L = 256, 8, 4
    tr = DT.DeformableTransformer(d_model=d_model, nhead=nhead, num_encoder_layers=1, num_decoder_layers=1,
                                  dim_feedforward=1024, dropout=0.0, return_intermediate_dec=True,
                                  num_feature_levels=L, dec_n_points=4, enc_n_points=4).double().eval()
    tr.load_state_dict(formula_state_dict(tr.state_dict()))
    B, sizes = 2, [(12, 16), (6, 8), (3, 4), (2, 2)]
    srcs = [torch.randn(B, d_model, h, w).double() for h, w in sizes]
    poss = [(torch.randn(B, d_model, h, w) * 0.5).double() for h, w in sizes]
    masks = []
    for h, w in sizes:  # image 1 is padded on its right/bottom quarter
        m = torch.zeros(B, h, w, dtype=torch.bool)
        m[1, :, w - max(1, w // 4):] = True
        m[1, h - max(1, h // 4):, :] = True
        masks.append(m)
    query_embed = torch.randn(20, 2 * d_model).double()
    with torch.no_grad():
        out = tr(srcs, masks, poss, query_embed, is_tracing=None)
    f32 = lambda x: _np(x).astype(np.float32)  # noqa: E731
    save = dict(cfg=np.array([d_model, nhead, 1, 1, 1024, L, 4, 4]), query_embed=f32(query_embed),
                hs=f32(out["hs"]), inter_references_out=f32(out["inter_references_out"]))
    for i in range(L):
        save[f"src{i}"], save[f"pos{i}"], save[f"mask{i}"] = f32(srcs[i]), f32(poss[i]), _np(masks[i])
        save[f"memory{i}"] = f32(out["memory"][i])
    np.savez_compressed(os.path.join(OUT, "g12_deformable_transformer_d256.npz"), **save)


def g7(ref):
    """Full RAFT forward (reference model, formula weights), 128x160 pair, 4 iterations, fp32 on CPU.

    The frame must be at least 128 px on both sides: below that the top pyramid level is 1 pixel wide/high and the
    reference's own lookup divides by (size - 1) = 0 and returns NaN flow."""
    sys.path.insert(0, os.path.dirname(OUT))
    from helpers import formula_state_dict

    R = importlib.import_module("alonet.raft.raft")
    torch.manual_seed(77)
    model = R.RAFT().eval()
    model.load_state_dict(formula_state_dict(model.state_dict()))
    img1 = torch.rand(2, 3, 128, 160) * 2 - 1
    img2 = torch.roll(img1, shifts=(2, -3), dims=(2, 3)) + 0.01 * torch.randn(2, 3, 128, 160)
    with torch.no_grad():
        img1, img2 = img1.half().float(), img2.half().float()  # fp16-representable inputs: half the fixture size
        outs = model(ref.Frame(img1, "minmax_sym"), ref.Frame(img2, "minmax_sym"), iters=4, only_last=False)
    assert all(torch.isfinite(o["flow"]).all() for o in outs)
    np.savez_compressed(
        os.path.join(OUT, "g7_raft.npz"), img1=_np(img1.half()), img2=_np(img2.half()),
        flow=np.stack([_np(o["flow"]) for o in outs]), up_flow_last=_np(outs[-1]["up_flow"]),
        up_flow_first=_np(outs[0]["up_flow"]), hidden_last=_np(outs[-1]["hidden_state"]),
    )


def g9(ref):
    """PositionEmbeddingSine (centred + normalised, as Deformable-DETR builds it; and the DETR default)."""
    PE = importlib.import_module("alonet.transformers.position_encoding")
    mask = torch.zeros(2, 1, 7, 9)
    mask[1, :, 5:, :] = 1
    mask[1, :, :, 6:] = 1
    ft = torch.zeros(2, 4, 7, 9)
    out_c = PE.PositionEmbeddingSine(16, normalize=True, center=True)((ft, mask))
    out_d = PE.PositionEmbeddingSine(16, normalize=True)((ft, mask))
    np.savez_compressed(os.path.join(OUT, "g9_posenc.npz"), mask=_np(mask), centered=_np(out_c), default=_np(out_d))


def g10(ref):
    """Vanilla DETR transformer of the reference (post-norm, 2 + 2 layers) with padding mask, fp64."""
    sys.path.insert(0, os.path.dirname(OUT))
    from helpers import formula_state_dict

    _shell("alonet.detr", REF + "/alonet/detr")
    TR = importlib.import_module("alonet.detr.transformer")
    torch.manual_seed(1010)
    tr = TR.Transformer(d_model=64, nhead=4, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=96,
                        dropout=0.0, return_intermediate_dec=True).double().eval()
    tr.load_state_dict(formula_state_dict(tr.state_dict()))
    src = torch.randn(2, 64, 5, 7, dtype=torch.float64)
    pos = torch.randn(2, 64, 5, 7, dtype=torch.float64) * 0.5
    mask = torch.zeros(2, 5, 7, dtype=torch.bool)
    mask[1, :, 5:] = True
    query = torch.randn(9, 64, dtype=torch.float64)
    with torch.no_grad():
        out = tr(src, mask, query, pos)
    np.savez_compressed(os.path.join(OUT, "g10_detr_transformer.npz"), src=_np(src), pos=_np(pos), mask=_np(mask),
                        query=_np(query), hs=_np(out["hs"]), memory=_np(out["memory"]))


def g11(ref):
    """PanopticHead building blocks of the reference: MHAttentionMap and FPNstyleCNN (fp64, formula weights)."""
    sys.path.insert(0, os.path.dirname(OUT))
    from helpers import formula_state_dict

    _shell("alonet.detr_panoptic", REF + "/alonet/detr_panoptic")
    _shell("alonet.detr_panoptic.nn", REF + "/alonet/detr_panoptic/nn")
    MH = importlib.import_module("alonet.detr_panoptic.nn.MHAttention")
    FP = importlib.import_module("alonet.detr_panoptic.nn.FPNstyle")
    torch.manual_seed(1111)
    att = MH.MHAttentionMap(32, 32, 8, dropout=0.0).double().eval()
    att.load_state_dict(formula_state_dict(att.state_dict()))
    q = torch.randn(2, 5, 32, dtype=torch.float64)
    k = torch.randn(2, 32, 4, 6, dtype=torch.float64)
    mask = torch.zeros(2, 4, 6, dtype=torch.bool)
    mask[1, :, 4:] = True
    head = FP.FPNstyleCNN(32 + 8, [48, 24, 16], 32 * 4).double().eval()  # context_dim 128 -> inter dims 64,32,16,8
    head.load_state_dict(formula_state_dict(head.state_dict()))
    x = torch.randn(2, 32, 4, 6, dtype=torch.float64)
    fpns = [torch.randn(2, 48, 8, 12, dtype=torch.float64), torch.randn(2, 24, 16, 24, dtype=torch.float64),
            torch.randn(2, 16, 32, 48, dtype=torch.float64)]
    with torch.no_grad():
        w = att(q, k, mask=mask)
        seg = head(x, w, fpns)
    np.savez_compressed(os.path.join(OUT, "g11_panoptic_nn.npz"), q=_np(q), k=_np(k), mask=_np(mask), weights=_np(w),
                        x=_np(x), fpn0=_np(fpns[0]), fpn1=_np(fpns[1]), fpn2=_np(fpns[2]), seg=_np(seg))


def g17(ref):
    """The case of the reference's TensorRT-plugin test for the same kernel (alonet/torch2trt/plugins/ms_deform_im2col/test.py:103-121:
    N, M, D = 1, 8, 32; Lq, L, P = 12000, 4, 4; square levels 64 .. 8; value / locations ~ U(0, 1), weights normalised over L * P),
    compared there with ms_deform_attn_core_pytorch.  The inputs are 24 MB of uniform noise: the fixture keeps the seed, a sha256 of
    each input (the test re-draws them with the same torch CPU generator and checks the digests) and the reference's outputs for
    every 32nd query + the last 16, in float32 (what that test compares) and float64."""
    import hashlib

    N, M, D, Lq, L, P = 1, 8, 32, 12000, 4, 4
    shapes = torch.as_tensor([(64, 64), (32, 32), (16, 16), (8, 8)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    seed = 17
    torch.manual_seed(seed)
    value = torch.rand(N, S, M, D)
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    out32 = ref.func.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
    out64 = ref.func.ms_deform_attn_core_pytorch(value.double(), shapes, loc.double(), attn.double())
    keep = torch.cat([torch.arange(0, Lq, 32), torch.arange(Lq - 16, Lq)]).unique()
    digest = lambda t: np.frombuffer(hashlib.sha256(_np(t).tobytes()).digest(), dtype=np.uint8)
    np.savez_compressed(
        os.path.join(OUT, "g17_msda_trt_plugin_case.npz"),
        seed=np.int64(seed), dims=np.array([N, M, D, Lq, L, P], dtype=np.int64),
        shapes=_np(shapes.to(torch.int32)), level_start=_np(_level_start(shapes)),
        sha_value=digest(value), sha_loc=digest(loc), sha_attn=digest(attn),
        keep=_np(keep), out_f32=_np(out32[:, keep]), out_f64=_np(out64[:, keep]),
    )


def g18(ref):
    """The reference's CorrBlock UNDER AUTOGRAD (corr.py:12-60 is plain torch code; RAFT fine-tunes through it): one pyramid, three
    lookups at moving coordinates (RAFT's pattern: coordinates detached), a loss that weighs every window feature, gradients with
    respect to both feature maps; and once more with the coordinates attached.  Odd sizes, 3 levels, radius 2, fp64 run stored fp32."""
    B, C, H, W = 2, 24, 13, 18
    torch.manual_seed(18)
    f1 = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
    f2 = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
    grid = ref.rutils.coords_grid(B, H, W).double()
    coords = [grid + torch.randn(B, 2, H, W, dtype=torch.float64) * s for s in (0.0, 1.5, 4.0)]
    coords[1][:, :, 0, :] = coords[1][:, :, 0, :].round()
    coords[2][:, 0, 1, :] = -6.0                      # a row of windows left of the map
    wts = [torch.randn(B, 3 * 25, H, W, dtype=torch.float64) for _ in coords]
    blk = ref.corr.CorrBlock(f1, f2, num_levels=3, radius=2)
    outs = [blk(c) for c in coords]
    loss = sum((o.double() * w).sum() for o, w in zip(outs, wts))
    g1, g2 = torch.autograd.grad(loss, (f1, f2))
    c_att = coords[1].clone().requires_grad_(True)
    out_c = ref.corr.CorrBlock(f1.detach(), f2.detach(), num_levels=3, radius=2)(c_att)
    (gc,) = torch.autograd.grad((out_c.double() * wts[1]).sum(), c_att)
    f32 = lambda t: _np(t).astype(np.float32)
    np.savez_compressed(
        os.path.join(OUT, "g18_corr_grad.npz"),
        f1=f32(f1), f2=f32(f2), coords=np.stack([f32(c) for c in coords]), weights=np.stack([f32(w) for w in wts]),
        out=np.stack([f32(o) for o in outs]), grad_f1=f32(g1), grad_f2=f32(g2), grad_coords1=f32(gc),
    )


def main():
    if not os.path.isdir(REF):
        sys.exit("make_golden.py needs the reference checkout at /root/reference (build container only)")
    torch.set_num_threads(4)
    ref = load_reference()
    todo = [fn for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g17, g18)
            if len(sys.argv) == 1 or fn.__name__ in sys.argv[1:]]   # `make_golden.py g12` regenerates one fixture
    for fn in todo:
        fn(ref)
        print("wrote", fn.__name__)
    if len(sys.argv) == 1 or "g13" in sys.argv[1:]:
        # the criterion fixture needs the reference's real aloscene: its own interpreter (see make_golden_criterion.py)
        import subprocess

        subprocess.check_call([sys.executable, os.path.join(OUT, "make_golden_criterion.py")])
    models = [a for a in sys.argv[1:] if a in ("g14", "g14b", "g15", "g16")]
    if len(sys.argv) == 1 or models:
        # model-level fixtures (the reference's DeformableDETR / Detr / PanopticHead / Frame): own interpreter, real aloscene
        import subprocess

        subprocess.check_call([sys.executable, os.path.join(OUT, "make_golden_models.py")] + models)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print(f"total fixture bytes: {total}")


if __name__ == "__main__":
    main()
