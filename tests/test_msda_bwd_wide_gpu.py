"""The wide MSDA backward (csrc/msda_bwd_wide.hip: 16x16 query blocks sorted on chip) against the float64 C oracle, through the C ABI.

Reference semantics: alonet/deformable_detr/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159 (bilinear adjoint), :301-403."""
import ctypes
import os

import numpy as np
import pytest
import torch

import alo_hip
import oracle as O
from helpers import DETR_SHAPES, level_start

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x.to(dtype) if dtype is not None else x


def path_of(N, S, M, D, Lq, vdt, shapes_l, L=4, P=4):
    hint = None if shapes_l is None else (ctypes.c_int32 * (2 * len(shapes_l)))(*[int(v) for hw in shapes_l for v in hw])
    return alo_hip.lib().alo_msda_backward_path(N, S, M, D, L, Lq, P, vdt, alo_hip.ALO_F32, hint)


def pyramid_refs(shapes_l):
    refs = []
    for (h, w) in shapes_l:
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        refs.append(np.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1))
    return np.concatenate(refs, 0)


def encoder_case(shapes_l, N, M, rng, spread_px, heavy_tail=False, D=32):
    ref = pyramid_refs(shapes_l)
    S = ref.shape[0]
    norm = np.array([[w, h] for h, w in shapes_l], np.float64)[None, None, None, :, None, :]
    if heavy_tail:   # Student-t, 3 degrees of freedom: a few samples land far outside the block's window (the per-corner route)
        off = spread_px * rng.standard_t(3, (N, S, M, 4, 4, 2))
    else:
        off = rng.uniform(-spread_px, spread_px, (N, S, M, 4, 4, 2))
    loc = (ref[None, :, None, None, None, :] + off / norm).astype(np.float32)
    value = rng.standard_normal((N, S, M, D)).astype(np.float32)
    attn = rng.random((N, S, M, 4, 4)).astype(np.float32)
    attn /= attn.reshape(N, S, M, 16).sum(-1)[..., None, None]
    go = rng.standard_normal((N, S, M * D)).astype(np.float32)
    shapes = np.asarray(shapes_l, np.int32)
    return dict(value=value, shapes=shapes, level_start=level_start(shapes), loc=loc, attn=attn, grad_out=go)


def away_from_pixel_edges(loc, shapes_l, eps=1e-3):
    size = np.array([[w, h] for h, w in shapes_l], np.float64)[None, None, None, :, None, :]
    with np.errstate(invalid="ignore"):
        im = loc.astype(np.float64) * size - 0.5
        return np.nan_to_num(np.abs(im - np.round(im)), nan=1.0, posinf=1.0) > eps


def run_and_check(c, shapes_l, dtype=torch.float32, tol=1e-4):
    N, S, M, D = c["value"].shape
    vdt = alo_hip.ALO_F32 if dtype == torch.float32 else alo_hip.ALO_BF16
    assert path_of(N, S, M, D, S, vdt, shapes_l) == 2, "this launch must take msda_bwd_wide_kernel"
    shapes = dev(c["shapes"])
    shapes._alo_shapes = [tuple(int(v) for v in hw) for hw in shapes_l]
    args = [dev(c["value"], dtype), shapes, dev(c["level_start"]), dev(c["loc"]), dev(c["attn"]), dev(c["grad_out"], dtype)]
    if dtype != torch.float32:   # the oracle sees what the kernel sees: bf16-rounded value / grad_out, fp32 geometry
        c = dict(c, value=args[0].float().cpu().numpy(), grad_out=args[5].float().cpu().numpy())
        args[3], args[4] = args[3].to(dtype), args[4].to(dtype)
        c["loc"], c["attn"] = args[3].float().cpu().numpy(), args[4].float().cpu().numpy()
    gv, gl, ga = (x.float().cpu().numpy() for x in alo_hip.msda_backward(*args))
    rgv, rgl, rga = O.msda_backward(c["value"].astype(np.float64), c["shapes"], c["level_start"], c["loc"].astype(np.float64),
                                    c["attn"].astype(np.float64), c["grad_out"].astype(np.float64))
    vtol = tol if dtype == torch.float32 else 1e-2   # the wrapper narrows grad_value to bf16
    assert np.isfinite(gv).all() and np.isfinite(gl).all() and np.isfinite(ga).all()
    assert np.abs(gv - rgv).max() <= vtol * max(1.0, np.abs(rgv).max())
    assert np.abs(ga - rga).max() <= (tol if dtype == torch.float32 else 1e-2) * max(1.0, np.abs(rga).max())
    ok = away_from_pixel_edges(c["loc"], shapes_l).all(-1, keepdims=True)
    assert np.abs((gl - rgl) * ok).max() <= (tol if dtype == torch.float32 else 1e-2) * max(1.0, np.abs(rgl).max())
    return gv, gl, ga


def test_which_launches_take_the_wide_kernel():
    S = sum(h * w for h, w in DETR_SHAPES)
    assert path_of(4, S, 8, 32, S, alo_hip.ALO_F32, DETR_SHAPES) == 2
    assert path_of(4, S, 8, 32, S, alo_hip.ALO_BF16, DETR_SHAPES) == 2     # bf16 training no longer falls to per-corner atomics
    assert path_of(4, S, 8, 32, S, alo_hip.ALO_F32, None) == 1             # no host copy of the shapes: 4x4 tiles of 16 consecutive queries
    assert path_of(4, S, 8, 32, 300, alo_hip.ALO_F32, DETR_SHAPES) == 1    # decoder cross-attention (Lq != S)
    assert path_of(4, S, 8, 32, S, alo_hip.ALO_BF16, None) == 0
    assert path_of(4, S, 8, 64, S, alo_hip.ALO_F32, DETR_SHAPES) == 2     # D = 64 (d_model 512 at 8 heads): one workgroup per CU
    assert path_of(4, S, 8, 128, S, alo_hip.ALO_F32, DETR_SHAPES) == 0 and path_of(4, S, 8, 64, S, alo_hip.ALO_F32, None) == 0
    assert path_of(4, S, 8, 32, S, alo_hip.ALO_F32, [(100, 167), (50, 84), (25, 42), (13, 20)]) == 1   # host shapes that do not add up to S
    os.environ["ALO_MSDA_BWD"] = "tiled"
    try:
        assert path_of(4, S, 8, 32, S, alo_hip.ALO_F32, DETR_SHAPES) == 1
    finally:
        del os.environ["ALO_MSDA_BWD"]


PYRAMIDS = [
    [(37, 53), (19, 27), (10, 14), (5, 7)],      # odd sizes, ragged block edges on every level
    [(40, 40), (13, 13), (7, 9), (2, 3)],        # level ratios that are not 2
    [(64, 64), (8, 8), (4, 4), (2, 2)],          # a ratio of 8 between the first two levels (block side 2 on level 1)
    [(5, 7), (10, 14), (19, 27), (37, 53)],      # coarse level first
    [(3, 4), (2, 2), (1, 1), (1, 1)],            # smaller than one block
    [(16, 16), (16, 16), (16, 16), (16, 16)],    # four levels of the same size
]


@pytest.mark.parametrize("shapes_l", PYRAMIDS, ids=lambda s: "x".join(f"{h}.{w}" for h, w in s))
@pytest.mark.parametrize("spread,heavy", [(1.0, False), (3.0, True), (30.0, False)])
def test_wide_backward_vs_oracle_on_pyramids(shapes_l, spread, heavy):
    """Taps within a pixel of the query (every corner in the block's window), a heavy-tailed spread (a few samples take the per-corner
    route), and a spread wider than the window (most do), incl. taps off the border, on pyramids that stress the block table."""
    rng = np.random.default_rng(int(spread * 7) + len(shapes_l[0]))
    c = encoder_case(shapes_l, 2, 8, rng, spread, heavy)
    c["loc"][0, :5] += 2.0          # a few queries sample entirely outside the maps
    run_and_check(c, shapes_l)


@pytest.mark.parametrize("M,N", [(3, 1), (1, 3), (16, 2)])
def test_wide_backward_with_other_head_and_batch_counts(M, N):
    shapes_l = [(21, 30), (11, 15), (6, 8), (3, 4)]
    rng = np.random.default_rng(M * 10 + N)
    run_and_check(encoder_case(shapes_l, N, M, rng, 2.5, True), shapes_l)


def test_wide_backward_bf16_values():
    shapes_l = [(37, 53), (19, 27), (10, 14), (5, 7)]
    rng = np.random.default_rng(5)
    run_and_check(encoder_case(shapes_l, 2, 8, rng, 3.0, True), shapes_l, dtype=torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("spread,heavy", [(1.0, False), (3.0, True), (30.0, False)])
def test_wide_backward_with_64_channels_per_head(dtype, spread, heavy):
    """D = 64: 8 lanes per entry, both 128-byte halves of a row flushed by the same half wave."""
    shapes_l = [(37, 53), (19, 27), (10, 14), (5, 7)]
    rng = np.random.default_rng(64 + int(spread))
    run_and_check(encoder_case(shapes_l, 2, 4, rng, spread, heavy, D=64), shapes_l, dtype=dtype)


def test_wide_backward_with_nan_inf_and_huge_locations():
    """Samples the reference skips (NaN, +-inf, far outside) contribute nothing and get zero gradients; their neighbours are unharmed."""
    shapes_l = [(21, 30), (11, 15), (6, 8), (3, 4)]
    rng = np.random.default_rng(17)
    c = encoder_case(shapes_l, 1, 8, rng, 2.0)
    loc = c["loc"]
    loc[0, 3, 0, 0, 0] = np.nan
    loc[0, 4, 1, 1, 2, 0] = np.inf
    loc[0, 5, 2, 2, 1, 1] = -np.inf
    loc[0, 6, 3, 3, 3] = 1e30
    loc[0, 7, 4, 0, 1] = -1e30
    gv, gl, ga = run_and_check(c, shapes_l)
    for q, m, l, p in [(3, 0, 0, 0), (4, 1, 1, 2), (5, 2, 2, 1), (6, 3, 3, 3), (7, 4, 0, 1)]:
        assert ga[0, q, m, l, p] == 0 and (gl[0, q, m, l, p] == 0).all()


def test_wide_backward_when_every_query_hits_the_same_pixels():
    """Rows far longer than a 32-entry work item (the last segment of a row takes whatever is left): every sample of every query
    of a block lands on the same 2x2 pixels of each level."""
    shapes_l = [(20, 24), (10, 12), (5, 6), (3, 3)]
    rng = np.random.default_rng(23)
    c = encoder_case(shapes_l, 1, 8, rng, 0.0)
    ref = pyramid_refs(shapes_l)
    # the first 16x16 block of level 0 (and whatever else falls there) looks at the pixel (7.3, 6.6) of level 0 scaled to every level
    target = np.array([7.8 / 24, 7.1 / 20], np.float32)
    c["loc"][...] = target + (rng.uniform(-0.2, 0.2, c["loc"].shape) / np.array([[w, h] for h, w in shapes_l])[None, None, None, :, None, :]).astype(np.float32)
    assert ref.shape[0] == c["loc"].shape[1]
    run_and_check(c, shapes_l, tol=3e-4)   # sums of ~4000 fp32 terms per pixel


def test_wide_and_tiled_agree_at_full_size():
    S = sum(h * w for h, w in DETR_SHAPES)
    rng = np.random.default_rng(31)
    c = encoder_case(DETR_SHAPES, 2, 8, rng, 2.0, True)
    shapes = dev(c["shapes"])
    shapes._alo_shapes = list(DETR_SHAPES)
    args = (dev(c["value"]), shapes, dev(c["level_start"]), dev(c["loc"]), dev(c["attn"]), dev(c["grad_out"]))
    wide = alo_hip.msda_backward(*args)
    os.environ["ALO_MSDA_BWD"] = "tiled"
    try:
        tiled = alo_hip.msda_backward(*args)
    finally:
        del os.environ["ALO_MSDA_BWD"]
    for a, b in zip(wide, tiled):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())


def test_a_non_finite_grad_out_row_reaches_only_the_pixels_its_query_samples():
    """The reference adds w * attn * grad_out[q] to the four corners of q's samples and to nothing else (cuh:301-403), so an inf / NaN row
    of grad_out poisons exactly q's footprint.  The sorted gather multiplies idle slots by weight 0 — they must then re-read an entry of
    their OWN row, or 0 x inf from some other query's row would leak in."""
    shapes_l = [(21, 30), (11, 15), (6, 8), (3, 4)]
    rng = np.random.default_rng(41)
    c = encoder_case(shapes_l, 1, 8, rng, 2.0)
    S = c["loc"].shape[1]
    bad = [17, 400, S - 3]
    clean = run_and_check(dict(c), shapes_l)[0]
    go = c["grad_out"].copy()
    go[0, bad[0]] = np.inf
    go[0, bad[1]] = np.nan
    go[0, bad[2], :32] = -np.inf          # head 0 only
    shapes = dev(c["shapes"])
    shapes._alo_shapes = [tuple(hw) for hw in shapes_l]
    gv = alo_hip.msda_backward(dev(c["value"]), shapes, dev(c["level_start"]), dev(c["loc"]), dev(c["attn"]), dev(go))[0].cpu().numpy()
    # footprint of the three queries (per head for the last one)
    hit = np.zeros((S, 8), bool)
    st = level_start(c["shapes"])
    for q in bad:
        for lvl, (h, w) in enumerate(shapes_l):
            x = c["loc"][0, q, :, lvl, :, 0] * np.float32(w) - np.float32(0.5)
            y = c["loc"][0, q, :, lvl, :, 1] * np.float32(h) - np.float32(0.5)
            valid = (y > -1) & (x > -1) & (y < h) & (x < w)
            x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
            for dy in (0, 1):
                for dx in (0, 1):
                    yy, xx = y0 + dy, x0 + dx
                    ok = valid & (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                    heads = np.broadcast_to(np.arange(8)[:, None], x.shape)
                    if q == bad[2]:
                        ok = ok & (heads == 0)
                    hit[int(st[lvl]) + yy[ok] * w + xx[ok], heads[ok]] = True
    assert hit.any() and not hit.all()
    untouched = ~hit
    assert np.isfinite(gv[0][untouched]).all(), "a non-finite grad_out row leaked outside its query's footprint"
    assert np.abs(gv[0][untouched] - clean[0][untouched]).max() <= 1e-5 * max(1.0, np.abs(clean).max())


def test_autograd_function_in_bf16_takes_the_wide_kernel_and_matches_fp32_on_the_rounded_operands():
    """bf16 training of the encoder's self-attention: MSDeformAttnFunction.apply with bf16 value / locations / weights, backward through
    the dispatcher op; every gradient comes back in its input's dtype and equals the fp32 run on the same (rounded) operands to bf16
    rounding of the result."""
    from alonet.deformable_detr.ops.functions import MSDeformAttnFunction

    shapes_l = [(37, 53), (19, 27), (10, 14), (5, 7)]
    rng = np.random.default_rng(12)
    c = encoder_case(shapes_l, 2, 8, rng, 2.0, True)
    shapes = dev(c["shapes"])
    shapes._alo_shapes = [tuple(hw) for hw in shapes_l]
    start = dev(c["level_start"])
    outs = {}
    for dtype in (torch.bfloat16, torch.float32):
        v = dev(c["value"]).bfloat16().to(dtype).requires_grad_(True)
        loc = dev(c["loc"]).bfloat16().to(dtype).requires_grad_(True)
        attn = dev(c["attn"]).bfloat16().to(dtype).requires_grad_(True)
        out = MSDeformAttnFunction.apply(v, shapes, start, loc, attn, 64)
        out.backward(dev(c["grad_out"]).bfloat16().to(dtype))
        assert v.grad.dtype == dtype and loc.grad.dtype == dtype and attn.grad.dtype == dtype
        outs[dtype] = (out.detach().float(), v.grad.float(), loc.grad.float(), attn.grad.float())
    for got, ref in zip(outs[torch.bfloat16], outs[torch.float32]):
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() <= 2.0 ** -7 * max(1.0, ref.abs().max().item())
