"""Parity of the HIP multi-scale deformable attention (through the C ABI) with the golden vectors and the oracle."""
import numpy as np
import pytest
import torch

import alo_hip
import oracle as O
from helpers import DETR_SHAPES, level_start, msda_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x.to(dtype) if dtype is not None else x


def hip_forward(c, dtype):
    return alo_hip.msda_forward(dev(c["value"], dtype), dev(c["shapes"]), dev(c["level_start"]),
                                dev(c["loc"], dtype), dev(c["attn"], dtype), 64)


def hip_backward(c, dtype):
    return alo_hip.msda_backward(dev(c["value"], dtype), dev(c["shapes"]), dev(c["level_start"]),
                                 dev(c["loc"], dtype), dev(c["attn"], dtype), dev(c["grad_out"], dtype), 64)


def test_native_library_is_the_one_running():
    assert alo_hip.is_available()
    assert "libalo_hotpath.so" in open("/proc/self/maps").read()


# ---- golden vectors (reference outputs) -------------------------------------------------------------------------
def test_g1_reference_optest_case_fp64_and_fp32(golden):
    g = golden("g1_msda_optest.npz")
    c = dict(value=g["value"], shapes=g["shapes"], level_start=g["level_start"], loc=g["loc"], attn=g["attn"])
    out = hip_forward(c, torch.float64).cpu().numpy()
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-5, atol=1e-8)  # torch.allclose defaults (ops/test.py:62)
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-12, atol=1e-15)  # and in fact to rounding
    c = dict(value=g["value_b"], shapes=g["shapes"], level_start=g["level_start"], loc=g["loc_b"], attn=g["attn_b"])
    out = hip_forward(c, torch.float32).cpu().numpy()
    np.testing.assert_allclose(out, g["out_f32"], rtol=1e-2, atol=1e-3)  # the reference's fp32 bar (ops/test.py:83)
    np.testing.assert_allclose(out, g["out_f32"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("D", [30, 32, 64, 71])
def test_g2_gradcheck_set_fp64(golden, D):
    g = golden(f"g2_msda_grad_D{D}.npz")
    c = {k: g[k] for k in ("value", "shapes", "level_start", "loc", "attn", "grad_out")}
    np.testing.assert_allclose(hip_forward(c, torch.float64).cpu().numpy(), g["out"], rtol=1e-12, atol=1e-15)
    gv, gl, ga = (x.cpu().numpy() for x in hip_backward(c, torch.float64))
    np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("dtype,atol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
def test_g3_borders_fwd_bwd(golden, dtype, atol):
    g = golden("g3_msda_medium.npz")
    c = {k: g[k] for k in ("value", "shapes", "level_start", "loc", "attn", "grad_out")}
    np.testing.assert_allclose(hip_forward(c, dtype).double().cpu().numpy(), g["out"], rtol=0, atol=atol)
    gv, gl, ga = (x.double().cpu().numpy() for x in hip_backward(c, dtype))
    np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-5, atol=max(atol, 2e-6))
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=1e-5, atol=2e-4)  # |grad_loc| reaches 160 here
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=1e-5, atol=max(atol, 1e-5))


def test_g8_known_answers(golden):
    g = golden("g8_known_answers.npz")
    c = {k: g[k] for k in ("value", "shapes", "level_start", "loc", "attn")}
    np.testing.assert_allclose(hip_forward(c, torch.float64).cpu().numpy().ravel(), g["expected"], atol=1e-15)
    np.testing.assert_allclose(hip_forward(c, torch.float32).cpu().numpy().ravel(), g["expected"], atol=1e-6)


def test_g17_the_reference_trt_plugin_test_case_at_full_size(golden):
    """The case the reference's TensorRT-plugin test feeds the same kernel (N, M, D = 1, 8, 32; Lq = 12000; levels 64^2 .. 8^2;
    torch2trt/plugins/ms_deform_im2col/test.py:103-136), all 12000 queries through the op; the kept rows against the reference's own
    outputs: float64 to rounding, float32 far inside the reference's bar, bf16 values within half a bf16 ulp of the result."""
    from helpers import g17_inputs

    g = golden("g17_msda_trt_plugin_case.npz")
    drawn = g17_inputs(g)
    assert drawn is not None, "torch's CPU generator draws another stream than the build the fixture was made with"
    value, loc, attn = drawn
    keep = torch.from_numpy(g["keep"]).to(DEV)
    shapes, start = torch.from_numpy(g["shapes"]).to(DEV), torch.from_numpy(g["level_start"]).to(DEV)
    want64, want32 = g["out_f64"], g["out_f32"]
    out64 = alo_hip.msda_forward(value.double().to(DEV), shapes, start, loc.double().to(DEV), attn.double().to(DEV))[:, keep].cpu().numpy()
    np.testing.assert_allclose(out64, want64, rtol=1e-12, atol=1e-14)
    out32 = alo_hip.msda_forward(value.to(DEV), shapes, start, loc.to(DEV), attn.to(DEV))[:, keep].cpu().numpy()
    np.testing.assert_allclose(out32, want32, rtol=1e-2, atol=1e-3)   # the reference's fp32 bar (ops/test.py:83)
    err = np.abs(out32 - want64)
    assert err.max() <= 5e-6 and err.mean() <= 5e-7, (err.max(), err.mean())   # what the plugin test prints: mean / max abs error
    vb = value.to(torch.bfloat16)
    outb = alo_hip.msda_forward(vb.to(DEV), shapes, start, loc.to(DEV), attn.to(DEV))[:, keep].float().cpu().numpy()
    exact = O.msda_forward(vb.double().numpy(), g["shapes"], g["level_start"], loc[:, g["keep"]].double().numpy(), attn[:, g["keep"]].double().numpy())
    assert np.all(np.abs(outb - exact) <= 2.0 ** -8 * np.abs(exact) + 1e-6)   # half a bf16 ulp is at most 2^-8 of the value


# ---- oracle on seeded inputs: every kernel variant ---------------------------------------------------------------
CASES = [  # (N, M, D, Lq, shapes, P)          which plan it exercises (fp32)
    (2, 8, 32, 77, [(16, 21), (8, 11), (4, 6), (2, 3)], 4),  # vec4 / group 8 / unrolled LP=16  (the DETR shape)
    (1, 8, 32, 300, [(20, 27), (10, 14), (5, 7), (3, 4)], 4),  # decoder-like query count
    (2, 4, 16, 33, [(9, 7), (5, 4)], 8),  # vec4 / group 4 / unrolled LP=16
    (1, 2, 64, 19, [(6, 5), (3, 3), (2, 2)], 2),  # vec4 / group 16 / runtime LP
    (1, 1, 256, 9, [(5, 5)], 3),  # vec4 / group 64
    (1, 2, 512, 5, [(4, 6)], 2),  # vec4 / group 64, two channel chunks
    (1, 3, 30, 21, [(6, 4), (3, 2)], 2),  # scalar path (D % 4 != 0), group 64
    (2, 2, 71, 11, [(6, 4), (3, 2)], 2),  # scalar path, two chunks
    (1, 2, 2, 2, [(6, 4), (3, 2)], 2),  # scalar path, group 8 (the reference test's D)
    (1, 2, 8, 13, [(7, 3)], 1),  # vec4 / group 4, L*P = 1
    (2, 8, 32, 454, [(16, 21), (8, 11), (4, 6), (2, 3)], 4),  # Lq == S: the tiled backward groups queries as 8x8 blocks
    (1, 4, 32, 130, [(9, 13), (5, 7), (3, 4), (2, 2)], 4),  # tiled backward, 16-query runs with a ragged tail, 4 heads
    (1, 6, 32, 168, [(9, 13), (5, 7), (3, 4), (2, 2)], 4),  # tiled backward, Lq == S, a head count that is not a power of two
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"N{c[0]}M{c[1]}D{c[2]}Lq{c[3]}L{len(c[4])}P{c[5]}")
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_forward_backward_vs_oracle(case, dtype):
    N, M, D, Lq, shapes, P = case
    npdt = np.float32 if dtype == torch.float32 else np.float64
    c = msda_case(1234 + D + Lq, N, M, D, Lq, shapes, P, npdt, loc_range=(-0.3, 1.3))
    ref = O.msda_forward(c["value"].astype(np.float64), c["shapes"], c["level_start"], c["loc"].astype(np.float64),
                         c["attn"].astype(np.float64))
    rgv, rgl, rga = O.msda_backward(c["value"].astype(np.float64), c["shapes"], c["level_start"],
                                    c["loc"].astype(np.float64), c["attn"].astype(np.float64),
                                    c["grad_out"].astype(np.float64))
    out = hip_forward(c, dtype).double().cpu().numpy()
    gv, gl, ga = (x.double().cpu().numpy() for x in hip_backward(c, dtype))
    if dtype == torch.float64:
        tol = dict(rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(out, ref, **tol)
        np.testing.assert_allclose(gv, rgv, **tol)
        np.testing.assert_allclose(gl, rgl, rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(ga, rga, **tol)
    else:  # fp32 kernel against the fp64 oracle on fp32-representable inputs: <= 1e-5 abs at O(1) values
        np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gv, rgv, rtol=1e-4, atol=2e-5)
        scale = max(1.0, np.abs(rgl).max())
        np.testing.assert_allclose(gl, rgl, rtol=1e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(ga, rga, rtol=1e-4, atol=1e-4)


def test_bf16_storage_matches_oracle_on_bf16_rounded_inputs():
    """bf16 value/out, fp32 geometry and accumulation: only the final rounding of `out` separates it from the oracle."""
    c = msda_case(77, 2, 8, 32, 50, [(16, 21), (8, 11), (4, 6), (2, 3)], 4, np.float32)
    vb = dev(c["value"]).bfloat16()
    out = alo_hip.msda_forward(vb, dev(c["shapes"]), dev(c["level_start"]), dev(c["loc"]), dev(c["attn"]), 64)
    assert out.dtype == torch.bfloat16
    ref = O.msda_forward(vb.float().cpu().numpy().astype(np.float64), c["shapes"], c["level_start"],
                         c["loc"].astype(np.float64), c["attn"].astype(np.float64))
    err = np.abs(out.float().cpu().numpy() - ref)
    assert np.all(err <= np.abs(ref) * 2.0 ** -8 + 1e-6)  # half an ulp of bf16 (8 significant bits) + fp32 noise
    # gradients of the bf16 path accumulate in fp32
    go = dev(c["grad_out"]).bfloat16()
    gv, gl, ga = alo_hip.msda_backward(vb, dev(c["shapes"]), dev(c["level_start"]), dev(c["loc"]), dev(c["attn"]), go, 64)
    rgv, rgl, rga = O.msda_backward(vb.float().cpu().numpy().astype(np.float64), c["shapes"], c["level_start"],
                                    c["loc"].astype(np.float64), c["attn"].astype(np.float64),
                                    go.float().cpu().numpy().astype(np.float64))
    assert gv.dtype == torch.bfloat16 and gl.dtype == torch.float32
    np.testing.assert_allclose(gl.cpu().numpy(), rgl, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(ga.cpu().numpy(), rga, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gv.float().cpu().numpy(), rgv, rtol=2.0 ** -7, atol=1e-3)


def test_nan_outside_the_sampled_footprint_does_not_leak():
    """Corners outside the map are never read (buffer bounds check), as in the reference's guarded loads."""
    c = msda_case(5, 1, 2, 32, 6, [(4, 4)], 2, np.float32, loc_range=(1.2, 1.6))  # every sample is out of range
    c["value"][:] = np.nan
    out = hip_forward(c, torch.float32).cpu().numpy()
    assert np.all(out == 0)


# ---- reference-style API on the GPU ------------------------------------------------------------------------------
def test_autograd_function_and_gradcheck():
    from alonet.deformable_detr.ops.functions import MSDeformAttnFunction

    for D in (30, 32):  # ops/test.py:130-131 runs gradcheck for D in {30, 32, 64, 71}
        c = msda_case(3 + D, 1, 2, D, 2, [(6, 4), (3, 2)], 2, np.float64, loc_range=(0.0, 1.0))
        c["value"] *= 0.01
        value = dev(c["value"]).requires_grad_(True)
        loc = dev(c["loc"]).requires_grad_(True)
        attn = dev(c["attn"]).requires_grad_(True)
        assert torch.autograd.gradcheck(MSDeformAttnFunction.apply,
                                        (value, dev(c["shapes"]), dev(c["level_start"]), loc, attn, 2))


def test_module_on_gpu_matches_reference_module(golden):
    from alonet.deformable_detr.ops.modules import MSDeformAttn

    g = golden("g4_msda_module.npz")
    d_model, n_levels, n_heads, n_points = (int(x) for x in g["cfg"])
    m = MSDeformAttn(d_model, n_levels, n_heads, n_points).double()
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    m = m.to(DEV)
    shapes = dev(g["shapes"]).to(torch.int32)
    with torch.no_grad():
        out2 = m(dev(g["query"]), dev(g["ref2"]), dev(g["src"]), shapes, dev(g["level_start"]), dev(g["mask"]))
        out4 = m(dev(g["query"]), dev(g["ref4"]), dev(g["src"]), shapes, dev(g["level_start"]), dev(g["mask"]))
    np.testing.assert_allclose(out2.cpu().numpy(), g["out2"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(out4.cpu().numpy(), g["out4"], rtol=1e-10, atol=1e-11)
    m32 = m.float()
    with torch.no_grad():
        o32 = m32(dev(g["query"]).float(), dev(g["ref2"]).float(), dev(g["src"]).float(), shapes,
                  dev(g["level_start"]), dev(g["mask"]))
    np.testing.assert_allclose(o32.cpu().numpy(), g["out2"], rtol=1e-3, atol=1e-3)  # north-star bar: <= 1e-3 max-abs


def test_error_behaviour_matches_the_reference_op():
    c = msda_case(1, 2, 2, 8, 3, [(4, 4)], 2, np.float32)
    v, sh, st, loc, attn = dev(c["value"]), dev(c["shapes"]), dev(c["level_start"]), dev(c["loc"]), dev(c["attn"])
    with pytest.raises(RuntimeError, match="value tensor has to be contiguous"):
        alo_hip.msda_forward(v.transpose(2, 3), sh, st, loc, attn, 64)
    with pytest.raises(RuntimeError, match="sampling_loc must be a CUDA tensor"):
        alo_hip.msda_forward(v, sh, st, loc.cpu(), attn, 64)
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        torch.ops.alonet_custom.ms_deform_attn_forward(torch.cat([v, v[:1]]), sh, st, torch.cat([loc, loc[:1]]),
                                                       torch.cat([attn, attn[:1]]), 2)
    out_a = alo_hip.msda_forward(v, sh, st, loc, attn, 1)  # im2col_step is a hint: results do not depend on it
    out_b = alo_hip.msda_forward(v, sh, st, loc, attn, 64)
    assert torch.equal(out_a, out_b)


# ---- BASELINE.json sizes: size-independent properties + oracle on the full encoder call --------------------------
def _full_size_case(N, Lq, seed):
    rng = np.random.default_rng(seed)
    shapes = np.asarray(DETR_SHAPES, np.int32)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.standard_normal((N, S, 8, 32)).astype(np.float32)
    loc = rng.uniform(-0.05, 1.05, (N, Lq, 8, 4, 4, 2)).astype(np.float32)
    attn = rng.random((N, Lq, 8, 4, 4)).astype(np.float32)
    attn /= attn.reshape(N, Lq, 8, 16).sum(-1)[..., None, None]
    return dict(value=value, shapes=shapes, level_start=level_start(shapes), loc=loc, attn=attn)


def test_full_size_encoder_call_vs_oracle():
    """One batch item of the config-2 encoder call (S = Lq = 22223, M=8, D=32, L=P=4) against the C oracle."""
    c = _full_size_case(1, 22223, 11)
    out = hip_forward(c, torch.float32).cpu().numpy()
    ref = O.msda_forward(c["value"], c["shapes"], c["level_start"], c["loc"], c["attn"])  # fp32 oracle, same inputs
    assert np.abs(out - ref).max() <= 1e-5


def test_full_size_properties_batch8():
    """Linearity in `value`, batch independence and constant reproduction at N=8 (the bench workload)."""
    c = _full_size_case(8, 22223, 12)
    sh, st = dev(c["shapes"]), dev(c["level_start"])
    v, loc, attn = dev(c["value"]), dev(c["loc"]), dev(c["attn"])
    out = alo_hip.msda_forward(v, sh, st, loc, attn, 64)
    # (1) linear in value
    out2 = alo_hip.msda_forward(v * 2.0 + 1.0, sh, st, loc, attn, 64)
    ones = alo_hip.msda_forward(torch.ones_like(v), sh, st, loc, attn, 64)
    assert (out2 - (2.0 * out + ones)).abs().max().item() <= 2e-5
    # (2) a constant map is reproduced times the in-range bilinear mass, which never exceeds the attention mass (1)
    assert ones.max().item() <= 1.0 + 1e-5 and ones.min().item() >= 0.0
    inner = alo_hip.msda_forward(torch.ones_like(v), sh, st, loc.clamp(0.2, 0.8), attn, 64)
    assert (inner - 1.0).abs().max().item() <= 1e-5  # all four corners inside: weights sum to exactly one
    # (3) batch items are independent: item 3 alone gives the same bits
    solo = alo_hip.msda_forward(v[3:4].contiguous(), sh, st, loc[3:4].contiguous(), attn[3:4].contiguous(), 64)
    assert torch.equal(solo[0], out[3])
    # (4) bf16 storage stays within bf16 rounding of the fp32 result on the same (bf16-rounded) values
    vb = v.bfloat16()
    ob = alo_hip.msda_forward(vb, sh, st, loc, attn, 64).float()
    of = alo_hip.msda_forward(vb.float(), sh, st, loc, attn, 64)
    assert ((ob - of).abs() <= of.abs() * 2.0 ** -8 + 1e-6).all()


# ---- fused prologue (softmax + sampling-location arithmetic inside the kernel) ------------------------------------
def _prologue_in_torch(offsets, logits, ref, shapes, P):
    """What MSDeformAttn.forward does between its linear layers and the op (reference ms_deform_attn.py:119-133)."""
    N, Lq, M, L = offsets.shape[:4]
    attn = torch.softmax(logits, -1).view(N, Lq, M, L, P)
    if ref.shape[-1] == 2:
        normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
        loc = ref[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + offsets / P * ref[:, :, None, :, None, 2:] * 0.5
    return loc.contiguous(), attn.contiguous()


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("shape", [(2, 8, 32, 77, [(16, 21), (8, 11), (4, 6), (2, 3)], 4),  # L*P = 16: shuffle softmax
                                   (1, 4, 16, 300, [(9, 7), (5, 4), (3, 3)], 2),  # L*P = 6: generic softmax
                                   (1, 2, 30, 21, [(6, 4), (3, 2)], 2)])  # scalar channel path
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
def test_fused_prologue_matches_unfused(shape, ref_dim, dtype):
    N, M, D, Lq, shapes_l, P = shape
    L = len(shapes_l)
    gen = torch.Generator(device=DEV).manual_seed(99 + D + ref_dim)
    shapes = torch.tensor(shapes_l, dtype=torch.int32, device=DEV)
    start = dev(level_start(shapes_l))
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    geo = torch.float64 if dtype == torch.float64 else torch.float32
    value = torch.randn(N, S, M, D, generator=gen, device=DEV, dtype=geo).to(dtype)
    offsets = (torch.randn(N, Lq, M, L, P, 2, generator=gen, device=DEV, dtype=geo) * 2.5).to(dtype)
    logits = (torch.randn(N, Lq, M, L * P, generator=gen, device=DEV, dtype=geo) * 2.0).to(dtype)
    ref = torch.rand(N, Lq, L, ref_dim, generator=gen, device=DEV, dtype=geo)
    if ref_dim == 4:
        ref[..., 2:] *= 0.4
    fused = alo_hip.msda_forward_fused(value, shapes, start, offsets, logits, ref)
    loc, attn = _prologue_in_torch(offsets.to(geo), logits.to(geo), ref, shapes, P)
    plain = alo_hip.msda_forward(value, shapes, start, loc, attn, 64)
    assert fused.dtype == dtype and fused.shape == plain.shape
    if dtype == torch.float64:
        assert (fused - plain).abs().max().item() <= 1e-12
    elif dtype == torch.float32:
        assert (fused - plain).abs().max().item() <= 5e-6  # softmax summation order, exp
    else:  # both round the same fp32 sum to bf16: at most one bf16 ulp apart
        assert ((fused.float() - plain.float()).abs() <= plain.float().abs() * 2.0 ** -7 + 1e-6).all()


def test_module_uses_fused_prologue_in_inference_and_unfused_under_grad(golden):
    from alonet.deformable_detr.ops.modules import MSDeformAttn

    g = golden("g4_msda_module.npz")
    d_model, n_levels, n_heads, n_points = (int(x) for x in g["cfg"])
    m = MSDeformAttn(d_model, n_levels, n_heads, n_points).double()
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    m = m.to(DEV)
    args = (dev(g["query"]), dev(g["ref4"]), dev(g["src"]), dev(g["shapes"]).to(torch.int32), dev(g["level_start"]),
            dev(g["mask"]))
    with alo_hip.LaunchTimer() as t1, torch.no_grad():
        fused = m(*args)
    with alo_hip.LaunchTimer() as t2:
        plain = m(*args)  # parameters require grad -> the reference-shaped op (differentiable) is used
    assert any(k.startswith("msda_fwd_fused") for k in t1.summary())
    assert any(k.startswith("msda_fwd/") for k in t2.summary())
    np.testing.assert_allclose(fused.cpu().numpy(), g["out4"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(plain.detach().cpu().numpy(), g["out4"], rtol=1e-10, atol=1e-11)
    plain.sum().backward()
    assert m.sampling_offsets.weight.grad is not None and torch.isfinite(m.value_proj.weight.grad).all()


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("N,M,D,Lq", [(2, 8, 32, 77), (1, 8, 32, 16), (3, 4, 16, 301), (1, 2, 8, 5)])
def test_head_major_path_is_bit_identical_and_masks_padding(N, M, D, Lq, ref_dim):
    """alo_value_head_major + alo_msda_forward_fused_hm == masked_fill + alo_msda_forward_fused, bit for bit."""
    shapes_l = [(16, 21), (8, 11), (4, 6), (2, 3)]
    gen = torch.Generator(device=DEV).manual_seed(7 + D + Lq)
    shapes = torch.tensor(shapes_l, dtype=torch.int32, device=DEV)
    start = dev(level_start(shapes_l))
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = torch.randn(N, S, M, D, generator=gen, device=DEV).bfloat16()
    offsets = (torch.randn(N, Lq, M, 4, 4, 2, generator=gen, device=DEV) * 2.5).bfloat16()
    logits = (torch.randn(N, Lq, M, 16, generator=gen, device=DEV) * 2.0).bfloat16()
    ref = torch.rand(N, Lq, 4, ref_dim, generator=gen, device=DEV)
    if ref_dim == 4:
        ref[..., 2:] *= 0.4
    mask = torch.rand(N, S, generator=gen, device=DEV) < 0.3
    assert alo_hip.head_major_supported(value, 4, 4)
    vhm = alo_hip.value_head_major(value, mask)
    assert vhm.shape == (N, M, S, D)
    assert torch.equal(vhm, value.masked_fill(mask[..., None, None], 0).permute(0, 2, 1, 3))
    assert torch.equal(alo_hip.value_head_major(value, None), value.permute(0, 2, 1, 3))
    got = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref)
    want = alo_hip.msda_forward_fused(value.masked_fill(mask[..., None, None], 0), shapes, start, offsets, logits, ref)
    assert torch.equal(got, want)
    # ... and directly against the float64 definition (oracle), not only against the sibling kernel
    loc, attn = _prologue_in_torch(offsets.float(), logits.float(), ref, shapes, 4)
    exact = O.msda_forward(value.masked_fill(mask[..., None, None], 0).double().cpu().numpy(), shapes.cpu().numpy(),
                           start.cpu().numpy(), loc.double().cpu().numpy(), attn.double().cpu().numpy())
    err = np.abs(got.double().cpu().numpy() - exact)
    assert np.all(err <= np.abs(exact) * 2.0 ** -8 + 2e-5)   # half a bf16 ulp + the fp32 prologue (hardware exp / rcp)


def test_head_major_rejects_other_shapes():
    shapes = torch.tensor([(4, 4), (2, 2)], dtype=torch.int32, device=DEV)
    start = dev(level_start([(4, 4), (2, 2)]))
    v = torch.randn(1, 2, 20, 32, device=DEV).bfloat16()  # (N, M, S, D) but L = 2
    off = torch.zeros(1, 3, 2, 2, 4, 2, device=DEV).bfloat16()
    lg = torch.zeros(1, 3, 2, 8, device=DEV).bfloat16()
    with pytest.raises(RuntimeError, match="L = P = 4"):
        alo_hip.msda_forward_fused_hm(v, shapes, start, off, lg, torch.rand(1, 3, 2, 2, device=DEV))
    assert not alo_hip.head_major_supported(torch.zeros(1, 4, 2, 32, device=DEV), 4, 4)  # fp32


@pytest.mark.parametrize("N,Lq", [(2, 77), (1, 300)])
def test_head_major_takes_offsets_and_logits_as_slices_of_a_merged_projection(N, Lq):
    """alo_msda_forward_fused_hm_rows: raw offsets and logits as column slices of one (N, Lq, 3*M*L*P) buffer give the bits of the
    dense call."""
    M, D, L, P = 8, 32, 4, 4
    shapes_l = [(16, 21), (8, 11), (4, 6), (2, 3)]
    gen = torch.Generator(device=DEV).manual_seed(11 + Lq)
    shapes = torch.tensor(shapes_l, dtype=torch.int32, device=DEV)
    start = dev(level_start(shapes_l))
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    vhm = torch.randn(N, M, S, D, generator=gen, device=DEV).bfloat16()
    both = (torch.randn(N, Lq, M * L * P * 3, generator=gen, device=DEV) * 2.0).bfloat16()
    ref = torch.rand(N, Lq, L, 2, generator=gen, device=DEV)
    offsets = both[..., :M * L * P * 2].view(N, Lq, M, L, P, 2)
    logits = both[..., M * L * P * 2:].view(N, Lq, M, L * P)
    assert not offsets.is_contiguous() and not logits.is_contiguous()
    got = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref)
    want = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets.contiguous(), logits.contiguous(), ref)
    assert torch.equal(got, want)


# ---- the exact kernel bench.py times: bf16, fused prologue, head-major, merged-projection slices, padding mask ---------
def _bench_path_case(N, Lq_is_S, shapes_l, seed, encoder_like):
    """Inputs shaped as MSDeformAttn.forward hands them to alo_msda_forward_fused_hm_rows in the bf16 inference path."""
    M, D, L, P = 8, 32, 4, 4
    gen = torch.Generator(device=DEV).manual_seed(seed)
    shapes = torch.tensor(shapes_l, dtype=torch.int32, device=DEV)
    shapes._alo_shapes = [tuple(hw) for hw in shapes_l]   # the host copy DeformableTransformer attaches: enables the LDS-resident kernel
    start = dev(level_start(shapes_l))
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    Lq = S if Lq_is_S else 300
    value = torch.randn(N, S, M, D, generator=gen, device=DEV).bfloat16()
    mask = torch.rand(N, S, generator=gen, device=DEV) < 0.1
    both = torch.randn(N, Lq, M * L * P * 3, generator=gen, device=DEV)
    both[..., :M * L * P * 2] *= 3.0   # offsets of a few pixels
    both = both.bfloat16()
    if encoder_like:   # reference points = every pixel's own centre on every level (what the encoder passes)
        refs = []
        for (h, w) in shapes_l:
            ys, xs = torch.meshgrid(torch.arange(h, device=DEV), torch.arange(w, device=DEV), indexing="ij")
            refs.append(torch.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1))
        ref = torch.cat(refs, 0)[None, :, None, :].expand(N, S, L, 2).contiguous()
    else:
        ref = torch.rand(N, Lq, L, 2, generator=gen, device=DEV)
    offsets = both[..., :M * L * P * 2].view(N, Lq, M, L, P, 2)
    logits = both[..., M * L * P * 2:].view(N, Lq, M, L * P)
    return value, mask, offsets, logits, ref, shapes, start


def _oracle_on_queries(value, mask, offsets, logits, ref, shapes, start, qsel):
    """oracle.msda_forward (float64) on the query subset ``qsel``, fed the kernel's own bf16-rounded inputs."""
    loc, attn = _prologue_in_torch(offsets[:, qsel].double(), logits[:, qsel].double(), ref[:, qsel].double(), shapes, 4)
    v = value.masked_fill(mask[..., None, None], 0).double().cpu().numpy()
    return O.msda_forward(v, shapes.cpu().numpy(), start.cpu().numpy(), loc.cpu().numpy(), attn.cpu().numpy())


@pytest.mark.parametrize("encoder_like", [True, False])
def test_bench_kernel_direct_vs_oracle_small(encoder_like):
    shapes_l = [(16, 21), (8, 11), (4, 6), (2, 3)]
    value, mask, offsets, logits, ref, shapes, start = _bench_path_case(2, encoder_like, shapes_l, 31, encoder_like)
    vhm = alo_hip.value_head_major(value, mask)
    assert not offsets.is_contiguous()   # the merged-projection slices go to alo_msda_forward_fused_hm_rows as they are
    got = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref)
    exact = _oracle_on_queries(value, mask, offsets, logits, ref, shapes, start, slice(None))
    err = np.abs(got.double().cpu().numpy() - exact)
    assert np.all(err <= np.abs(exact) * 2.0 ** -8 + 2e-5)   # half a bf16 ulp of the result + the fp32 prologue


def test_bench_kernel_direct_vs_oracle_full_size_batch8():
    """BASELINE configs[1] shape of the encoder call (N = 8, S = Lq = 22223): every 41st query of the launch bench.py times
    (msda_fwd_bf16_resident_kernel: levels 2-3 of every (image, head) slab resident in LDS) against the float64 oracle, and the
    whole output against the plain head-major kernel."""
    value, mask, offsets, logits, ref, shapes, start = _bench_path_case(8, True, DETR_SHAPES, 32, True)
    vhm = alo_hip.value_head_major(value, mask)
    with alo_hip.LaunchTimer() as timer:
        got = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref)
    assert "msda_fwd_fused_resident/Lq=22223" in timer.summary(), timer.summary().keys()
    assert torch.equal(got, alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=False))
    qsel = slice(0, None, 41)
    exact = _oracle_on_queries(value, mask, offsets, logits, ref, shapes, start, qsel)
    err = np.abs(got[:, qsel].double().cpu().numpy() - exact)
    assert np.all(err <= np.abs(exact) * 2.0 ** -8 + 2e-5)
    # the last queries of the launch (tail tiles) too
    tail = slice(22223 - 70, None)
    exact = _oracle_on_queries(value, mask, offsets, logits, ref, shapes, start, tail)
    err = np.abs(got[:, tail].double().cpu().numpy() - exact)
    assert np.all(err <= np.abs(exact) * 2.0 ** -8 + 2e-5)


@pytest.mark.parametrize("kind", ["survey", "trained", "uniform"])
def test_bench_kernel_full_size_on_the_wider_sampling_distributions(kind):
    """The kernel bench.py times, at N = 8, S = Lq = 22223, AWAY from the init-time ring the benchmark's random-init model samples:
    SURVEY 8(d)'s micro-benchmark locations (own pixel centre + U(-0.05, 0.05) of the map: +-8 x +-5 px on level 0) and locations
    uniform over the whole map (no locality between neighbouring queries: where the resident kernel's ~190-consecutive-queries
    locality is gone), and the trained-like offsets of tools/kbench.py (ring + heavy-tailed 1.5-3 px spread per level: what the
    `trained_like` leg of bench.py samples).  Resident == plain head-major kernel bit for bit, and a strided query subset + the tail within half a bf16
    ulp of the float64 oracle (tools/kbench.py generates the same inputs for the `micro` entries of the bench line)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kbench

    value, shapes, start, offsets, logits, ref = kbench.fused_inputs(8, torch.bfloat16, seed=5, kind=kind)
    mask = torch.zeros(value.shape[:2], dtype=torch.bool, device=DEV)
    mask[:, ::17] = True
    vhm = alo_hip.value_head_major(value, mask)
    with alo_hip.LaunchTimer() as timer:
        got = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref)
    assert "msda_fwd_fused_resident/Lq=22223" in timer.summary(), timer.summary().keys()
    assert torch.equal(got, alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=False))
    for qsel in (slice(0, None, 53), slice(22223 - 40, None)):
        exact = _oracle_on_queries(value, mask, offsets, logits, ref, shapes, start, qsel)
        err = np.abs(got[:, qsel].double().cpu().numpy() - exact)
        assert np.all(err <= np.abs(exact) * 2.0 ** -8 + 2e-5)


# ---- backward at the config-4 encoder size -------------------------------------------------------------------------------
def _away_from_pixel_edges(loc, shapes_l, eps=1e-3):
    """grad_sampling_loc is the derivative of a piecewise-bilinear function: it JUMPS where a coordinate crosses an integer.
    A sample within fp32 rounding of such an edge may legitimately take either side (fp32 vs fp64 evaluation of loc * size - 0.5,
    fused or unfused): mask those samples out of the grad_loc comparison.  (N, Lq, M, L, P, 1) bool."""
    size = np.array([[w, h] for h, w in shapes_l], np.float64)[None, None, None, :, None, :]
    im = loc.astype(np.float64) * size - 0.5
    return (np.abs(im - np.round(im)) > eps).all(-1, keepdims=True)


def _encoder_like_loc(N, shapes_l, rng, spread_px=4.0):
    """Sampling locations of an encoder call: every query is a pixel of the pyramid, its points lie within a few pixels of
    its own position on every level (N, S, 8, 4, 4, 2)."""
    refs = []
    for (h, w) in shapes_l:
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        refs.append(np.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1))
    ref = np.concatenate(refs, 0)
    S = ref.shape[0]
    norm = np.array([[w, h] for h, w in shapes_l], np.float64)
    off = rng.uniform(-spread_px, spread_px, (N, S, 8, 4, 4, 2)) / norm[None, None, None, :, None, :]
    return (ref[None, :, None, None, None, :] + off).astype(np.float32)


@pytest.mark.parametrize("kind,N", [("encoder", 1), ("uniform", 1), ("encoder", 4), ("trained", 4)])
def test_full_size_backward_vs_oracle(kind, N):
    """alo_msda_backward at S = Lq = 22223 (fp32) against the C oracle: encoder-like locations (the tiled path) and uniformly
    random ones (no locality at all: the per-corner path) at N = 1; at N = 4 — BASELINE configs[3]'s per-GPU batch, the launch
    bench.py's training leg times — encoder-like locations and the trained-like offsets of tools/kbench.py."""
    rng = np.random.default_rng(21)
    c = _full_size_case(N, 22223, 13)
    if kind == "encoder":
        c["loc"] = _encoder_like_loc(N, DETR_SHAPES, rng)
    elif kind == "trained":
        import sys, os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        import kbench

        c["loc"] = kbench.msda_inputs(N, 22223, "trained", torch.float32, seed=3)[3].cpu().numpy()
    c["grad_out"] = rng.standard_normal((N, 22223, 256)).astype(np.float32)
    gv, gl, ga = (x.cpu().numpy() for x in hip_backward(c, torch.float32))
    rgv, rgl, rga = O.msda_backward(c["value"].astype(np.float64), c["shapes"], c["level_start"], c["loc"].astype(np.float64),
                                    c["attn"].astype(np.float64), c["grad_out"].astype(np.float64))
    assert np.abs(gv - rgv).max() <= 2e-4 * max(1.0, np.abs(rgv).max())   # sums of up to hundreds of fp32 terms per pixel
    assert np.abs(ga - rga).max() <= 1e-4 * max(1.0, np.abs(rga).max())   # 32-term fp32 dot products of O(1) values
    ok = _away_from_pixel_edges(c["loc"], DETR_SHAPES)
    assert ok.mean() > 0.99 and np.abs((gl - rgl) * ok).max() <= 1e-4 * max(1.0, np.abs(rgl).max())


@pytest.mark.parametrize("spread", [1.5, 6.0, 40.0])
def test_tiled_backward_on_encoder_like_locations(spread):
    """The window-dense route of the tiled fp32 backward (Lq == S, taps within `spread` pixels of the query's own position on
    every level: windows resident in LDS for small spreads, per-corner route for the levels whose window outgrows it), incl.
    taps off the border and queries whose whole window is outside, against the float64 oracle."""
    shapes_l = [(21, 30), (11, 15), (6, 8), (3, 4)]
    rng = np.random.default_rng(int(spread * 10))
    N, M = 2, 8
    loc = _encoder_like_loc(N, shapes_l, rng, spread_px=spread)
    S = loc.shape[1]
    loc[0, :7] += 2.0      # a few queries sample entirely outside
    value = rng.standard_normal((N, S, M, 32)).astype(np.float32)
    attn = rng.random((N, S, M, 4, 4)).astype(np.float32)
    attn /= attn.reshape(N, S, M, 16).sum(-1)[..., None, None]
    go = rng.standard_normal((N, S, M * 32)).astype(np.float32)
    shapes = np.asarray(shapes_l, np.int32)
    c = dict(value=value, shapes=shapes, level_start=level_start(shapes), loc=loc, attn=attn, grad_out=go)
    gv, gl, ga = (x.cpu().numpy() for x in hip_backward(c, torch.float32))
    rgv, rgl, rga = O.msda_backward(value.astype(np.float64), shapes, c["level_start"], loc.astype(np.float64),
                                    attn.astype(np.float64), go.astype(np.float64))
    assert np.abs(gv - rgv).max() <= 1e-4 * max(1.0, np.abs(rgv).max())
    assert np.abs(ga - rga).max() <= 1e-4 * max(1.0, np.abs(rga).max())
    ok = _away_from_pixel_edges(loc, shapes_l)
    assert ok.mean() > 0.98 and np.abs((gl - rgl) * ok).max() <= 1e-4 * max(1.0, np.abs(rgl).max())


def _touched_pixels(loc, shapes_l):
    """(N, S) bool: the pixels some valid sample's in-range corner lands on (cuh:285-291 validity, :38-78 corner guards), any head."""
    N, Lq = loc.shape[:2]
    st = level_start(np.asarray(shapes_l, np.int32))
    S = sum(h * w for h, w in shapes_l)
    hit = np.zeros((N, S), bool)
    for lvl, (h, w) in enumerate(shapes_l):
        x = loc[:, :, :, lvl, :, 0].astype(np.float32) * np.float32(w) - np.float32(0.5)
        y = loc[:, :, :, lvl, :, 1].astype(np.float32) * np.float32(h) - np.float32(0.5)
        valid = (y > -1) & (x > -1) & (y < h) & (x < w)
        x0, y0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
        n_idx = np.broadcast_to(np.arange(N)[:, None, None, None], x.shape)
        for dy in (0, 1):
            for dx in (0, 1):
                yy, xx = y0 + dy, x0 + dx
                ok = valid & (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                hit[n_idx[ok], int(st[lvl]) + yy[ok] * w + xx[ok]] = True
    return hit


def test_tiled_backward_with_nan_outside_the_sampled_footprint():
    """The window-dense backward reads every value row of a tile's bounding box, touched or not (they go straight into the matrix
    operand); a row no sample touches must still not reach any gradient — the reference's guarded loads never see it.  NaN in
    every untouched pixel (inside the windows too): grad_sampling_loc / grad_attn_weight bit-equal to the clean run, grad_value
    equal up to the order of the atomic sums and exactly zero on the untouched pixels."""
    shapes_l = [(21, 30), (11, 15), (6, 8), (3, 4)]
    rng = np.random.default_rng(9)
    N, M = 2, 8
    loc = _encoder_like_loc(N, shapes_l, rng, spread_px=3.0)
    # snap every sample into the cell between pixels 3k and 3k + 1 (both axes): columns / rows 3k + 2 are never a corner, so 5 of 9
    # pixels stay untouched although they lie inside the tiles' windows
    for lvl, (h, w) in enumerate(shapes_l):
        for axis, size in ((0, w), (1, h)):
            px = loc[:, :, :, lvl, :, axis].astype(np.float64) * size - 0.5
            px = 3.0 * np.floor(px / 3.0) + 0.25 + 0.5 * rng.random(px.shape)
            loc[:, :, :, lvl, :, axis] = ((px + 0.5) / size).astype(np.float32)
    S = loc.shape[1]
    value = rng.standard_normal((N, S, M, 32)).astype(np.float32)
    attn = rng.random((N, S, M, 4, 4)).astype(np.float32)
    go = rng.standard_normal((N, S, M * 32)).astype(np.float32)
    shapes = np.asarray(shapes_l, np.int32)
    c = dict(value=value, shapes=shapes, level_start=level_start(shapes), loc=loc, attn=attn, grad_out=go)
    gv, gl, ga = (x.cpu().numpy() for x in hip_backward(c, torch.float32))
    hit = _touched_pixels(loc, shapes_l)
    assert 0.05 < hit.mean() < 0.9
    poisoned = value.copy()
    poisoned[~hit] = np.nan
    pv, pl, pa = (x.cpu().numpy() for x in hip_backward(dict(c, value=poisoned), torch.float32))
    assert np.isfinite(pv).all() and np.isfinite(pl).all() and np.isfinite(pa).all()
    assert np.array_equal(pl, gl) and np.array_equal(pa, ga)
    assert np.abs(pv - gv).max() <= 1e-5 * max(1.0, np.abs(gv).max()) and np.all(pv[~hit] == 0)


def test_backward_is_linear_in_grad_out_at_batch4():
    """Config-4 per-GPU batch (N = 4): grad(2 g1 + g2) == 2 grad(g1) + grad(g2) on all three gradients."""
    rng = np.random.default_rng(22)
    c = _full_size_case(4, 22223, 14)
    c["loc"] = _encoder_like_loc(4, DETR_SHAPES, rng)
    sh, st = dev(c["shapes"]), dev(c["level_start"])
    v, loc, attn = dev(c["value"]), dev(c["loc"]), dev(c["attn"])
    g1 = torch.randn(4, 22223, 256, device=DEV)
    g2 = torch.randn(4, 22223, 256, device=DEV)
    a = alo_hip.msda_backward(v, sh, st, loc, attn, g1, 64)
    b = alo_hip.msda_backward(v, sh, st, loc, attn, g2, 64)
    ab = alo_hip.msda_backward(v, sh, st, loc, attn, 2 * g1 + g2, 64)
    for x, y, z in zip(a, b, ab):
        assert (z - (2 * x + y)).abs().max().item() <= 1e-3 * max(1.0, z.abs().max().item())


def test_tiled_backward_with_two_pyramids_of_the_same_total_size():
    """Two pyramids with the same S but different level shapes, their `spatial_shapes` tensors built fresh per call the way the
    reference transformer builds them (freed and re-allocated, possibly at the same address): each backward must tile the
    queries by ITS pyramid.  (Round-2 advisor finding: a host cache keyed on the storage pointer handed the second call the first
    call's shapes; the kernel now derives the tiling from the device copy and the host copy rides on the tensor object.)"""
    rng = np.random.default_rng(5)
    N, M = 1, 8
    for shapes_l in ([(20, 30), (10, 15), (5, 8), (3, 4)], [(30, 20), (15, 10), (8, 5), (4, 3)], [(20, 30), (10, 15), (5, 8), (3, 4)]):
        loc = _encoder_like_loc(N, shapes_l, rng, spread_px=2.0)
        S = loc.shape[1]
        assert S == 802
        value = rng.standard_normal((N, S, M, 32)).astype(np.float32)
        attn = rng.random((N, S, M, 4, 4)).astype(np.float32)
        attn /= attn.reshape(N, S, M, 16).sum(-1)[..., None, None]
        go = rng.standard_normal((N, S, M * 32)).astype(np.float32)
        shapes = np.asarray(shapes_l, np.int32)
        c = dict(value=value, shapes=shapes, level_start=level_start(shapes), loc=loc, attn=attn, grad_out=go)
        gv, gl, ga = (x.cpu().numpy() for x in hip_backward(c, torch.float32))   # fresh device tensors, dropped on return
        rgv, rgl, rga = O.msda_backward(value.astype(np.float64), shapes, c["level_start"], loc.astype(np.float64),
                                        attn.astype(np.float64), go.astype(np.float64))
        assert np.isfinite(gl).all() and np.isfinite(ga).all()
        assert np.abs(gv - rgv).max() <= 1e-4 * max(1.0, np.abs(rgv).max())
        assert np.abs(ga - rga).max() <= 1e-4 * max(1.0, np.abs(rga).max())
        ok = _away_from_pixel_edges(loc, shapes_l)
        assert np.abs((gl - rgl) * ok).max() <= 1e-4 * max(1.0, np.abs(rgl).max())


def test_tiled_backward_survives_a_host_hint_that_disagrees_with_the_device_shapes():
    """The C ABI's host copy of the shapes only sizes the grid: handing alo_msda_backward_hinted the shapes of ANOTHER pyramid
    with the same S and the same tile count must not change a bit of which queries are served (transposed levels: 25 x 38 tiles
    either way)."""
    import ctypes

    rng = np.random.default_rng(6)
    shapes_l = [(20, 28), (10, 14), (5, 8), (3, 4)]
    wrong = [(28, 20), (14, 10), (8, 5), (4, 3)]
    loc = _encoder_like_loc(1, shapes_l, rng, spread_px=2.0)
    S = loc.shape[1]
    value = dev(rng.standard_normal((1, S, 8, 32)).astype(np.float32))
    attn = rng.random((1, S, 8, 4, 4)).astype(np.float32)
    attn = dev(attn / attn.reshape(1, S, 8, 16).sum(-1)[..., None, None])
    go = dev(rng.standard_normal((1, S, 256)).astype(np.float32))
    shapes = np.asarray(shapes_l, np.int32)
    sh, st, tl = dev(shapes), dev(level_start(shapes)), dev(loc)
    outs = []
    for hint_shapes in (shapes_l, wrong):
        gv, gl, ga = torch.empty_like(value), torch.full_like(tl, float("nan")), torch.full_like(attn, float("nan"))
        hint = (ctypes.c_int32 * 8)(*[int(v) for hw in hint_shapes for v in hw])
        rc = alo_hip.lib().alo_msda_backward_hinted(
            *(ctypes.c_void_p(t.data_ptr()) for t in (value, sh, st, tl, attn, go, gv, gl, ga)), 1, S, 8, 32, 4, S, 4,
            alo_hip.ALO_F32, alo_hip.ALO_F32, hint, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        outs.append((gv, gl, ga))
    assert torch.isfinite(outs[1][1]).all() and torch.isfinite(outs[1][2]).all()   # every query was served
    # the wide path (msda_bwd_wide.hip) answers a disagreeing hint by sending every sample down the per-corner route: the same sums
    # in another order, so the comparison is to fp32 rounding, not to the bit
    for i in (0, 1, 2):
        assert (outs[0][i] - outs[1][i]).abs().max().item() <= 1e-5 * max(1.0, outs[0][i].abs().max().item())


def test_tiled_backward_with_a_host_hint_that_under_counts_the_device_tiles():
    """Round-3 advisor finding: a host copy with the same S but FEWER 4x4 tiles than the device shapes have ((5, 8) -> 4 tiles,
    (2, 20) -> 5) sized a grid that left the device's last tiles — their queries' grad_loc / grad_attn rows — unserved, silently.
    The kernel now compares the two counts and, on an under-count, groups the whole launch 16 queries in a row (the host's grid
    always holds ceil(Lq / 16) groups): every query served, results those of the oracle."""
    import ctypes

    rng = np.random.default_rng(16)
    shapes_l = [(20, 28), (10, 14), (2, 20), (3, 4)]
    under = [(20, 28), (10, 14), (5, 8), (3, 4)]       # same S; level 2: 2 x 2 = 4 tiles instead of 1 x 5 = 5
    assert sum(h * w for h, w in shapes_l) == sum(h * w for h, w in under)
    tiles = lambda sh: sum(-(-h // 4) * -(-w // 4) for h, w in sh)   # noqa: E731
    assert tiles(under) < tiles(shapes_l)
    loc = _encoder_like_loc(1, shapes_l, rng, spread_px=2.0)
    S = loc.shape[1]
    value_np = rng.standard_normal((1, S, 8, 32)).astype(np.float32)
    attn_np = rng.random((1, S, 8, 4, 4)).astype(np.float32)
    attn_np /= attn_np.reshape(1, S, 8, 16).sum(-1)[..., None, None]
    go_np = rng.standard_normal((1, S, 256)).astype(np.float32)
    value, attn, go, tl = dev(value_np), dev(attn_np), dev(go_np), dev(loc)
    shapes = np.asarray(shapes_l, np.int32)
    sh, st = dev(shapes), dev(level_start(shapes))
    gv, gl, ga = torch.empty_like(value), torch.full_like(tl, float("nan")), torch.full_like(attn, float("nan"))
    hint = (ctypes.c_int32 * 8)(*[int(v) for hw in under for v in hw])
    rc = alo_hip.lib().alo_msda_backward_hinted(
        *(ctypes.c_void_p(t.data_ptr()) for t in (value, sh, st, tl, attn, go, gv, gl, ga)), 1, S, 8, 32, 4, S, 4,
        alo_hip.ALO_F32, alo_hip.ALO_F32, hint, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(gl).all() and torch.isfinite(ga).all()          # no query was left out
    rgv, rgl, rga = O.msda_backward(value_np.astype(np.float64), shapes, level_start(shapes), loc.astype(np.float64),
                                    attn_np.astype(np.float64), go_np.astype(np.float64))
    assert np.abs(gv.cpu().numpy() - rgv).max() <= 2e-4 * max(1.0, np.abs(rgv).max())
    assert np.abs(ga.cpu().numpy() - rga).max() <= 1e-4 * max(1.0, np.abs(rga).max())
    ok = _away_from_pixel_edges(loc, shapes_l)
    assert np.abs((gl.cpu().numpy() - rgl) * ok).max() <= 1e-4 * max(1.0, np.abs(rgl).max())


# ---- coarse levels resident in LDS (alo_msda_forward_fused_hm_resident) ---------------------------------------------------------------
RESIDENT_CASES = [  # (N, shapes, Lq or None = S, ref_dim)                        which route
    (2, [(96, 128), (48, 64), (24, 32), (12, 16)], None, 2),   # a 1024 x 768 frame's pyramid: levels 2-3 = 960 rows resident
    (2, [(40, 50), (20, 25), (10, 13), (5, 7)], None, 2),      # levels 2-3 resident (165 rows), 3 workgroups per slab, ragged tail run
    (1, [(40, 50), (20, 25), (10, 13), (5, 7)], 1000, 4),      # free queries with box reference points
    (3, [(64, 80), (37, 37), (35, 37), (10, 10)], 2500, 2),    # levels 2-3 = 1395 rows: just fits (90 032 bytes of LDS for the image)
    (1, [(30, 40), (15, 20), (8, 10), (4, 5)], None, 2),       # N * M = 8 slabs: many workgroups per slab
]


def _resident_case(N, shapes_l, Lq, ref_dim, seed):
    M, D, L, P = 8, 32, 4, 4
    gen = torch.Generator(device=DEV).manual_seed(seed)
    shapes = torch.tensor(shapes_l, dtype=torch.int32, device=DEV)
    shapes._alo_shapes = [tuple(hw) for hw in shapes_l]
    start = dev(level_start(shapes_l))
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    Lq = S if Lq is None else Lq
    value = torch.randn(N, S, M, D, generator=gen, device=DEV).bfloat16()
    mask = torch.rand(N, S, generator=gen, device=DEV) < 0.1
    offsets = (torch.randn(N, Lq, M, L, P, 2, generator=gen, device=DEV) * 4.0).bfloat16()
    logits = (torch.randn(N, Lq, M, L * P, generator=gen, device=DEV) * 2.0).bfloat16()
    ref = torch.rand(N, Lq, L, ref_dim, generator=gen, device=DEV) * 1.2 - 0.1     # some reference points outside the map
    if ref_dim == 4:
        ref[..., 2:] = ref[..., 2:].abs() * 0.4
    return value, mask, offsets, logits, ref, shapes, start


@pytest.mark.parametrize("case", RESIDENT_CASES, ids=lambda c: f"N{c[0]}-{c[1][2][0]}x{c[1][2][1]}-Lq{c[2]}-ref{c[3]}")
def test_resident_forward_is_bit_identical_to_the_plain_head_major_kernel(case):
    """Same descriptors, same products, same accumulation order: serving the coarse levels from LDS must not change a bit —
    borders, samples outside the map (zero row in LDS / buffer range check), padded pixels and the ragged last run included.
    And against the float64 oracle directly."""
    N, shapes_l, Lq, ref_dim = case
    value, mask, offsets, logits, ref, shapes, start = _resident_case(N, shapes_l, Lq, ref_dim, 41 + N)
    vhm = alo_hip.value_head_major(value, mask)
    with alo_hip.LaunchTimer() as timer:
        got = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident="always")   # small launches too
    assert any(k.startswith("msda_fwd_fused_resident") for k in timer.summary()), timer.summary().keys()
    want = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=False)
    assert torch.equal(got, want)
    sel = slice(0, None, 7)
    loc, attn = _prologue_in_torch(offsets[:, sel].float(), logits[:, sel].float(), ref[:, sel], shapes, 4)
    exact = O.msda_forward(value.masked_fill(mask[..., None, None], 0).double().cpu().numpy(), shapes.cpu().numpy(),
                           start.cpu().numpy(), loc.double().cpu().numpy(), attn.double().cpu().numpy())
    err = np.abs(got[:, sel].double().cpu().numpy() - exact)
    assert np.all(err <= np.abs(exact) * 2.0 ** -8 + 2e-5)


def test_resident_forward_with_nan_outside_the_sampled_footprint():
    """NaN in pixels no sample touches — inside the LDS-resident levels too — must not reach any output: corners outside the map
    read the all-zero LDS row, never a neighbouring pixel."""
    N, shapes_l = 1, [(40, 50), (20, 25), (10, 13), (5, 7)]
    value, mask, offsets, logits, ref, shapes, start = _resident_case(N, shapes_l, None, 2, 77)
    offsets.zero_()                      # every query samples exactly its reference point on every level
    ref[:] = 0.25                        # ... which lies in the interior of every level
    vhm = alo_hip.value_head_major(value, None)
    clean = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident="always")
    poisoned = vhm.clone()
    st = level_start(shapes_l)
    for lvl, (h, w) in enumerate(shapes_l):
        keep = torch.zeros(h, w, dtype=torch.bool, device=DEV)
        y, x = int(np.floor(0.25 * h - 0.5)), int(np.floor(0.25 * w - 0.5))
        keep[y:y + 2, x:x + 2] = True
        rows = torch.nonzero(~keep.view(-1)).view(-1) + int(st[lvl])
        poisoned[:, :, rows] = float("nan")
    with alo_hip.LaunchTimer() as timer:
        got = alo_hip.msda_forward_fused_hm(poisoned, shapes, start, offsets, logits, ref, resident="always")
    assert any(k.startswith("msda_fwd_fused_resident") for k in timer.summary())
    assert torch.isfinite(got.float()).all() and torch.equal(got, clean)


def test_resident_forward_ignores_a_host_copy_that_disagrees_with_the_device_metadata():
    """The host copy of the shapes picks the resident levels, lays out the LDS image and sizes the grid; a WRONG one (another
    pyramid with the same S) must only cost speed: the kernel compares it with the device copy and serves every level through the
    buffer path.  (The third copy is right about the resident levels and wrong about the others: that one stays resident.)"""
    import ctypes

    N, shapes_l = 2, [(40, 50), (20, 25), (10, 13), (5, 7)]
    value, mask, offsets, logits, ref, shapes, start = _resident_case(N, shapes_l, None, 2, 55)
    vhm = alo_hip.value_head_major(value, mask)
    want = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=False)
    S, Lq = vhm.shape[2], offsets.shape[1]
    # true shapes: (40, 50), (20, 25), (10, 13), (5, 7); every wrong copy has the same total S = 2665
    for wrong in ([(40, 50), (20, 25), (13, 10), (5, 7)], [(40, 50), (20, 25), (10, 13), (7, 5)], [(50, 40), (25, 20), (10, 13), (5, 7)],
                  [(40, 50), (25, 20), (5, 26), (35, 1)]):
        assert sum(h * w for h, w in wrong) == S
        out = torch.full_like(want, float("nan"))
        hint = (ctypes.c_int32 * 8)(*[v for hw in wrong for v in hw])
        rc = alo_hip.lib().alo_msda_forward_fused_hm_resident(
            *(ctypes.c_void_p(t.data_ptr()) for t in (vhm, shapes, start, offsets, logits)), 8 * 16 * 2, 8 * 16,
            ctypes.c_void_p(ref.data_ptr()), ctypes.c_void_p(out.data_ptr()), N, S, 8, 32, 4, Lq, 4, 2, alo_hip.ALO_BF16, hint,
            alo_hip.RESIDENT_ALWAYS, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, alo_hip.lib().alo_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out, want), wrong


def test_resident_policy_auto_takes_the_kernel_that_is_faster_at_the_launch_size():
    """ALO_RESIDENT_AUTO: the resident kernel from one 16-query run per wave of the chip upwards (N * M * ceil(Lq / 16) >= CUs * 12),
    the plain head-major kernel below (tools/exp/res_sweep.py: 1.2-1.6 x slower there), and never when levels 2-3 do not fit in LDS;
    ALO_RESIDENT_ALWAYS wherever the levels fit.  Same bits in every case."""
    import ctypes

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    small, big, huge = [(32, 40), (16, 20), (8, 10), (4, 5)], [(64, 84), (32, 42), (16, 21), (8, 11)], [(150, 200), (75, 100), (38, 50), (19, 25)]
    for N, shapes_l, Lq, auto, always in ((1, small, None, 0, 2), (8, big, None, 2, 2), (1, huge, 6000, 0, 0)):
        S = sum(h * w for h, w in shapes_l)
        host = (ctypes.c_int32 * 8)(*[v for hw in shapes_l for v in hw])
        lq = S if Lq is None else Lq
        runs = N * 8 * ((lq + 15) // 16)
        assert (runs >= cus * 12) == bool(auto) or shapes_l is huge
        assert alo_hip.lib().alo_msda_resident_levels(host, N, S, 8, 4, lq, alo_hip.RESIDENT_AUTO) == auto
        assert alo_hip.lib().alo_msda_resident_levels(host, N, S, 8, 4, lq, alo_hip.RESIDENT_ALWAYS) == always
        value, mask, offsets, logits, ref, shapes, start = _resident_case(N, shapes_l, Lq, 2, 5)
        vhm = alo_hip.value_head_major(value, mask)
        outs = []
        for mode, expect in ((True, auto), ("always", always), (False, 0)):
            with alo_hip.LaunchTimer() as timer:
                outs.append(alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=mode))
            assert any(k.startswith("msda_fwd_fused_resident") for k in timer.summary()) == bool(expect), (N, shapes_l, mode)
        assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])

