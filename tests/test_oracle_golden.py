"""The oracle (C and torch restatements) against the golden vectors produced by the reference itself.  CPU only."""
import numpy as np
import pytest
import torch

import oracle as O
import torch_ref as T


def test_g1_optest_case(golden):
    g = golden("g1_msda_optest.npz")
    f64 = [g[k].astype(np.float64) for k in ("value", "loc", "attn")]
    out = O.msda_forward(f64[0], g["shapes"], g["level_start"], f64[1], f64[2])
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-12, atol=1e-15)
    out32 = O.msda_forward(g["value_b"], g["shapes"], g["level_start"], g["loc_b"], g["attn_b"])
    # the reference's own fp32 bar is rtol 1e-2 / atol 1e-3 (ops/test.py:83); the restatement is far inside it
    np.testing.assert_allclose(out32, g["out_f32"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("D", [30, 32, 64, 71])
def test_g2_gradcheck_set(golden, D):
    g = golden(f"g2_msda_grad_D{D}.npz")
    v, l, a, go = (g[k].astype(np.float64) for k in ("value", "loc", "attn", "grad_out"))
    np.testing.assert_allclose(O.msda_forward(v, g["shapes"], g["level_start"], l, a), g["out"], rtol=1e-12, atol=1e-15)
    gv, gl, ga = O.msda_backward(v, g["shapes"], g["level_start"], l, a, go)
    np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=1e-10, atol=1e-14)


def test_g3_borders_and_out_of_range(golden):
    g = golden("g3_msda_medium.npz")
    v, l, a, go = (g[k].astype(np.float64) for k in ("value", "loc", "attn", "grad_out"))
    np.testing.assert_allclose(O.msda_forward(v, g["shapes"], g["level_start"], l, a), g["out"], rtol=1e-12, atol=1e-13)
    gv, gl, ga = O.msda_backward(v, g["shapes"], g["level_start"], l, a, go)
    # the fixture stores the gradients as float32
    np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=1e-6, atol=2e-5)
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=1e-6, atol=1e-6)
    # the torch restatement of the reference's CPU path
    out_t = T.msda_core(torch.from_numpy(v), g["shapes"], torch.from_numpy(l), torch.from_numpy(a)).numpy()
    np.testing.assert_allclose(out_t, g["out"], rtol=1e-12, atol=1e-13)


def test_g8_known_answers(golden):
    g = golden("g8_known_answers.npz")
    out = O.msda_forward(g["value"], g["shapes"], g["level_start"], g["loc"], g["attn"])
    np.testing.assert_allclose(out.ravel(), g["expected"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(g["out"].ravel(), g["expected"], rtol=0, atol=1e-15)  # the reference agrees


def test_g17_the_reference_trt_plugin_test_case(golden):
    """N, M, D = 1, 8, 32; Lq = 12000; square levels 64 .. 8 (torch2trt/plugins/ms_deform_im2col/test.py:103-121): the C oracle on the
    kept queries against the reference's float64 and float32 outputs."""
    from helpers import g17_inputs

    g = golden("g17_msda_trt_plugin_case.npz")
    drawn = g17_inputs(g)
    assert drawn is not None, "torch's CPU generator draws another stream than the build the fixture was made with"
    value, loc, attn = drawn
    keep = g["keep"]
    l, a = loc[:, keep].numpy(), attn[:, keep].numpy()
    out = O.msda_forward(value.numpy().astype(np.float64), g["shapes"], g["level_start"], l.astype(np.float64), a.astype(np.float64))
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-12, atol=1e-14)
    out32 = O.msda_forward(value.numpy(), g["shapes"], g["level_start"], l, a)
    np.testing.assert_allclose(out32, g["out_f32"], rtol=1e-2, atol=1e-3)   # the reference's fp32 bar
    np.testing.assert_allclose(out32, g["out_f64"], rtol=0, atol=5e-6)


def test_f32_oracle_tracks_f64(golden):
    g = golden("g3_msda_medium.npz")
    args32 = [g[k].astype(np.float32) for k in ("value", "loc", "attn")]
    out = O.msda_forward(args32[0], g["shapes"], g["level_start"], args32[1], args32[2])
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=2e-5)


def test_g6_corr_pyramid_and_lookup(golden):
    g = golden("g6_corr.npz")
    pyr = O.corr_pyramid(g["f1"], g["f2"])
    for lvl in range(4):
        assert pyr[lvl].shape == g[f"lvl{lvl}"].shape
        # fp32 matmul summation order vs one rounding of a double accumulation
        np.testing.assert_allclose(pyr[lvl], g[f"lvl{lvl}"], rtol=0, atol=1e-5)
    ref_pyr = [g[f"lvl{lvl}"] for lvl in range(4)]
    for k in "abc":
        np.testing.assert_allclose(O.corr_lookup(ref_pyr, g["coords_" + k]), g["out_" + k], rtol=0, atol=1e-6)
        np.testing.assert_allclose(O.corr_lookup(pyr, g["coords_" + k]), g["out_" + k], rtol=0, atol=1e-5)


def test_g6_odd_sizes_batched_radius3(golden):
    g = golden("g6_corr.npz")
    pyr = O.corr_pyramid(g["f1o"], g["f2o"])
    assert [p.shape[2:] for p in pyr] == [(17, 18), (8, 9), (4, 4), (2, 2)]
    for lvl in range(4):
        np.testing.assert_allclose(pyr[lvl], g[f"lvl{lvl}o"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(O.corr_lookup(pyr, g["coords_o"], 3), g["out_o"], rtol=0, atol=1e-5)


def test_torch_restatement_of_corrblock(golden):
    g = golden("g6_corr.npz")
    blk = T.CorrBlockRef(torch.from_numpy(g["f1"]), torch.from_numpy(g["f2"]))
    for lvl in range(4):
        np.testing.assert_array_equal(blk.corr_pyramid[lvl].numpy(), g[f"lvl{lvl}"])
    for k in "abc":
        np.testing.assert_array_equal(blk(torch.from_numpy(g["coords_" + k])).numpy(), g["out_" + k])
    blk = T.CorrBlockRef(torch.from_numpy(g["f1o"]), torch.from_numpy(g["f2o"]), radius=3)
    np.testing.assert_array_equal(blk(torch.from_numpy(g["coords_o"])).numpy(), g["out_o"])


def test_lookup_window_axis_order():
    """First window axis moves x, second moves y (the reference's meshgrid(dy, dx) quirk, corr.py:37-44)."""
    H, W, r = 8, 10, 1
    lvl0 = np.zeros((H * W, 1, H, W), np.float32)
    lvl0[:, 0] = np.arange(H * W, dtype=np.float32).reshape(H, W)  # every query sees value = 10*y + x
    pyr = [lvl0]
    coords = np.zeros((1, 2, H, W), np.float32)
    coords[0, 0], coords[0, 1] = 4.0, 3.0  # every query looks at (x=4, y=3)
    out = O.corr_lookup(pyr, coords, r)  # channels: a*3 + c
    centre = 3 * W + 4
    assert out[0, 1 * 3 + 1, 0, 0] == centre
    assert out[0, 2 * 3 + 1, 0, 0] == centre + 1  # a = 2 -> x + 1
    assert out[0, 1 * 3 + 2, 0, 0] == centre + W  # c = 2 -> y + 1
