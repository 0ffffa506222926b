"""Host-side mirror of the reference's operator API: names, schema, error behaviour, module arithmetic.  CPU only."""
import numpy as np
import pytest
import torch

from alonet.deformable_detr.ops.functions import (
    MSDeformAttnFunction,
    load_MultiScaleDeformableAttention,
    load_ops,
    ms_deform_attn_core_pytorch,
)
from alonet.deformable_detr.ops.modules import MSDeformAttn
from alonet.raft.corr import CorrBlock
from alonet.raft.utils.utils import coords_grid

t = torch.from_numpy


def test_dispatcher_ops_registered_with_reference_schema():
    load_MultiScaleDeformableAttention()
    load_ops()  # idempotent
    fwd = torch.ops.alonet_custom.ms_deform_attn_forward.default._schema
    bwd = torch.ops.alonet_custom.ms_deform_attn_backward.default._schema
    assert [a.name for a in fwd.arguments] == ["value", "spatial_shapes", "level_start_index", "sampling_loc",
                                                "attn_weight", "im2col_step"]
    assert [a.name for a in bwd.arguments][-2:] == ["grad_output", "im2col_step"]
    assert str(fwd.returns[0].type) == "Tensor" and str(bwd.returns[0].type) == "List[Tensor]"


def test_cpu_tensors_raise_like_the_reference(golden):
    g = golden("g1_msda_optest.npz")
    args = (t(g["value"]), t(g["shapes"]), t(g["level_start"]), t(g["loc"]), t(g["attn"]), 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(*args)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        torch.ops.alonet_custom.ms_deform_attn_forward(*args)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        CorrBlock(torch.zeros(1, 8, 16, 16), torch.zeros(1, 8, 16, 16))


def test_meta_kernels_give_output_shapes():
    v = torch.empty(2, 30, 4, 8, device="meta")
    loc = torch.empty(2, 7, 4, 2, 3, 2, device="meta")
    attn = torch.empty(2, 7, 4, 2, 3, device="meta")
    sh = torch.empty(2, 2, dtype=torch.int32, device="meta")
    st = torch.empty(2, dtype=torch.int32, device="meta")
    out = torch.ops.alonet_custom.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    assert out.shape == (2, 7, 32)
    gv, gl, ga = torch.ops.alonet_custom.ms_deform_attn_backward(v, sh, st, loc, attn, out, 64)
    assert gv.shape == v.shape and gl.shape == loc.shape and ga.shape == attn.shape


def test_tracing_branch_matches_reference_core(golden):
    g = golden("g3_msda_medium.npz")
    out = ms_deform_attn_core_pytorch(t(g["value"]).double(), t(g["shapes"]), t(g["loc"]).double(), t(g["attn"]).double())
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-12, atol=1e-13)


def _module_from_fixture(g):
    d_model, n_levels, n_heads, n_points = (int(x) for x in g["cfg"])
    m = MSDeformAttn(d_model, n_levels, n_heads, n_points).double()
    missing = m.load_state_dict({k[3:]: t(g[k]) for k in g.files if k.startswith("sd.")})
    assert not missing.missing_keys and not missing.unexpected_keys  # state-dict keys are the reference's
    return m


def test_module_arithmetic_against_reference_module(golden):
    g = golden("g4_msda_module.npz")
    m = _module_from_fixture(g)
    common = (t(g["src"]), t(g["shapes"]), t(g["level_start"]))
    with torch.no_grad():
        out2 = m(t(g["query"]), t(g["ref2"]), *common, t(g["mask"]), is_tracing=None)
        out4 = m(t(g["query"]), t(g["ref4"]), *common, t(g["mask"]), is_tracing=None)
        out2n = m(t(g["query"]), t(g["ref2"]), *common, None, is_tracing=None)
    np.testing.assert_allclose(out2.numpy(), g["out2"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(out4.numpy(), g["out4"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(out2n.numpy(), g["out2_nomask"], rtol=1e-12, atol=1e-13)
    with pytest.raises(ValueError, match="must be 2 or 4"):
        m(t(g["query"]), t(g["ref2"])[..., :1], *common, None, is_tracing=None)


def test_module_default_init_is_the_reference_ring():
    m = MSDeformAttn(256, 4, 8, 4)
    assert set(k for k, _ in m.named_parameters()) == {
        "sampling_offsets.weight", "sampling_offsets.bias", "attention_weights.weight", "attention_weights.bias",
        "value_proj.weight", "value_proj.bias", "output_proj.weight", "output_proj.bias"}
    bias = m.sampling_offsets.bias.detach().view(8, 4, 4, 2)
    assert torch.all(m.sampling_offsets.weight == 0) and torch.all(m.attention_weights.weight == 0)
    np.testing.assert_allclose(bias[0, 0, :, 0].numpy(), [1, 2, 3, 4], atol=1e-6)  # head 0 points along +x
    np.testing.assert_allclose(bias[2, 1, :, 1].numpy(), [1, 2, 3, 4], atol=1e-6)  # head 2 points along +y
    np.testing.assert_allclose(bias[4, 3, :, 0].numpy(), [-1, -2, -3, -4], atol=1e-6)
    with pytest.raises(ValueError):
        MSDeformAttn(30, 4, 8, 4)


def test_coords_grid_layout():
    g = coords_grid(2, 3, 5)
    assert g.shape == (2, 2, 3, 5)
    assert torch.equal(g[0, 0, 1], torch.arange(5.0)) and torch.equal(g[1, 1, :, 2], torch.arange(3.0))


# ---- host-side plumbing of the inference fast paths (CPU) ------------------------------------------------------------
def test_conv1x1_as_gemm_equals_conv2d_on_nhwc_rows():
    import torch.nn.functional as F

    from alonet.detr.backbone import conv1x1_as_gemm

    torch.manual_seed(0)
    x = torch.randn(2, 12, 9, 11).contiguous(memory_format=torch.channels_last)
    w, b = torch.randn(20, 12, 1, 1), torch.randn(20)
    for stride in ((1, 1), (2, 2)):
        for relu in (False, True):
            for bias in (b, None):
                want = F.conv2d(x, w, bias, stride)
                want = want.relu() if relu else want
                got = conv1x1_as_gemm(x, w, bias, stride, relu=relu)
                assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
                assert (got - want).abs().max().item() <= 1e-5


def test_level_geometry_is_cached_and_carries_host_copies():
    from alonet.deformable_detr.deformable_transformer import _level_geometry

    shapes = ((5, 7), (3, 4))
    a = _level_geometry(shapes, torch.device("cpu"))
    b = _level_geometry(shapes, torch.device("cpu"))
    assert a[0] is b[0] and a[1] is b[1]  # no rebuild, no host->device copy per forward
    assert a[0].dtype == torch.int32 and a[0].tolist() == [[5, 7], [3, 4]] and a[1].tolist() == [0, 35]
    assert a[0]._alo_total == 47 and a[0]._alo_shapes == [(5, 7), (3, 4)]
    assert _level_geometry(((5, 7),), torch.device("cpu"))[0] is not a[0]


def test_joiner_skips_requested_positional_encodings():
    import aloscene
    from alonet.deformable_detr.backbone import Backbone, Joiner
    from alonet.transformers import PositionEmbeddingSine

    torch.manual_seed(0)
    joiner = Joiner(Backbone("resnet50", False, True, False), PositionEmbeddingSine(128, normalize=True, center=True)).eval()
    frames = aloscene.Frame.batch_list([aloscene.Frame(torch.rand(3, 64, 96) * 255).norm_resnet()])
    with torch.no_grad():
        feats, pos = joiner(frames, skip_pos_levels=(0,))
        feats_all, pos_all = joiner(frames)
    assert pos[0] is None and all(p is not None for p in pos[1:]) and len(feats) == len(feats_all) == 4
    assert all(torch.equal(a, b) for a, b in zip(pos[1:], pos_all[1:])) and pos_all[0].shape[-2:] == feats_all[0][0].shape[-2:]


def test_cached_derived_weights_follow_parameter_updates():
    from alonet.raft.update import _cached

    lin = torch.nn.Linear(3, 2)
    first = _cached(lin, "_double", (lin.weight,), lambda: 2 * lin.weight)
    assert _cached(lin, "_double", (lin.weight,), lambda: 2 * lin.weight) is first
    with torch.no_grad():
        lin.weight.add_(1.0)  # in-place update bumps the version counter
    second = _cached(lin, "_double", (lin.weight,), lambda: 2 * lin.weight)
    assert second is not first and torch.equal(second, 2 * lin.weight)
