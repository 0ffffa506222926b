"""The C-ABI library loads without a GPU and exports exactly what include/alo_hotpath.h declares.  CPU only."""
import ctypes
import os
import re

import pytest

import alo_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "alo_hotpath.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alo_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    assert declared_functions() == sorted(
        ["alo_abi_version", "alo_last_error", "alo_msda_forward", "alo_msda_forward_fused", "alo_msda_backward", "alo_msda_backward_hinted", "alo_msda_backward_path", "alo_corr_level_shape",
         "alo_corr_build_workspace_bytes", "alo_corr_build", "alo_corr_lookup", "alo_corr_lookup_backward", "alo_corr_lookup_backward_coords", "alo_add_layernorm", "alo_bias_act", "alo_msda_forward_fused_hm", "alo_msda_forward_fused_hm_rows", "alo_msda_forward_fused_hm_resident", "alo_msda_resident_levels", "alo_value_head_major", "alo_bias_act_nchw", "alo_gru_gate", "alo_gru_update", "alo_pos_sine_flat", "alo_linear_shortk", "alo_ffn256", "alo_pack_mfma_b", "alo_value_proj_head_major", "alo_conv3x3_nhwc", "alo_conv3x3_workspace_bytes", "alo_stem_conv_pool", "alo_mask_pyramid", "alo_panoptic_onehot", "alo_encoder_reference_points", "alo_linear_packed", "alo_conv1x1_nhwc", "alo_groupnorm_rows", "alo_groupnorm_rows_workspace_bytes", "alo_groupnorm_rows_act", "alo_upsample_add_nhwc", "alo_conv3x3_small_nhwc"]
    )


def test_library_exports_every_declared_symbol():
    lib = alo_hip.lib()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} missing from {alo_hip.LIB_PATH}"
    assert lib.alo_abi_version() == 2
    assert lib.alo_last_error() is not None


def test_no_torch_symbols_in_the_abi():
    """The boundary is plain C: the shared object must not link libtorch / libc10."""
    import subprocess

    out = subprocess.run(["readelf", "-d", alo_hip.LIB_PATH], capture_output=True, text=True).stdout
    needed = "\n".join(line for line in out.splitlines() if "NEEDED" in line)   # library names only, not addresses
    assert "torch" not in needed and "c10" not in needed
    assert "libamdhip64" in needed


def test_pyramid_shape_helpers():
    assert alo_hip.corr_level_shapes(90, 160, 4) == [(90, 160), (45, 80), (22, 40), (11, 20)]
    assert alo_hip.corr_level_shapes(17, 18, 4) == [(17, 18), (8, 9), (4, 4), (2, 2)]
    lib = alo_hip.lib()
    # fp16-split copies of both feature maps (2 terms x 2 bytes per channel, 16-channel slices) + one int32 power of two per pixel of
    # each map (padded by a row tile of 256 + 4 entries: the last tile reads whole) + fmap2's per-item magnitudes (256 bytes) + the fp32
    # pooled chain, the split copy and the per-pixel exponents of level 3 for a 4-level pyramid; every piece rounded up to 256 bytes
    up = lambda x: (x + 255) // 256 * 256  # noqa: E731
    kexp = lambda B, n: up((B * n + 256 + 4) * 4)  # noqa: E731
    split0 = 4 * 16 * 2 * 14400 * 16 * 2
    assert lib.alo_corr_build_workspace_bytes(4, 256, 90, 160, 3) == 2 * split0 + 2 * kexp(4, 14400) + 256
    assert lib.alo_corr_build_workspace_bytes(4, 256, 90, 160, 4) == (2 * split0 + 2 * kexp(4, 14400) + 256 + 4 * 256 * (3600 + 880 + 220) * 4
                                                                     + 4 * 16 * 2 * 220 * 16 * 2 + kexp(4, 220))
    assert lib.alo_corr_build_workspace_bytes(1, 8, 16, 16, 1) == 2 * 1 * 1 * 2 * 256 * 16 * 2 + 2 * kexp(1, 256) + 256


def test_argument_errors_are_reported_before_any_launch():
    lib = alo_hip.lib()
    rc = lib.alo_msda_forward(None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, 0, None)
    assert rc == 1 and b"null pointer" in lib.alo_last_error()
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    rc = lib.alo_msda_forward(one, one, one, one, one, one, 1, 1, 1, 0, 1, 1, 1, 0, 0, None)
    assert rc == 1 and b"positive" in lib.alo_last_error()
    rc = lib.alo_msda_forward(one, one, one, one, one, one, 1, 1, 1, 1, 1, 1, 1, 2, 2, None)
    assert rc == 2 and b"dtype" in lib.alo_last_error()
    rc = lib.alo_add_layernorm(one, None, one, one, one, one, None, 4, 256, 1e-5, 0, None)
    assert rc == 1 and b"go together" in lib.alo_last_error()
    rc = lib.alo_add_layernorm(one, None, one, one, one, None, None, 4, 258, 1e-5, 0, None)
    assert rc == 1 and b"multiple of 4" in lib.alo_last_error()
    rc = lib.alo_bias_act(one, one, None, one, 4, 64, 1, 1, None)
    assert rc == 2 and b"dtype" in lib.alo_last_error()
    rc = lib.alo_msda_forward(one, one, one, one, one, one, 1, 1, 1, 1, 33, 1, 1, 0, 0, None)
    assert rc == 2 and b"levels" in lib.alo_last_error()
    ptrs = (ctypes.c_void_p * 4)(16, 16, 16, 16)
    rc = lib.alo_corr_lookup(ptrs, one, one, 1, 8, 8, 9, 4, None)
    assert rc == 2 and b"radius" in lib.alo_last_error()
    rc = lib.alo_corr_lookup(ptrs, one, one, 1, 8, 8, 4, 4, None)  # level 3 of an 8x8 grid is 1x1
    assert rc == 1 and b"level 3" in lib.alo_last_error()
    rc = lib.alo_corr_build(one, one, ptrs, None, 0, 1, 8, 16, 16, 4, None)
    assert rc == 1 and b"workspace" in lib.alo_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(alo_hip, "_lib", None)
    monkeypatch.setattr(alo_hip, "LIB_PATH", str(tmp_path / "libalo_hotpath.so"))
    with pytest.raises(alo_hip.HotpathUnavailable, match="no CPU fallback"):
        alo_hip.lib()


def test_resident_forward_plan_is_host_logic():
    """alo_msda_resident_levels: which launches the LDS-resident forward serves (csrc/msda.hip resident_plan; 256 CUs assumed when no
    device answers).  AUTO = at least one 16-query run per wave of the chip and levels 2-3 within the LDS image; ALWAYS = they fit."""
    lib = alo_hip.lib()

    def plan(shapes, N, Lq, policy):
        host = (ctypes.c_int32 * 8)(*[v for hw in shapes for v in hw])
        S = sum(h * w for h, w in shapes)
        return lib.alo_msda_resident_levels(host, N, S, 8, 4, S if Lq is None else Lq, policy)

    detr = [(100, 167), (50, 84), (25, 42), (13, 21)]
    assert plan(detr, 8, None, alo_hip.RESIDENT_AUTO) == 2 and plan(detr, 1, None, alo_hip.RESIDENT_AUTO) == 2
    assert plan(detr, 8, 300, alo_hip.RESIDENT_AUTO) == 0           # the decoder's 300 queries: 64 x 19 runs < 256 x 12
    assert plan(detr, 8, 300, alo_hip.RESIDENT_ALWAYS) == 2          # ... possible all the same (19 runs per slab >= 12 waves)
    assert plan(detr, 8, 100, alo_hip.RESIDENT_ALWAYS) == 0          # 7 runs per slab: not even one per wave
    small = [(32, 40), (16, 20), (8, 10), (4, 5)]
    assert plan(small, 1, None, alo_hip.RESIDENT_AUTO) == 0 and plan(small, 1, None, alo_hip.RESIDENT_ALWAYS) == 2
    assert plan(small, 4, None, alo_hip.RESIDENT_AUTO) == 2          # 32 x 107 = 3424 runs >= 3072
    big = [(150, 200), (75, 100), (38, 50), (19, 25)]                # level 2 alone is 1900 pixels: does not fit
    assert plan(big, 8, None, alo_hip.RESIDENT_AUTO) == 0 and plan(big, 8, None, alo_hip.RESIDENT_ALWAYS) == 0
    assert plan([(64, 80), (37, 37), (35, 37), (10, 10)], 3, 2500, alo_hip.RESIDENT_ALWAYS) == 2   # 89 640 of 90 032 bytes
    assert plan([(64, 80), (37, 37), (36, 37), (10, 10)], 3, 2500, alo_hip.RESIDENT_ALWAYS) == 0   # 92 016 bytes
    # a host copy whose pixel count is not S, or another level count: nothing resident
    host = (ctypes.c_int32 * 8)(*[v for hw in detr for v in hw])
    assert lib.alo_msda_resident_levels(host, 8, 22222, 8, 4, 22222, alo_hip.RESIDENT_ALWAYS) == 0
    assert lib.alo_msda_resident_levels(host, 8, 22223, 8, 3, 22223, alo_hip.RESIDENT_ALWAYS) == 0
    assert lib.alo_msda_resident_levels(None, 8, 22223, 8, 4, 22223, alo_hip.RESIDENT_ALWAYS) == 0


def test_backward_dispatch_table_without_a_gpu():
    """alo_msda_backward_path is pure host logic: which backward kernel a launch takes (csrc/msda.hip backward_impl)."""
    lib = alo_hip.lib()
    shapes = [(100, 167), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes)
    hint = (ctypes.c_int32 * 8)(*[v for hw in shapes for v in hw])
    f32, f64, bf16 = alo_hip.ALO_F32, alo_hip.ALO_F64, alo_hip.ALO_BF16
    path = lambda D, Lq, vdt, ldt, h, L=4, P=4: lib.alo_msda_backward_path(4, S, 8, D, L, Lq, P, vdt, ldt, h)  # noqa: E731
    assert path(32, S, f32, f32, hint) == 2 and path(32, S, bf16, f32, hint) == 2 and path(64, S, f32, f32, hint) == 2
    assert path(32, S, f32, f32, None) == 1 and path(32, 300, f32, f32, hint) == 1
    assert path(32, S, f64, f64, hint) == 0 and path(128, S, f32, f32, hint) == 0 and path(32, S, f32, f32, hint, P=8) == 0
    assert path(32, S, f32, f64, hint) == -1     # not a supported dtype pair
    wrong = (ctypes.c_int32 * 8)(100, 167, 50, 84, 25, 42, 13, 20)   # does not add up to S: no block table
    assert path(32, S, f32, f32, wrong) == 1
