"""ctypes binding of ``libalo_hotpath.so`` (the C ABI declared in ``include/alo_hotpath.h``).

This is the only place the host code touches the native library.  PyTorch is used for what it is good at here —
device memory, streams, dtypes — and nothing else: every function below takes torch tensors, validates them the way
the reference op validates its ``at::Tensor`` arguments (alonet/deformable_detr/ops/src/cuda/ms_deform_attn_cuda.cu:28-52),
allocates the outputs (so torch owns the memory, as with the reference op) and enqueues the HIP kernels on the
current torch stream.

There is NO fallback: if the library is missing or a tensor lives on the CPU these functions raise ``RuntimeError``.
"""
import ctypes
import warnings
import os
import subprocess

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_ROOT, "libalo_hotpath.so")
CSRC_DIR = os.path.join(_PKG_ROOT, "csrc")

ALO_F32, ALO_F64, ALO_BF16 = 0, 1, 2
RESIDENT_AUTO, RESIDENT_ALWAYS = 0, 1   # ALO_RESIDENT_* of include/alo_hotpath.h
_DTYPE_CODE = {torch.float32: ALO_F32, torch.float64: ALO_F64, torch.bfloat16: ALO_BF16}

_lib = None
_warned_inference_tensor = False


def tensor_version(t):
    """``t._version`` for the ``(version, data_ptr)`` cache keys of this package.  Inference tensors (``torch.inference_mode()``)
    carry no version counter and raise when asked for one (round-3 advisor finding), yet can still be updated in place inside
    inference mode with their ``data_ptr`` unchanged — so for them the key is a fresh object that equals nothing, itself of an
    earlier call included: whatever is derived from an inference tensor is re-derived on every use instead of being cached
    (round-4 advisor finding: a constant there made stale packed weights / host shapes possible)."""
    if t.is_inference():
        global _warned_inference_tensor
        if not _warned_inference_tensor:
            _warned_inference_tensor = True
            warnings.warn("alo_hip: a parameter / shape tensor created under torch.inference_mode() carries no version counter, so what is "
                          "derived from it (packed MFMA weights, folded convolutions, the host copy of spatial_shapes) is re-derived on "
                          "every call instead of being cached — results are right, the forward is slower.  Build / load the model outside "
                          "inference_mode() (torch.no_grad() is enough for inference).", RuntimeWarning, stacklevel=3)
        return object()
    return None if t.is_inference() else t._version


class HotpathUnavailable(RuntimeError):
    """libalo_hotpath.so cannot be loaded (not built, or built for another ABI)."""


def build(force=False):
    """Compile the HIP kernels for gfx950 with hipcc (``make -C aloception-oss_amd/csrc``). Needs no GPU."""
    cmd = ["make", "-C", CSRC_DIR, "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def _declare(lib):
    c = ctypes
    vp, ip, sz = c.c_void_p, c.c_int, c.c_size_t
    lib.alo_abi_version.restype = ip
    lib.alo_abi_version.argtypes = []
    lib.alo_last_error.restype = c.c_char_p
    lib.alo_last_error.argtypes = []
    lib.alo_msda_forward.restype = ip
    lib.alo_msda_forward.argtypes = [vp] * 6 + [ip] * 9 + [vp]
    lib.alo_msda_forward_fused.restype = ip
    lib.alo_msda_forward_fused.argtypes = [vp] * 7 + [ip] * 9 + [vp]
    lib.alo_msda_backward.restype = ip
    lib.alo_msda_backward.argtypes = [vp] * 9 + [ip] * 9 + [vp]
    lib.alo_msda_backward_hinted.restype = ip
    lib.alo_msda_backward_hinted.argtypes = [vp] * 9 + [ip] * 9 + [c.POINTER(c.c_int32), vp]
    lib.alo_msda_backward_path.restype = ip
    lib.alo_msda_backward_path.argtypes = [ip] * 9 + [c.POINTER(c.c_int32)]
    lib.alo_corr_level_shape.restype = None
    lib.alo_corr_level_shape.argtypes = [ip, ip, ip, c.POINTER(ip), c.POINTER(ip)]
    lib.alo_corr_build_workspace_bytes.restype = sz
    lib.alo_corr_build_workspace_bytes.argtypes = [ip] * 5
    lib.alo_corr_build.restype = ip
    lib.alo_corr_build.argtypes = [vp, vp, c.POINTER(vp), vp, sz] + [ip] * 5 + [vp]
    lib.alo_corr_lookup.restype = ip
    lib.alo_corr_lookup.argtypes = [c.POINTER(vp), vp, vp] + [ip] * 5 + [vp]
    lib.alo_corr_lookup_backward.restype = ip
    lib.alo_corr_lookup_backward.argtypes = [c.POINTER(vp), vp, vp] + [ip] * 5 + [vp]
    lib.alo_corr_lookup_backward_coords.restype = ip
    lib.alo_corr_lookup_backward_coords.argtypes = [c.POINTER(vp), vp, vp, vp] + [ip] * 5 + [vp]
    lib.alo_msda_forward_fused_hm.restype = ip
    lib.alo_msda_forward_fused_hm_rows.restype = ip
    lib.alo_msda_forward_fused_hm_rows.argtypes = [vp] * 5 + [c.c_long, c.c_long, vp, vp] + [ip] * 9 + [vp]
    lib.alo_msda_forward_fused_hm.argtypes = [vp] * 7 + [ip] * 9 + [vp]
    lib.alo_msda_forward_fused_hm_resident.restype = ip
    lib.alo_msda_forward_fused_hm_resident.argtypes = [vp] * 5 + [c.c_long, c.c_long, vp, vp] + [ip] * 9 + [c.POINTER(c.c_int32), ip, vp]
    lib.alo_msda_resident_levels.restype = ip
    lib.alo_msda_resident_levels.argtypes = [c.POINTER(c.c_int32)] + [ip] * 6
    lib.alo_value_head_major.restype = ip
    lib.alo_value_head_major.argtypes = [vp] * 3 + [ip] * 5 + [vp]
    lib.alo_bias_act_nchw.restype = ip
    lib.alo_bias_act_nchw.argtypes = [vp] * 3 + [ip] * 4 + [vp]
    lib.alo_gru_gate.restype = ip
    lib.alo_gru_gate.argtypes = [vp] * 4 + [ip] * 3 + [c.c_long, c.c_long, vp]
    lib.alo_gru_update.restype = ip
    lib.alo_gru_update.argtypes = [vp] * 5 + [ip] * 3 + [c.c_long, vp]
    lib.alo_pack_mfma_b.restype = ip
    lib.alo_pack_mfma_b.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.alo_value_proj_head_major.restype = ip
    lib.alo_value_proj_head_major.argtypes = [vp] * 5 + [ip] * 5 + [vp]
    lib.alo_conv3x3_nhwc.restype = ip
    lib.alo_conv3x3_nhwc.argtypes = [vp] * 5 + [ip] * 8 + [vp]
    lib.alo_conv3x3_workspace_bytes.restype = c.c_size_t
    lib.alo_conv3x3_workspace_bytes.argtypes = [ip] * 6
    lib.alo_stem_conv_pool.restype = ip
    lib.alo_stem_conv_pool.argtypes = [vp] * 4 + [ip] * 3 + [c.c_long] * 4 + [ip, vp]
    lib.alo_groupnorm_rows_workspace_bytes.restype = c.c_size_t
    lib.alo_groupnorm_rows_workspace_bytes.argtypes = [ip, ip, ip]
    lib.alo_groupnorm_rows.restype = ip
    lib.alo_groupnorm_rows.argtypes = [vp] * 5 + [ip] * 4 + [c.c_float, c.c_long, ip, vp]
    lib.alo_groupnorm_rows_act.restype = ip
    lib.alo_groupnorm_rows_act.argtypes = [vp] * 5 + [ip] * 4 + [c.c_float, c.c_long, ip, ip, vp]
    lib.alo_conv3x3_small_nhwc.restype = ip
    lib.alo_conv3x3_small_nhwc.argtypes = [vp] * 4 + [ip] * 6 + [vp]
    lib.alo_upsample_add_nhwc.restype = ip
    lib.alo_upsample_add_nhwc.argtypes = [vp] * 3 + [ip] * 8 + [vp]
    lib.alo_linear_packed.restype = ip
    lib.alo_linear_packed.argtypes = [vp] * 5 + [c.c_long, ip, ip, ip, ip, vp]
    lib.alo_mask_pyramid.restype = ip
    lib.alo_mask_pyramid.argtypes = [vp, ip, vp, vp, ip, ip, ip, ip, c.POINTER(c.c_int), c.c_uint, vp]
    lib.alo_encoder_reference_points.restype = ip
    lib.alo_encoder_reference_points.argtypes = [vp, vp, ip, ip, c.POINTER(c.c_int), vp]
    lib.alo_conv1x1_nhwc.restype = ip
    lib.alo_conv1x1_nhwc.argtypes = [vp, vp, ip, vp, vp, vp] + [ip] * 8 + [vp]
    lib.alo_panoptic_onehot.restype = ip
    lib.alo_panoptic_onehot.argtypes = [vp, vp] + [ip] * 6 + [c.c_float, vp]
    lib.alo_ffn256.restype = ip
    lib.alo_ffn256.argtypes = [vp] * 6 + [c.c_long, ip, ip, vp]
    lib.alo_linear_shortk.restype = ip
    lib.alo_linear_shortk.argtypes = [vp] * 5 + [c.c_long, ip, ip, ip, ip, vp]
    lib.alo_pos_sine_flat.restype = ip
    lib.alo_pos_sine_flat.argtypes = [vp] * 7 + [ip] * 6 + [c.c_float, c.c_float, ip, vp]
    lib.alo_add_layernorm.restype = ip
    lib.alo_add_layernorm.argtypes = [vp] * 7 + [c.c_long, ip, c.c_float, ip, vp]
    lib.alo_bias_act.restype = ip
    lib.alo_bias_act.argtypes = [vp] * 4 + [c.c_long, ip, ip, ip, vp]


def lib():
    """The loaded library; raises :class:`HotpathUnavailable` (never falls back) when it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HotpathUnavailable(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {CSRC_DIR}` (hipcc, --offload-arch=gfx950). There is no CPU fallback for this path."
            )
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the box
            raise HotpathUnavailable(f"cannot load {LIB_PATH}: {e}") from e
        _declare(handle)
        if handle.alo_abi_version() != 2:
            raise HotpathUnavailable(f"{LIB_PATH} has ABI version {handle.alo_abi_version()}, expected 2")
        _lib = handle
    return _lib


def is_available():
    try:
        lib()
        return True
    except HotpathUnavailable:
        return False


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().alo_last_error().decode() or f"alo_hotpath error {rc}")


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _require_cuda_contiguous(named):
    # same order and wording as the reference's AT_ASSERTM list (ms_deform_attn_cuda.cu:28-38,91-105)
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
    for name, t in named:
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")


class LaunchTimer:
    """Times every kernel launch made through this module with HIP events recorded on the launch stream.

    ``with alo_hip.LaunchTimer() as t: ...``; afterwards ``t.summary()`` maps a kernel tag (e.g. ``"msda_fwd/Lq=22223"``)
    to ``dict(calls, ms_total, ms_avg, alg_bytes_avg, alg_flops_avg)``.  Used by bench.py for the roofline numbers:
    the algorithmic byte / flop counts are the SURVEY.md section 8(d) formulas evaluated on the actual launch shape.
    """

    def __init__(self, only=None):
        self.records = []
        self.relaunch = {}  # tag -> closure that enqueues the tag's most recent launch again (same buffers)
        self.only = only    # tag prefix: time just these launches (an event pair per launch is not free on the GPU either)

    def __enter__(self):
        global _timer
        self._prev, _timer = _timer, self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = self._prev

    def replay_ms(self, tag, reps=20):
        """Average duration of ``reps`` back-to-back re-launches of the last launch recorded under ``tag`` (same device
        buffers).  An event pair around ONE launch also measures the dispatch gap either side of it (tens of microseconds
        on ROCm); a back-to-back train does not, and agrees with rocprofv3's per-kernel average."""
        fn = self.relaunch[tag]
        fn()
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            fn()
        stop.record()
        torch.cuda.synchronize()
        return start.elapsed_time(stop) / reps

    def replay_samples(self, tag, reps=7):
        """Durations (ms) of ``reps`` re-launches of the last launch recorded under ``tag``, each one timed by its own event
        pair with the next already queued behind it (so a sample is the kernel sequence itself, not the dispatch gap): for
        launches that happen once per step (the correlation build) a median and a minimum instead of a single reading."""
        fn = self.relaunch[tag]
        fn()
        torch.cuda.synchronize()
        events = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        events[0].record()
        for i in range(reps):
            fn()
            events[i + 1].record()
        torch.cuda.synchronize()
        return [events[i].elapsed_time(events[i + 1]) for i in range(reps)]

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, start, stop, nbytes, flops in self.records:
            d = out.setdefault(tag, dict(calls=0, ms_total=0.0, alg_bytes_avg=0.0, alg_flops_avg=0.0))
            d["calls"] += 1
            d["ms_total"] += start.elapsed_time(stop)
            d["alg_bytes_avg"] += nbytes
            d["alg_flops_avg"] += flops
        for d in out.values():
            d["ms_avg"] = d["ms_total"] / d["calls"]
            d["alg_bytes_avg"] /= d["calls"]
            d["alg_flops_avg"] /= d["calls"]
        return out


_timer = None


class _timed:
    def __init__(self, tag, nbytes=0.0, flops=0.0, relaunch=None):
        self.tag, self.nbytes, self.flops, self.relaunch = tag, nbytes, flops, relaunch

    def __enter__(self):
        self.on = _timer is not None and (_timer.only is None or self.tag.startswith(_timer.only))
        if self.on:
            self.start = torch.cuda.Event(enable_timing=True)
            self.stop = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.stop.record()
            _timer.records.append((self.tag, self.start, self.stop, self.nbytes, self.flops))
            if self.relaunch is not None:
                _timer.relaunch[self.tag] = self.relaunch


def msda_forward_bytes(N, S, M, D, L, Lq, P, elem, loc_elem=4):
    """Algorithmic HBM bytes of one forward launch: value once + out once + (loc, attn) once."""
    return elem * (N * S * M * D + N * Lq * M * D) + loc_elem * (N * Lq * M * L * P * 3)


def msda_backward_bytes(N, S, M, D, L, Lq, P, elem, loc_elem=4):
    return elem * (2 * N * S * M * D + N * Lq * M * D) + loc_elem * (N * Lq * M * L * P * 3 * 2)


def _msda_prepare(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, extra=()):
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")  # ms_deform_attn.h:38,60
    _require_cuda_contiguous(
        [("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
         ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)] + list(extra)
    )
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("value must be (N,S,M,D), sampling_loc (N,Lq,M,L,P,2), attn_weight (N,Lq,M,L,P)")
    if spatial_shapes.dtype != torch.int32 or level_start_index.dtype != torch.int32:
        # this fork of the op reads int32 metadata (ms_deform_attn_cuda.cu:67-68)
        raise RuntimeError("spatial_shapes and level_start_index must be int32 tensors")
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = sampling_loc.shape
    if (M2, two) != (M, 2) or tuple(attn_weight.shape) != (N, Lq, M, L, P) or sampling_loc.shape[0] != N:
        raise RuntimeError("sampling_loc / attn_weight shapes do not match value")
    if tuple(spatial_shapes.shape) != (L, 2) or tuple(level_start_index.shape) != (L,):
        raise RuntimeError("spatial_shapes must be (L,2) and level_start_index (L,)")
    step = min(N, int(im2col_step))
    if step <= 0 or N % step != 0:
        raise RuntimeError(f"batch({N}) must divide im2col_step({step})")
    vdt = _DTYPE_CODE.get(value.dtype)
    if vdt is None:
        raise RuntimeError(f"ms_deform_attn: unsupported value dtype {value.dtype}")
    if value.dtype == torch.bfloat16:
        # bf16 storage: sampling geometry stays fp32 (bf16 locations would cost ~0.3 px at 167-wide maps)
        sampling_loc, attn_weight = sampling_loc.float(), attn_weight.float()
    elif sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("sampling_loc and attn_weight must have the dtype of value")
    ldt = _DTYPE_CODE[sampling_loc.dtype]
    return (N, S, M, D, L, Lq, P), vdt, ldt, sampling_loc.contiguous(), attn_weight.contiguous()


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """-> (N, Lq, M*D) tensor of ``value``'s dtype.  Replaces ``alonet_custom::ms_deform_attn_forward``."""
    dims, vdt, ldt, loc, attn = _msda_prepare(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    N, S, M, D, L, Lq, P = dims
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    nbytes = msda_forward_bytes(N, S, M, D, L, Lq, P, value.element_size(), loc.element_size())
    with torch.cuda.device(value.device), _timed(f"msda_fwd/Lq={Lq}", nbytes):
        _check(lib().alo_msda_forward(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(loc), _ptr(attn),
                                      _ptr(out), N, S, M, D, L, Lq, P, vdt, ldt, _stream(value.device)))
    return out


def msda_forward_fused(value, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points):
    """MSDeformAttn's prologue + gather in one launch (inference): raw offsets (N,Lq,M,L,P,2) and raw attention logits
    (N,Lq,M,L*P) in ``value``'s dtype, reference points (N,Lq,L,2|4) in fp32 (fp64 for fp64 values) -> (N,Lq,M*D)."""
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_offsets.shape
    geo = torch.float64 if value.dtype == torch.float64 else torch.float32
    reference_points = reference_points.to(geo).contiguous()
    _require_cuda_contiguous([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                              ("sampling_offsets", sampling_offsets), ("attn_logits", attn_logits),
                              ("reference_points", reference_points)])
    if sampling_offsets.dtype != value.dtype or attn_logits.dtype != value.dtype:
        raise RuntimeError("sampling_offsets and attn_logits must have the dtype of value")
    if spatial_shapes.dtype != torch.int32 or level_start_index.dtype != torch.int32:
        raise RuntimeError("spatial_shapes and level_start_index must be int32 tensors")
    ref_dim = reference_points.shape[-1]
    if tuple(reference_points.shape) != (N, Lq, L, ref_dim) or attn_logits.numel() != N * Lq * M * L * P:
        raise RuntimeError("reference_points must be (N,Lq,L,2|4) and attn_logits (N,Lq,M,L*P)")
    vdt = _DTYPE_CODE.get(value.dtype)
    if vdt is None:
        raise RuntimeError(f"ms_deform_attn: unsupported value dtype {value.dtype}")
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    # bytes actually streamed by the fused launch: value + out + raw offsets/logits (value dtype) + reference points
    e = value.element_size()
    nbytes = e * (N * S * M * D + N * Lq * M * D + N * Lq * M * L * P * 3) + reference_points.element_size() * reference_points.numel()
    with torch.cuda.device(value.device), _timed(f"msda_fwd_fused/Lq={Lq}", nbytes):
        _check(lib().alo_msda_forward_fused(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index),
                                            _ptr(sampling_offsets), _ptr(attn_logits), _ptr(reference_points), _ptr(out),
                                            N, S, M, D, L, Lq, P, ref_dim, vdt, _stream(value.device)))
    return out


def head_major_supported(value, L, P):
    """The head-major fast path of the fused forward exists for the DETR-family shape only: bf16, L = P = 4, D in {8..32}."""
    return (value.is_cuda and value.dtype == torch.bfloat16 and L == 4 and P == 4 and value.shape[-1] % 8 == 0
            and value.shape[-1] <= 32)


def value_head_major(value, padding_mask=None):
    """(N, S, M, D) bf16 -> (N, M, S, D) with the rows of padded pixels zeroed: ``value.masked_fill(mask[..., None], 0)``
    and the re-layout for ``msda_forward_fused_hm`` in one pass."""
    _require_cuda_contiguous([("value", value)])
    N, S, M, D = value.shape
    if padding_mask is not None:
        if padding_mask.dtype != torch.bool or tuple(padding_mask.shape) != (N, S) or not padding_mask.is_cuda:
            raise RuntimeError("padding_mask must be a (N, S) bool CUDA tensor")
        padding_mask = padding_mask.contiguous()
    out = torch.empty((N, M, S, D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device), _timed(f"value_head_major/S={S}", 2 * value.element_size() * value.numel()):
        _check(lib().alo_value_head_major(_ptr(value), None if padding_mask is None else _ptr(padding_mask), _ptr(out),
                                          N, S, M, D, _DTYPE_CODE[value.dtype], _stream(value.device)))
    return out


def _query_rows(t, inner):
    """Row stride (elements) of a (N, Lq, ...) tensor whose per-query block of ``inner`` elements is dense and whose queries are
    evenly spaced — a dense tensor or a column slice of a wider (N, Lq, C) buffer; None otherwise."""
    if t.dim() < 3 or t.shape[0] == 0 or t.shape[1] == 0:
        return None
    expect = 1
    for size, stride in zip(reversed(t.shape[2:]), reversed(t.stride()[2:])):
        if size != 1 and stride != expect:
            return None
        expect *= size
    if expect != inner:
        return None
    rs = t.stride(1) if t.shape[1] > 1 else inner
    if rs < inner or (t.shape[0] > 1 and t.stride(0) != rs * t.shape[1]):
        return None
    return rs


def msda_forward_fused_hm(value_hm, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points,
                          resident=True):
    """``msda_forward_fused`` on a head-major value (N, M, S, D) (see ``value_head_major``).
    ``sampling_offsets`` (N, Lq, M, L, P, 2) and ``attn_logits`` (N, Lq, M, L*P) may be column slices of one wider (N, Lq, C)
    buffer (a merged projection): only their per-query blocks have to be dense.
    ``resident`` (default): when a host copy of ``spatial_shapes`` rides on the tensor (``_alo_shapes``, set by
    DeformableTransformer) and D = 32, large launches keep the coarse pyramid levels in LDS
    (``alo_msda_forward_fused_hm_resident``: same products, fp32 accumulation order of the levels unchanged; the library takes
    the plain head-major kernel by itself where that one is faster — launches with less than one 16-query run per wave of the
    chip).  ``resident="always"`` takes the resident kernel wherever it can run (ALO_RESIDENT_ALWAYS), ``resident=False`` always
    runs the plain kernel, whose output is bit-identical to ``msda_forward_fused``."""
    if not value_hm.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    N, M, S, D = value_hm.shape
    _, Lq, _, L, P, _ = sampling_offsets.shape
    reference_points = reference_points.float().contiguous()
    _require_cuda_contiguous([("value", value_hm), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index)])
    if sampling_offsets.dtype != value_hm.dtype or attn_logits.dtype != value_hm.dtype:
        raise RuntimeError("sampling_offsets and attn_logits must have the dtype of value")
    if spatial_shapes.dtype != torch.int32 or level_start_index.dtype != torch.int32:
        raise RuntimeError("spatial_shapes and level_start_index must be int32 tensors")
    ref_dim = reference_points.shape[-1]
    if tuple(reference_points.shape) != (N, Lq, L, ref_dim) or attn_logits.numel() != N * Lq * M * L * P:
        raise RuntimeError("reference_points must be (N,Lq,L,2|4) and attn_logits (N,Lq,M,L*P)")
    off_rs, log_rs = _query_rows(sampling_offsets, M * L * P * 2), _query_rows(attn_logits, M * L * P)
    if off_rs is None or log_rs is None or off_rs % 8 or log_rs % 8 or not sampling_offsets.is_cuda or not attn_logits.is_cuda:
        sampling_offsets, attn_logits = sampling_offsets.contiguous(), attn_logits.contiguous()
        off_rs, log_rs = M * L * P * 2, M * L * P
    out = torch.empty((N, Lq, M * D), dtype=value_hm.dtype, device=value_hm.device)
    e = value_hm.element_size()
    nbytes = e * (N * S * M * D + N * Lq * M * D + N * Lq * M * L * P * 3) + 4 * reference_points.numel()
    # coarse levels resident in LDS: needs a HOST copy of the shapes (it picks the resident levels, fixes the LDS layout and sizes the
    # grid; the kernel re-checks it against the device copy).  Only a copy that is already at hand is used — no device read in a forward.
    host = getattr(spatial_shapes, "_alo_shapes", None) if resident else None
    if host is None and resident:
        hit = getattr(spatial_shapes, "_alo_shapes_read", None)
        host = hit[1] if hit is not None and hit[0] == tensor_version(spatial_shapes) else None
    starts = None
    if host is not None and D == 32 and len(host) == L and sum(int(h) * int(w) for h, w in host) == S:
        starts = (ctypes.c_int32 * (2 * L))(*[int(v) for hw in host for v in hw])

    policy = RESIDENT_ALWAYS if resident == "always" else RESIDENT_AUTO

    def launch():
        if starts is not None:
            _check(lib().alo_msda_forward_fused_hm_resident(_ptr(value_hm), _ptr(spatial_shapes), _ptr(level_start_index),
                                                            _ptr(sampling_offsets), _ptr(attn_logits), off_rs, log_rs,
                                                            _ptr(reference_points), _ptr(out), N, S, M, D, L, Lq, P, ref_dim,
                                                            _DTYPE_CODE[value_hm.dtype], starts, policy, _stream(value_hm.device)))
            return
        _check(lib().alo_msda_forward_fused_hm_rows(_ptr(value_hm), _ptr(spatial_shapes), _ptr(level_start_index),
                                                    _ptr(sampling_offsets), _ptr(attn_logits), off_rs, log_rs,
                                                    _ptr(reference_points), _ptr(out), N, S, M, D, L, Lq, P, ref_dim,
                                                    _DTYPE_CODE[value_hm.dtype], _stream(value_hm.device)))

    tag = "msda_fwd_fused_resident" if starts is not None and lib().alo_msda_resident_levels(starts, N, S, M, L, Lq, policy) else "msda_fwd_fused"
    with torch.cuda.device(value_hm.device), _timed(f"{tag}/Lq={Lq}", nbytes, relaunch=launch if _timer else None):
        launch()
    return out


def _host_spatial_shapes(spatial_shapes):
    """Host copy of a device ``spatial_shapes`` tensor.  It rides on the tensor OBJECT (``_alo_shapes``, tagged with the version
    counter it was read at; DeformableTransformer attaches its own list when it builds the tensor), so it dies with the tensor: a cache
    keyed on the storage pointer would hand a stale copy to the next 32-byte tensor the caching allocator places at that address.
    Without the attribute: one device-to-host read, then cached on the object."""
    host = getattr(spatial_shapes, "_alo_shapes", None)   # attached by the model that built the tensor
    if host is not None:
        return host
    hit = getattr(spatial_shapes, "_alo_shapes_read", None)
    if hit is not None and hit[0] == tensor_version(spatial_shapes):
        return hit[1]
    host = [tuple(int(v) for v in hw) for hw in spatial_shapes.tolist()]
    spatial_shapes._alo_shapes_read = (tensor_version(spatial_shapes), host)
    return host


def _tiled_backward_eligible(value, dims, ldt):
    """The launches msda_bwd_wide_kernel covers (csrc/msda_bwd_wide.hip): fp32 / bf16 values, D = 32 or 64, L = P = 4, queries = the
    pyramid's own pixels.  Only then is the host copy of the shapes worth a device-to-host read."""
    N, S, M, D, L, Lq, P = dims
    return value.dtype in (torch.float32, torch.bfloat16) and ldt == ALO_F32 and D in (32, 64) and L == 4 and P == 4 and Lq == S


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step=64):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight].  Replaces ``alonet_custom::ms_deform_attn_backward``."""
    dims, vdt, ldt, loc, attn = _msda_prepare(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                              im2col_step, extra=[("grad_output", grad_output)])
    N, S, M, D, L, Lq, P = dims
    if grad_output.dtype != value.dtype or grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output must be (N, Lq, M*D) with the dtype of value")
    gdt = torch.float64 if value.dtype == torch.float64 else torch.float32
    grad_value = torch.empty(value.shape, dtype=gdt, device=value.device)
    grad_loc = torch.empty(loc.shape, dtype=gdt, device=value.device)
    grad_attn = torch.empty(attn.shape, dtype=gdt, device=value.device)
    nbytes = msda_backward_bytes(N, S, M, D, L, Lq, P, value.element_size(), loc.element_size())
    # only the encoder's self-attention on the DETR-family shape can use the hint (it sizes the grid of 4x4 query tiles)
    host = _host_spatial_shapes(spatial_shapes) if _tiled_backward_eligible(value, dims, ldt) else None
    hint = None if host is None else (ctypes.c_int32 * (2 * L))(*[int(v) for hw in host for v in hw])

    def launch():
        _check(lib().alo_msda_backward_hinted(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(loc), _ptr(attn),
                                              _ptr(grad_output), _ptr(grad_value), _ptr(grad_loc), _ptr(grad_attn),
                                              N, S, M, D, L, Lq, P, vdt, ldt, hint, _stream(value.device)))

    with torch.cuda.device(value.device), _timed(f"msda_bwd/Lq={Lq}", nbytes, relaunch=launch if _timer else None):
        launch()
    return [grad_value.to(value.dtype), grad_loc.to(sampling_loc.dtype), grad_attn.to(attn_weight.dtype)]


def corr_level_shapes(H, W, num_levels):
    out = []
    h, w = ctypes.c_int(), ctypes.c_int()
    for lvl in range(num_levels):
        lib().alo_corr_level_shape(H, W, lvl, ctypes.byref(h), ctypes.byref(w))
        out.append((h.value, w.value))
    return out


def _require_f32_cuda(name, t, ndim):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (the correlation path has no CPU implementation here)")
    if t.dtype != torch.float32 or t.dim() != ndim:
        raise RuntimeError(f"{name} must be a {ndim}-d float32 tensor, got {tuple(t.shape)} {t.dtype}")


def corr_build(fmap1, fmap2, num_levels=4):
    """fmaps (B,C,H,W) float32 -> list of ``num_levels`` tensors (B*H*W, 1, h_l, w_l)  (corr.py:13-27)."""
    _require_f32_cuda("fmap1", fmap1, 4)
    _require_f32_cuda("fmap2", fmap2, 4)
    if fmap1.shape != fmap2.shape:
        raise RuntimeError("fmap1 and fmap2 must have the same shape")
    fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
    B, C, H, W = fmap1.shape
    shapes = corr_level_shapes(H, W, num_levels)
    levels = [torch.empty((B * H * W, 1, h, w), dtype=torch.float32, device=fmap1.device) for h, w in shapes]
    nbytes = lib().alo_corr_build_workspace_bytes(B, C, H, W, num_levels)
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=torch.float32, device=fmap1.device)
    ptrs = (ctypes.c_void_p * num_levels)(*[t.data_ptr() for t in levels])
    ncols = sum(h * w for h, w in shapes)
    def launch():   # keeps fmaps, levels and workspace alive for LaunchTimer.replay_samples
        _check(lib().alo_corr_build(_ptr(fmap1), _ptr(fmap2), ptrs, _ptr(ws), nbytes, B, C, H, W, num_levels,
                                    _stream(fmap1.device)))

    with torch.cuda.device(fmap1.device), _timed("corr_build", 4.0 * (2 * B * C * H * W + B * H * W * ncols),
                                                 2.0 * B * (H * W) * (H * W) * C, relaunch=launch if _timer else None):
        launch()
    # ws may be released now: the caching allocator keeps the block bound to this stream until the kernels retire
    return levels


def corr_lookup(levels, coords, radius=4):
    """levels from :func:`corr_build`, coords (B,2,H,W) -> (B, L*(2r+1)^2, H, W) float32  (corr.py:29-50)."""
    _require_f32_cuda("coords", coords, 4)
    coords = coords.contiguous()
    B, two, H, W = coords.shape
    if two != 2:
        raise RuntimeError("coords must be (B,2,H,W)")
    L = len(levels)
    for lvl, t in enumerate(levels):
        _require_f32_cuda(f"corr_pyramid[{lvl}]", t, 4)
        if not t.is_contiguous() or t.shape[0] != B * H * W:
            raise RuntimeError(f"corr_pyramid[{lvl}] must be a contiguous (B*H*W,1,h,w) tensor")
    out = torch.empty((B, L * (2 * radius + 1) ** 2, H, W), dtype=torch.float32, device=coords.device)
    ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in levels])
    taps = (2 * radius + 2) ** 2
    nbytes = 4.0 * B * H * W * (L * (2 * radius + 1) ** 2 + L * taps + 2)
    with torch.cuda.device(coords.device), _timed("corr_lookup", nbytes):
        _check(lib().alo_corr_lookup(ptrs, _ptr(coords), _ptr(out), B, H, W, radius, L, _stream(coords.device)))
    return out


def corr_lookup_backward(grad_levels, coords, grad_out, radius=4):
    """Adds the pyramid gradients of ONE lookup to ``grad_levels`` (tensors shaped like the pyramid, zeroed by the caller before the
    first lookup whose gradients are to be summed): the adjoint of :func:`corr_lookup` with respect to the levels (autograd through
    the reference's bilinear_sampler, corr.py:29-50).  In place; returns ``grad_levels``."""
    _require_f32_cuda("coords", coords, 4)
    _require_f32_cuda("grad_out", grad_out, 4)
    coords, grad_out = coords.contiguous(), grad_out.contiguous()
    B, two, H, W = coords.shape
    L = len(grad_levels)
    if two != 2 or tuple(grad_out.shape) != (B, L * (2 * radius + 1) ** 2, H, W):
        raise RuntimeError("corr_lookup_backward: coords must be (B,2,H,W) and grad_out (B, L*(2r+1)^2, H, W)")
    for lvl, t in enumerate(grad_levels):
        _require_f32_cuda(f"grad_levels[{lvl}]", t, 4)
        if not t.is_contiguous() or t.shape[0] != B * H * W:
            raise RuntimeError(f"grad_levels[{lvl}] must be a contiguous (B*H*W,1,h,w) tensor")
    ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in grad_levels])
    taps = (2 * radius + 2) ** 2
    nbytes = 4.0 * B * H * W * (L * (2 * radius + 1) ** 2 + 2 * L * taps + 2)
    with torch.cuda.device(coords.device), _timed("corr_lookup_backward", nbytes):
        _check(lib().alo_corr_lookup_backward(ptrs, _ptr(coords), _ptr(grad_out), B, H, W, radius, L, _stream(coords.device)))
    return grad_levels


def corr_lookup_backward_coords(levels, coords, grad_out, radius=4):
    """Gradient of :func:`corr_lookup` with respect to ``coords`` -> (B, 2, H, W): grid_sample's gradient with respect to the grid
    chained through the reference's coordinate arithmetic (corr.py:29-50); per-level maps from the kernel, added here."""
    _require_f32_cuda("coords", coords, 4)
    _require_f32_cuda("grad_out", grad_out, 4)
    coords, grad_out = coords.contiguous(), grad_out.contiguous()
    B, two, H, W = coords.shape
    L = len(levels)
    if two != 2 or tuple(grad_out.shape) != (B, L * (2 * radius + 1) ** 2, H, W):
        raise RuntimeError("corr_lookup_backward_coords: coords must be (B,2,H,W) and grad_out (B, L*(2r+1)^2, H, W)")
    for lvl, t in enumerate(levels):
        _require_f32_cuda(f"corr_pyramid[{lvl}]", t, 4)
        if not t.is_contiguous() or t.shape[0] != B * H * W:
            raise RuntimeError(f"corr_pyramid[{lvl}] must be a contiguous (B*H*W,1,h,w) tensor")
    per_level = torch.empty((B, L, 2, H, W), dtype=torch.float32, device=coords.device)
    ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in levels])
    taps = (2 * radius + 2) ** 2
    nbytes = 4.0 * B * H * W * (L * (2 * radius + 1) ** 2 + L * taps + 2 + 2 * L)
    with torch.cuda.device(coords.device), _timed("corr_lookup_backward_coords", nbytes):
        _check(lib().alo_corr_lookup_backward_coords(ptrs, _ptr(coords), _ptr(grad_out), _ptr(per_level), B, H, W, radius, L,
                                                     _stream(coords.device)))
    return per_level.sum(1)


# ---- one-pass epilogues around the attention op (alo_add_layernorm / alo_bias_act) ---------------------------------------
def fusable(*tensors):
    """True when the fused epilogues may replace the stock ops: inference (no autograd graph), CUDA, fp32 or bf16."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return False
    first = tensors[0]
    return first.is_cuda and first.dtype in (torch.float32, torch.bfloat16)


def add_layernorm_supported(x):
    """What ``alo_add_layernorm`` can take: rows of C % 4 == 0, C <= 1024 elements, 16-byte aligned (callers fall back to
    ``norm(x + y)`` otherwise, e.g. for a non-default d_model)."""
    C = x.shape[-1]
    return C % 4 == 0 and 0 < C <= 1024 and (C * x.element_size()) % 16 == 0 and x.data_ptr() % 16 == 0


def invalidate_caches(module):
    """Drop every derived inference-time tensor this library cached on ``module``'s parameters and sub-modules (packed MFMA
    weights, folded batch-norm convolutions, merged projections).  The caches are keyed on ``(tensor._version, data_ptr)``;
    in-place writes through ``.data`` (``p.data.copy_``, EMA updates, ``nn.init.*_(w.data)``) do not bump the version
    counter, so call this after such weight surgery — ``alonet.common.load_weights`` and ``GraphedForward`` do."""
    module.__dict__["_cache_epoch_alo"] = module.__dict__.get("_cache_epoch_alo", 0) + 1   # GraphedForward re-captures on a new epoch
    for p in list(module.parameters()) + list(module.buffers()):
        for key in ("_alo_packed", "_alo_2d"):
            if key in getattr(p, "__dict__", {}):
                delattr(p, key)
    for m in module.modules():
        for key in [k for k in m.__dict__ if k.startswith("_alo_") or k in ("_folded", "_mask_quarter") or k.startswith("_zr")]:
            del m.__dict__[key]


def cache_epoch(module):
    """How many times :func:`invalidate_caches` ran on ``module`` — graphs captured under an older epoch read freed tensors."""
    return module.__dict__.get("_cache_epoch_alo", 0)


def add_layernorm(x, residual, weight, bias, eps=1e-5, pos=None):
    """``LayerNorm(x + residual)`` over the last dim in one pass; with ``pos`` also returns ``out + pos`` (the next
    layer's ``with_pos_embed``).  Replaces ``norm(src + dropout(src2))`` of the (de)formable transformer layers at
    inference.  -> out  |  (out, out_plus_pos)"""
    C = x.shape[-1]
    rows = x.numel() // C
    x = x.contiguous()
    residual = None if residual is None else residual.contiguous()
    pos = None if pos is None else pos.contiguous()
    for name, t in (("residual", residual), ("pos", pos)):
        if t is not None and (t.shape != x.shape or t.dtype != x.dtype):
            raise RuntimeError(f"add_layernorm: {name} must have the shape and dtype of x")
    weight, bias = weight.to(x.dtype).contiguous(), bias.to(x.dtype).contiguous()
    out = torch.empty_like(x)
    out_pos = None if pos is None else torch.empty_like(x)
    nbytes = x.element_size() * x.numel() * (2 + (residual is not None) + 2 * (pos is not None))
    with torch.cuda.device(x.device), _timed(f"add_layernorm/rows={rows}", nbytes):
        _check(lib().alo_add_layernorm(_ptr(x), None if residual is None else _ptr(residual), _ptr(weight), _ptr(bias),
                                       _ptr(out), None if pos is None else _ptr(pos),
                                       None if pos is None else _ptr(out_pos), rows, C, float(eps),
                                       _DTYPE_CODE[x.dtype], _stream(x.device)))
    return out if pos is None else (out, out_pos)


def bias_act_(x, bias, residual=None, relu=True):
    """In place on a channels-last activation ``x`` (N,C,H,W with NHWC strides) or a (rows, C) matrix:
    ``x = act(x + bias[c] (+ residual))``.  Replaces folded FrozenBatchNorm bias -> (+ identity) -> ReLU of the ResNet."""
    if x.dim() == 4:
        if not x.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError("bias_act_: 4-D input must be channels_last")
        C = x.shape[1]
        if residual is not None and not (residual.shape == x.shape and residual.is_contiguous(memory_format=torch.channels_last)):
            raise RuntimeError("bias_act_: residual must be channels_last with the shape of x")
    else:
        if not x.is_contiguous():
            raise RuntimeError("bias_act_: matrix input must be contiguous")
        C = x.shape[-1]
        if residual is not None and not (residual.shape == x.shape and residual.is_contiguous()):
            raise RuntimeError("bias_act_: residual must be contiguous with the shape of x")
    if residual is not None and residual.dtype != x.dtype:
        raise RuntimeError("bias_act_: residual must have the dtype of x")
    bias = bias.to(x.dtype).contiguous()
    nbytes = x.element_size() * x.numel() * (2 + (residual is not None))
    with torch.cuda.device(x.device), _timed(f"bias_act/C={C}", nbytes):
        _check(lib().alo_bias_act(_ptr(x), _ptr(bias), None if residual is None else _ptr(residual), _ptr(x),
                                  x.numel() // C, C, 1 if relu else 0, _DTYPE_CODE[x.dtype], _stream(x.device)))
    return x


# ---- RAFT update block glue (fp32, NCHW) ------------------------------------------------------------------------------------
def _require_f32_nchw(name, t):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.is_contiguous()):
        raise RuntimeError(f"{name} must be a contiguous float32 CUDA tensor (B, C, H, W)")


def bias_act_nchw_(x, bias, relu=True):
    """In place: ``x = act(x + bias[None, :, None, None])`` — the bias MIOpen would add in a second kernel plus the ReLU."""
    _require_f32_nchw("x", x)
    B, C, H, W = x.shape
    with torch.cuda.device(x.device), _timed(f"bias_act_nchw/C={C}", 8.0 * x.numel()):
        _check(lib().alo_bias_act_nchw(_ptr(x), _ptr(bias.float().contiguous()), _ptr(x), B, C, H * W, 1 if relu else 0,
                                       _stream(x.device)))
    return x


def gru_gate_(zr, bias_zr, hx, rhx, C):
    """``zr`` (B,2C,H,W): pre-activations [z | r].  z <- sigmoid(z + b) in place; ``rhx[:, :C] <- sigmoid(r + b) * hx[:, :C]``."""
    _require_f32_nchw("zr", zr); _require_f32_nchw("hx", hx); _require_f32_nchw("rhx", rhx)
    B, C2, H, W = zr.shape
    if C2 != 2 * C or hx.shape != rhx.shape or hx.shape[0] != B or hx.shape[2:] != zr.shape[2:] or hx.shape[1] < C:
        raise RuntimeError("gru_gate_: zr must be (B,2C,H,W) and hx / rhx (B,C+Cx,H,W)")
    with torch.cuda.device(zr.device), _timed(f"gru_gate/C={C}", 4.0 * B * C * H * W * 5):
        _check(lib().alo_gru_gate(_ptr(zr), _ptr(bias_zr), _ptr(hx), _ptr(rhx), B, C, H * W, hx.stride(0), rhx.stride(0),
                                  _stream(zr.device)))


def gru_update_(q, bias_q, zr, hx, C, net=None):
    """``hx[:, :C] <- (1 - z) * hx[:, :C] + z * tanh(q + b)`` with z = ``zr[:, :C]``; ``net`` (B,C,H,W) also receives the result."""
    _require_f32_nchw("q", q); _require_f32_nchw("zr", zr); _require_f32_nchw("hx", hx)
    B, Cq, H, W = q.shape
    if Cq != C or zr.shape != (B, 2 * C, H, W) or hx.shape[0] != B or hx.shape[2:] != q.shape[2:]:
        raise RuntimeError("gru_update_: q must be (B,C,H,W), zr (B,2C,H,W), hx (B,C+Cx,H,W)")
    if net is not None:
        _require_f32_nchw("net", net)
    with torch.cuda.device(q.device), _timed(f"gru_update/C={C}", 4.0 * B * C * H * W * (4 + (net is not None))):
        _check(lib().alo_gru_update(_ptr(q), _ptr(bias_q), _ptr(zr), _ptr(hx), None if net is None else _ptr(net), B, C,
                                    H * W, hx.stride(0), _stream(q.device)))


def pos_sine_flat(mask_flatten, spatial_shapes, level_start_index, dim_t, level_embed, normalize, center, scale, dtype,
                  eps=1e-6):
    """Sine positional encoding of the flattened pyramid + level embedding -> (B, S, 2F) in ``dtype`` (two launches).
    ``mask_flatten`` (B, S) bool, ``dim_t`` (F,) fp32, ``level_embed`` (L, 2F) or None."""
    B, S = mask_flatten.shape
    L, F = spatial_shapes.shape[0], dim_t.numel()
    if mask_flatten.dtype != torch.bool or dim_t.dtype != torch.float32:
        raise RuntimeError("pos_sine_flat: mask must be bool and dim_t float32")
    mask_flatten, dim_t = mask_flatten.contiguous(), dim_t.contiguous()
    if level_embed is not None:
        level_embed = level_embed.to(dtype).contiguous()
        if tuple(level_embed.shape) != (L, 2 * F):
            raise RuntimeError("pos_sine_flat: level_embed must be (L, 2 * num_pos_feats)")
    out = torch.empty((B, S, 2 * F), dtype=dtype, device=mask_flatten.device)
    work = torch.empty((B, S, 2), dtype=torch.float32, device=mask_flatten.device)
    with torch.cuda.device(out.device), _timed(f"pos_sine_flat/S={S}", out.element_size() * out.numel()):
        _check(lib().alo_pos_sine_flat(_ptr(mask_flatten), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(dim_t),
                                       None if level_embed is None else _ptr(level_embed), _ptr(out), _ptr(work), B, S, L, F,
                                       1 if normalize else 0, 1 if center else 0, float(scale), float(eps),
                                       _DTYPE_CODE[dtype], _stream(out.device)))
    return out


# ---- short-K linear layers on the streaming MFMA kernel (alo_linear_shortk) -----------------------------------------------------
def linear_shortk_supported(x, weight):
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.shape[-1] in (64, 128, 256)
            and weight.dim() == 2 and weight.shape[1] == x.shape[-1] and weight.shape[0] % 64 == 0)


def linear_shortk(x, weight, bias=None, relu=False, residual=None):
    """``act(x @ weight.T + bias [+ residual])`` over the last dim (64 / 128 / 256) of a bf16 ``x``; weight (N, K), N % 64 == 0;
    ``residual`` has the shape of the result and is added before the activation."""
    if not linear_shortk_supported(x, weight):
        raise RuntimeError("linear_shortk: needs bf16 CUDA tensors, K in (64, 128, 256) and N % 64 == 0")
    N, K = weight.shape
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = torch.empty((x2.shape[0], N), dtype=x.dtype, device=x.device)
    M = x2.shape[0]
    if residual is not None:
        residual = residual.reshape(-1, N)
        if residual.shape[0] != M or residual.dtype != x.dtype or not residual.is_contiguous():
            raise RuntimeError("linear_shortk: residual must be a contiguous (M, N) tensor of x's dtype")
    if M:
        nbytes = 2.0 * (x2.numel() + y.numel() * (2 if residual is not None else 1))
        with torch.cuda.device(x.device), _timed(f"linear_shortk/N={N},K={K}", nbytes, 2.0 * M * N * K):
            _check(lib().alo_linear_shortk(_ptr(x2), _ptr(weight.contiguous()), None if bias is None else _ptr(bias.contiguous()),
                                           None if residual is None else _ptr(residual), _ptr(y), M, N, K, 1 if relu else 0,
                                           ALO_BF16, _stream(x.device)))
    return y.view(*x.shape[:-1], N)


def linear_packed_supported(x, weight):
    """bf16 CUDA, K % 256 == 0 (K >= 512: below that linear_shortk keeps the weights in registers), N % 128 == 0, inference."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.dim() == 2
            and x.shape[-1] == weight.shape[1] and weight.shape[1] % 256 == 0 and weight.shape[1] >= 512
            and weight.shape[0] % 128 == 0 and not torch.is_grad_enabled())


def linear_packed(x, weight, bias=None, relu=False, residual=None):
    """``act(F.linear(x, weight, bias) [+ residual])`` with the weight streamed in MFMA fragment order (packed once per weight
    version, cached on the tensor)."""
    if not linear_packed_supported(x, weight):
        raise RuntimeError("linear_packed: needs bf16 CUDA tensors, K % 256 == 0, K >= 512, N % 128 == 0, no autograd")
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M, K = x2.shape
    N = weight.shape[0]
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if residual is not None:
        residual = residual.reshape(M, N)
        if residual.dtype != x.dtype or not residual.is_contiguous():
            raise RuntimeError("linear_packed: residual must be a contiguous (M, N) tensor of the input's dtype")
    if M:
        packed = pack_mfma_b(weight)
        bias_c = None if bias is None else bias.to(x.dtype).contiguous()
        with torch.cuda.device(x.device), _timed(f"linear_packed/K={K}/N={N}", 2.0 * (M * K + M * N * (2 if residual is not None else 1)),
                                                 2.0 * M * N * K):
            _check(lib().alo_linear_packed(_ptr(x2), _ptr(packed), None if bias_c is None else _ptr(bias_c),
                                           None if residual is None else _ptr(residual), _ptr(y), M, N, K, 1 if relu else 0,
                                           ALO_BF16, _stream(x.device)))
    return y.view(*x.shape[:-1], N)


def conv1x1_strided_supported(x, weight2d):
    """Strided 1x1 convolution of a channels-last bf16 map addressed inside the GEMM's tile loader: the shapes linear_auto
    would send to one of the streaming kernels."""
    if not (x.dim() == 4 and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    if os.environ.get("ALO_CONV1X1_GATHER") == "0":   # A/B knob: gather the kept pixels with a copy kernel first
        return False
    rows = x.permute(0, 2, 3, 1)
    if linear_shortk_supported(rows, weight2d):
        return True
    return linear_packed_supported(rows, weight2d) and (weight2d.shape[0] >= 1024 or tuple(weight2d.shape) == (128, 512))


def conv1x1_strided(x, weight2d, bias, stride, relu=False):
    """``act(F.conv2d(x, weight2d[:, :, None, None], bias, stride))`` for a channels-last bf16 ``x``; returns channels-last."""
    if not conv1x1_strided_supported(x, weight2d):
        raise RuntimeError("conv1x1_strided: unsupported dtype / layout / shape")
    n, cin, h, w_ = x.shape
    cout = weight2d.shape[0]
    ho, wo = (h - 1) // stride + 1, (w_ - 1) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    packed = not linear_shortk_supported(x.permute(0, 2, 3, 1), weight2d)
    wt = pack_mfma_b(weight2d) if packed else weight2d.contiguous()
    bias_c = None if bias is None else bias.to(x.dtype).contiguous()
    with torch.cuda.device(x.device), _timed(f"conv1x1_strided/K={cin}/N={cout}", 2.0 * (y.numel() // cout * cin + y.numel()),
                                             2.0 * y.numel() * cin):
        _check(lib().alo_conv1x1_nhwc(_ptr(x), _ptr(wt), 1 if packed else 0, None if bias_c is None else _ptr(bias_c), None, _ptr(y),
                                      n, h, w_, cin, cout, stride, 1 if relu else 0, ALO_BF16, _stream(x.device)))
    return y


def linear_auto(x, weight, bias=None, relu=False, residual=None):
    """Inference-time ``act(F.linear(x, weight, bias) [+ residual])``: the streaming MFMA kernels when the shape allows it
    (bf16; K in {64, 128, 256} with N % 64 == 0, or K % 256 == 0 with N % 128 == 0), otherwise the stock GEMM with the bias /
    ReLU epilogue."""
    if linear_shortk_supported(x, weight) and (bias is None or bias.dtype == x.dtype):
        return linear_shortk(x, weight, bias, relu, residual=residual)
    if linear_packed_supported(x, weight) and (bias is None or bias.dtype == x.dtype) and (
            residual is not None or weight.shape[0] >= 1024 or tuple(weight.shape) == (128, 512)):
        # measured on MI355X at the backbone's shapes: the streaming kernel wins with many output columns, with the identity
        # fused in, and at (N, K) = (128, 512); hipBLASLt wins the rest
        return linear_packed(x, weight, bias, relu, residual=residual)
    x2 = x.reshape(-1, x.shape[-1])
    fused_act = relu and residual is None
    if bias is not None:
        y = torch._addmm_activation(bias, x2, weight.t(), use_gelu=False) if fused_act else torch.addmm(bias, x2, weight.t())
    else:
        y = torch.mm(x2, weight.t())
        y = torch.relu_(y) if fused_act else y
    if residual is not None:
        y = y + residual.reshape(y.shape)
        y = torch.relu_(y) if relu else y
    return y.view(*x.shape[:-1], weight.shape[0])


def ffn256_supported(x, w1, w2):
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] == 256 and w1.dtype == torch.bfloat16
            and w2.dtype == torch.bfloat16 and w1.dim() == 2 and w1.shape[1] == 256 and w1.shape[0] % 256 == 0
            and tuple(w2.shape) == (256, w1.shape[0]))


def pack_mfma_b(weight):
    """(N, K) bf16 weight -> MFMA B-fragment order.  The packed copy rides on the weight tensor object itself, tagged
    with the version counter it was made from: packed once per weight update, gone with the tensor."""
    tag = (tensor_version(weight), weight.data_ptr())
    hit = getattr(weight, "_alo_packed", None)
    if hit is None or hit[0] != tag:
        w = weight.detach().contiguous()
        packed = torch.empty_like(w)
        with torch.cuda.device(w.device):
            _check(lib().alo_pack_mfma_b(_ptr(w), _ptr(packed), w.shape[0], w.shape[1], ALO_BF16, _stream(w.device)))
        hit = (tag, packed)
        weight._alo_packed = hit
    return hit[1]


def ffn256(x, w1, b1, w2, b2):
    """``relu(x @ w1.T + b1) @ w2.T + b2`` over the last dim (= 256) of a bf16 ``x`` in one kernel (hidden width % 256 == 0)."""
    if not ffn256_supported(x, w1, w2):
        raise RuntimeError("ffn256: needs bf16 CUDA tensors, d_model = 256 and a hidden width that is a multiple of 256")
    x2 = x.reshape(-1, 256)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = torch.empty_like(x2)
    M, Fh = x2.shape[0], w1.shape[0]
    if M:
        p1, p2 = pack_mfma_b(w1), pack_mfma_b(w2)
        with torch.cuda.device(x.device), _timed(f"ffn256/F={Fh}", 4.0 * x2.numel(), 4.0 * M * 256 * Fh):
            _check(lib().alo_ffn256(_ptr(x2), _ptr(p1), None if b1 is None else _ptr(b1.contiguous()),
                                    _ptr(p2), None if b2 is None else _ptr(b2.contiguous()), _ptr(y), M, Fh,
                                    ALO_BF16, _stream(x.device)))
    return y.view(x.shape)


def conv3x3_supported(x, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1):
    """3x3 / padding 1 / stride 1 or 2 convolution of a channels-last bf16 CUDA activation, Cin % 64 == 0, Cout % 64 == 0."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4 and weight.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) in ((1, 1), (2, 2)) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1] and x.shape[1] % 64 == 0
            and weight.shape[0] % 64 == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and not torch.is_grad_enabled())


def conv3x3(x, weight, bias=None, relu=False, stride=1):
    """``act(F.conv2d(x, weight, bias, stride, 1))`` for a channels-last bf16 ``x`` (N, Cin, H, W): implicit GEMM on MFMA with
    the bias and the ReLU in its epilogue.  Returns a channels-last (N, Cout, Ho, Wo) tensor."""
    stride = stride[0] if isinstance(stride, (tuple, list)) else stride
    if not conv3x3_supported(x, weight, (stride, stride)):
        raise RuntimeError("conv3x3: needs a channels-last bf16 CUDA activation, a (Cout, Cin, 3, 3) weight, Cin % 64 == 0, "
                           "Cout % 64 == 0, stride 1 or 2, no autograd")
    n, cin, h, w_ = x.shape
    cout = weight.shape[0]
    tag = (tensor_version(weight), weight.data_ptr())
    hit = getattr(weight, "_alo_packed", None)
    if hit is None or hit[0] != tag:
        # (Cout, ky, kx, Cin) row-major = the channels-last memory of the weight; pack it as a (Cout, 9 Cin) matrix
        wm = weight.detach().permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
        packed = torch.empty_like(wm)
        with torch.cuda.device(x.device):
            _check(lib().alo_pack_mfma_b(_ptr(wm), _ptr(packed), cout, 9 * cin, ALO_BF16, _stream(x.device)))
        hit = (tag, packed)
        weight._alo_packed = hit
    ho, wo = (h - 1) // stride + 1, (w_ - 1) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    bias_c = None if bias is None else bias.contiguous()
    ws_bytes = lib().alo_conv3x3_workspace_bytes(n, h, w_, cin, cout, stride)   # split-K partial sums (few-tile shapes only)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device) if ws_bytes else None
    with torch.cuda.device(x.device), _timed(f"conv3x3/C={cin}/s={stride}", 2.0 * (x.numel() + y.numel()), 2.0 * 9 * cin * y.numel()):
        _check(lib().alo_conv3x3_nhwc(_ptr(x), _ptr(hit[1]), None if bias_c is None else _ptr(bias_c), _ptr(y),
                                      None if ws is None else _ptr(ws), n, h, w_, cin, cout, stride, 1 if relu else 0, ALO_BF16,
                                      _stream(x.device)))
    return y


def stem_conv_pool_supported(x, weight):
    """bf16 CUDA (N, 3, H, W) image (any strides), (64, 3, 7, 7) bf16 weight, inference."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] == 3
            and tuple(weight.shape) == (64, 3, 7, 7) and not torch.is_grad_enabled())


def stem_conv_pool(x, weight, bias=None):
    """``max_pool2d(relu(conv2d(x, weight, bias, stride=2, padding=3)), 3, 2, 1)`` — the ResNet stem — in one kernel.
    Returns a channels-last (N, 64, Hp, Wp) bf16 tensor."""
    if not stem_conv_pool_supported(x, weight):
        raise RuntimeError("stem_conv_pool: needs a bf16 CUDA (N, 3, H, W) image, a (64, 3, 7, 7) bf16 weight, no autograd")
    n, _, h, w_ = x.shape
    tag = (tensor_version(weight), weight.data_ptr())
    hit = getattr(weight, "_alo_packed", None)
    if hit is None or hit[0] != tag:
        # (64, 7 tap rows x 24): per tap row the 7 taps x 3 channels interleaved as the image rows are, then 3 zero columns
        wm = torch.zeros((64, 8, 24), dtype=weight.dtype, device=weight.device)
        wm[:, :7, :21] = weight.detach().permute(0, 2, 3, 1).reshape(64, 7, 21)
        wm = wm.reshape(64, 192)[:, :176].contiguous()
        packed = torch.empty_like(wm)
        with torch.cuda.device(x.device):
            _check(lib().alo_pack_mfma_b(_ptr(wm), _ptr(packed), 64, 176, ALO_BF16, _stream(x.device)))
        hit = (tag, packed)
        weight._alo_packed = hit
    hc, wc = (h - 1) // 2 + 1, (w_ - 1) // 2 + 1
    hp, wp = (hc - 1) // 2 + 1, (wc - 1) // 2 + 1
    y = torch.empty((n, 64, hp, wp), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    bias_c = None if bias is None else bias.contiguous()
    sn, sc, sh, sw = x.stride()
    with torch.cuda.device(x.device), _timed("stem_conv_pool", 2.0 * (x.numel() + y.numel()), 2.0 * 147 * 64 * n * hc * wc):
        _check(lib().alo_stem_conv_pool(_ptr(x), _ptr(hit[1]), None if bias_c is None else _ptr(bias_c), _ptr(y), n, h, w_,
                                        sn, sc, sh, sw, ALO_BF16, _stream(x.device)))
    return y


def groupnorm_rows_supported(x, weight, groups, narrow=False):
    """bf16 CUDA channels-last rows (B, HW, C); C / 8 and ``groups`` divide 256; whole 8-channel slices per group — or, with
    ``narrow`` (``groupnorm_nhwc``), 2 or 4 channels per group."""
    c_ = x.shape[-1]
    cpg = c_ // groups if groups and c_ % groups == 0 else 0
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3 and weight is not None and weight.dtype == torch.bfloat16
            and c_ % 8 == 0 and 256 % (c_ // 8) == 0 and cpg > 0 and (cpg % 8 == 0 or (narrow and cpg in (2, 4))) and 256 % groups == 0
            and not torch.is_grad_enabled())


def groupnorm_nhwc_supported(x, norm):
    """``groupnorm_nhwc`` covers: bf16 CUDA (N, C, H, W) channels-last maps, affine GroupNorm with 2, 4 or a multiple of 8 channels
    per group, no autograd."""
    return (x.dim() == 4 and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last) and norm.affine
            and groupnorm_rows_supported(x.new_empty((1, 1, x.shape[1])), norm.weight, norm.num_groups, narrow=True))


def groupnorm_nhwc(x, norm, relu=False):
    """``relu?(norm(x))`` for a channels-last bf16 map (N, C, H, W) and an ``nn.GroupNorm``; channels-last result.  ATen's GroupNorm
    works on NCHW: on a channels-last activation it costs a layout copy before and (for the next convolution) after."""
    if not groupnorm_nhwc_supported(x, norm):
        raise RuntimeError("groupnorm_nhwc: needs a channels-last bf16 CUDA map and 2, 4 or 8k channels per group, no autograd")
    n, c_, h, w_ = x.shape
    rows = x.permute(0, 2, 3, 1)            # (N, H, W, C) view of the same memory
    out = torch.empty_like(x)               # preserves channels-last
    if n and h * w_:
        nbytes = lib().alo_groupnorm_rows_workspace_bytes(n, h * w_, norm.num_groups)
        ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device), _timed(f"groupnorm_nhwc/C={c_}", 6.0 * x.numel()):
            _check(lib().alo_groupnorm_rows_act(_ptr(rows), _ptr(norm.weight.contiguous()), _ptr(norm.bias.contiguous()), _ptr(out),
                                                _ptr(ws), n, h * w_, c_, norm.num_groups, float(norm.eps), h * w_ * c_,
                                                1 if relu else 0, ALO_BF16, _stream(x.device)))
    return out


def conv3x3_small_supported(x, conv):
    """``conv3x3_small`` covers: an ``nn.Conv2d`` 3x3 / stride 1 / padding 1 with Cin in (16, 32, 64) and Cout = 1 or 4k <= 32 on a
    channels-last bf16 CUDA map, no autograd."""
    w = conv.weight
    return (isinstance(conv, torch.nn.Conv2d) and x.dim() == 4 and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and w.shape[1] == x.shape[1] and x.shape[1] in (16, 32, 64)
            and (w.shape[0] == 1 or (w.shape[0] % 4 == 0 and w.shape[0] <= 32))
            and x.is_contiguous(memory_format=torch.channels_last) and not torch.is_grad_enabled())


def _small_conv_operands(conv):
    """(w_frag, bias32) of alo_conv3x3_small_nhwc, cached on the module per weight / bias version."""
    w, b = conv.weight, conv.bias
    key = (tensor_version(w), w.data_ptr(), None if b is None else (tensor_version(b), b.data_ptr()))
    hit = conv.__dict__.get("_alo_small_frag")
    if hit is None or hit[0] != key:
        cout, cin = w.shape[:2]
        with torch.no_grad():
            full = torch.zeros((32, 3, 3, cin), dtype=w.dtype, device=w.device)
            full[:cout] = w.detach().permute(0, 2, 3, 1)                                   # (m, ky, kx, c)
            frag = full.view(32, 9, cin // 16, 2, 8).permute(1, 2, 3, 0, 4).contiguous()   # (tap, cs, kg, m, 8) = [k-step][lane][8]
            bias32 = torch.zeros(32, dtype=torch.float32, device=w.device)
            if b is not None:
                bias32[:cout] = b.detach().float()
        hit = (key, frag, bias32)
        conv.__dict__["_alo_small_frag"] = hit
    return hit[1], hit[2]


def conv3x3_small(x, conv):
    """``conv(x)`` for the few-channel 3x3 convolutions of the mask decoder (channels-last bf16 in and out)."""
    if not conv3x3_small_supported(x, conv):
        raise RuntimeError("conv3x3_small: needs a channels-last bf16 CUDA map, a 3x3 / stride 1 / padding 1 convolution with Cin in "
                           "(16, 32, 64) and Cout = 1 or 4k <= 32, no autograd")
    n, cin, h, w_ = x.shape
    cout = conv.weight.shape[0]
    frag, bias32 = _small_conv_operands(conv)
    y = torch.empty((n, cout, h, w_), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if y.numel():
        with torch.cuda.device(x.device), _timed(f"conv3x3_small/C={cin}->{cout}", 2.0 * (x.numel() + y.numel()), 2.0 * 9 * cin * cout * n * h * w_):
            _check(lib().alo_conv3x3_small_nhwc(_ptr(x), _ptr(frag), _ptr(bias32), _ptr(y), n, h, w_, cin, cout, ALO_BF16, _stream(x.device)))
    return y


def upsample_add_supported(x_low, fpn):
    return (x_low.dim() == 4 and fpn.dim() == 4 and x_low.is_cuda and x_low.dtype == torch.bfloat16 and fpn.dtype == torch.bfloat16
            and x_low.shape[1] == fpn.shape[1] and x_low.shape[1] % 8 == 0 and fpn.shape[0] > 0 and x_low.shape[0] % fpn.shape[0] == 0
            and x_low.is_contiguous(memory_format=torch.channels_last) and fpn.is_contiguous(memory_format=torch.channels_last)
            and not torch.is_grad_enabled())


def upsample_add(x_low, fpn):
    """``fpn.repeat_interleave(Q, 0) + F.interpolate(x_low, size=fpn.shape[-2:], mode="nearest")`` in one pass (Q = x_low.shape[0]
    // fpn.shape[0]); channels-last bf16 in and out, bit-identical to the stock ops."""
    if not upsample_add_supported(x_low, fpn):
        raise RuntimeError("upsample_add: needs channels-last bf16 CUDA maps with C % 8 == 0 and x_low.shape[0] a multiple of fpn.shape[0]")
    bq, c_, h, w_ = x_low.shape
    b_, _, H, W = fpn.shape
    out = torch.empty((bq, c_, H, W), dtype=x_low.dtype, device=x_low.device, memory_format=torch.channels_last)
    if out.numel():
        with torch.cuda.device(x_low.device), _timed(f"upsample_add/C={c_}", 2.0 * (out.numel() + x_low.numel() + fpn.numel())):
            _check(lib().alo_upsample_add_nhwc(_ptr(x_low), _ptr(fpn), _ptr(out), bq, bq // b_, c_, h, w_, H, W, ALO_BF16, _stream(x_low.device)))
    return out


def groupnorm_rows(x, weight, bias, groups, eps=1e-5, out=None):
    """``F.group_norm`` over channels-last rows: x (B, HW, C) contiguous -> out (B, HW, C), which may be a slice
    ``flat[:, start:start + HW]`` of a larger (B, S, C) buffer (rows contiguous, any batch stride)."""
    if not groupnorm_rows_supported(x, weight, groups):
        raise RuntimeError("groupnorm_rows: needs bf16 CUDA rows (B, HW, C) with C / 8 and groups dividing 256, no autograd")
    x = x.contiguous()
    b_, hw, c_ = x.shape
    if out is None:
        out = torch.empty_like(x)
    if tuple(out.shape) != (b_, hw, c_) or out.dtype != x.dtype or out.stride(2) != 1 or out.stride(1) != c_:
        raise RuntimeError("groupnorm_rows: out must be (B, HW, C) of the input's dtype with contiguous rows")
    if b_ and hw:
        nbytes = lib().alo_groupnorm_rows_workspace_bytes(b_, hw, groups)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device), _timed(f"groupnorm_rows/HW={hw}", 6.0 * x.numel()):
            _check(lib().alo_groupnorm_rows(_ptr(x), _ptr(weight.contiguous()), _ptr(bias.contiguous()), _ptr(out), _ptr(ws), b_, hw,
                                            c_, groups, float(eps), out.stride(0) if b_ > 1 else hw * c_, ALO_BF16,
                                            _stream(x.device)))
    return out


def _host_shapes(shapes):
    flat = [int(v) for hw in shapes for v in hw]
    return (ctypes.c_int * len(flat))(*flat), len(flat) // 2


def mask_pyramid(frame_mask, shapes, nearest_levels=()):
    """Padding mask of every level + valid ratios from the frame mask, in two small kernels.

    frame_mask: (B, H, W) or (B, 1, H, W), float32 or bool / uint8, non-zero on padding.  shapes: [(h_l, w_l)].  Level l is
    resized like ``F.interpolate(mask.float(), (h_l, w_l), mode="bilinear", align_corners=False) != 0``, or with
    ``mode="nearest"`` for l in ``nearest_levels``.  Returns (mask_flat (B, S) bool, valid_ratios (B, L, 2) float32 as (w, h))."""
    if frame_mask.dim() == 4:
        frame_mask = frame_mask[:, 0]
    if not frame_mask.is_cuda or frame_mask.dim() != 3:
        raise RuntimeError("mask_pyramid: needs a CUDA (B, H, W) or (B, 1, H, W) mask")
    if frame_mask.dtype not in (torch.float32, torch.bool, torch.uint8):
        frame_mask = frame_mask != 0
    frame_mask = frame_mask.contiguous()
    b_, h, w_ = frame_mask.shape
    arr, L = _host_shapes(shapes)
    S = sum(int(a) * int(b) for a, b in shapes)
    mask_flat = torch.empty((b_, S), dtype=torch.uint8, device=frame_mask.device)
    ratios = torch.empty((b_, L, 2), dtype=torch.float32, device=frame_mask.device)
    bits = 0
    for l in nearest_levels:
        bits |= 1 << int(l)
    with torch.cuda.device(frame_mask.device), _timed("mask_pyramid", float(frame_mask.numel() + mask_flat.numel())):
        _check(lib().alo_mask_pyramid(_ptr(frame_mask), 1 if frame_mask.dtype == torch.float32 else 0, _ptr(mask_flat), _ptr(ratios),
                                      b_, h, w_, L, arr, bits, _stream(frame_mask.device)))
    return mask_flat.view(torch.bool), ratios


def encoder_reference_points(valid_ratios, shapes):
    """(B, S, L, 2) float32 reference points of the encoder (pixel centres over the valid extent) in one kernel."""
    if not valid_ratios.is_cuda or valid_ratios.dtype != torch.float32 or valid_ratios.dim() != 3 or valid_ratios.shape[2] != 2:
        raise RuntimeError("encoder_reference_points: needs CUDA float32 valid ratios of shape (B, L, 2)")
    valid_ratios = valid_ratios.contiguous()
    arr, L = _host_shapes(shapes)
    if L != valid_ratios.shape[1]:
        raise RuntimeError("encoder_reference_points: one (h, w) per level of valid_ratios")
    S = sum(int(a) * int(b) for a, b in shapes)
    out = torch.empty((valid_ratios.shape[0], S, L, 2), dtype=torch.float32, device=valid_ratios.device)
    with torch.cuda.device(valid_ratios.device), _timed("encoder_reference_points", 4.0 * out.numel()):
        _check(lib().alo_encoder_reference_points(_ptr(valid_ratios), _ptr(out), valid_ratios.shape[0], L, arr,
                                                  _stream(valid_ratios.device)))
    return out


def value_proj_head_major_supported(x, weight, heads):
    return (linear_shortk_supported(x, weight) and heads % 2 == 0 and weight.shape[0] == heads * 32 and x.dim() == 3)


def value_proj_head_major(x, weight, bias, padding_mask, heads):
    """``value_proj`` + ``masked_fill(padding_mask, 0)`` + head-major layout in one kernel: x (N, S, K) bf16 ->
    (N, heads, S, 32) for ``msda_forward_fused_hm``."""
    if not value_proj_head_major_supported(x, weight, heads):
        raise RuntimeError("value_proj_head_major: needs bf16 (N, S, K) input, K in (64, 128, 256), head dimension 32")
    N, S, K = x.shape
    x = x if x.is_contiguous() else x.contiguous()
    if padding_mask is not None:
        if padding_mask.dtype != torch.bool or tuple(padding_mask.shape) != (N, S):
            raise RuntimeError("padding_mask must be a (N, S) bool tensor")
        padding_mask = padding_mask.contiguous()
    out = torch.empty((N, heads, S, 32), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _timed(f"value_proj_hm/S={S}", 2.0 * (x.numel() + out.numel()), 2.0 * N * S * heads * 32 * K):
        _check(lib().alo_value_proj_head_major(_ptr(x), _ptr(weight.contiguous()), None if bias is None else _ptr(bias.contiguous()),
                                               None if padding_mask is None else _ptr(padding_mask), _ptr(out), N, S, heads, K,
                                               ALO_BF16, _stream(x.device)))
    return out



def panoptic_onehot(mask_logits, frame_size, threshold=0.5):
    """(B, Q, h, w) mask logits -> (B, Q, H, W) int64 one-hot instance masks: bilinear up-sampling, sigmoid, threshold and the
    per-pixel arg-max over the queries in one pass (PanopticHead.inference)."""
    if not mask_logits.is_cuda or mask_logits.dim() != 4:
        raise RuntimeError("panoptic_onehot: needs CUDA (B, Q, h, w) logits")
    x = mask_logits.float().contiguous()
    b_, q, h, w_ = x.shape
    H, W = int(frame_size[0]), int(frame_size[1])
    out = torch.empty((b_, q, H, W), dtype=torch.long, device=x.device)
    if out.numel():
        with torch.cuda.device(x.device), _timed("panoptic_onehot", 8.0 * out.numel()):
            _check(lib().alo_panoptic_onehot(_ptr(x), _ptr(out), b_, q, h, w_, H, W, float(threshold), _stream(x.device)))
    return out
