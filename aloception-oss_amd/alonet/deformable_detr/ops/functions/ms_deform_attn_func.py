"""Operator boundary of multi-scale deformable attention, backed by the gfx950 HIP kernels.

Mirrors the public names of the reference's ``alonet/deformable_detr/ops/functions/ms_deform_attn_func.py``:

* ``load_ops`` / ``load_MultiScaleDeformableAttention``  (reference :22-46) — the reference ``torch.ops.load_library``s
  a torch C++ extension and shells out to ``make.sh`` when it is missing.  Here they load ``libalo_hotpath.so`` through
  ctypes and register ``alonet_custom::ms_deform_attn_forward`` / ``..._backward`` with the reference's schema
  (``ops/src/vision.cpp:21-24``), so ``torch.ops.alonet_custom.*`` keeps resolving.
* ``MSDeformAttnFunction`` (reference :49-82) — autograd glue around the two dispatcher ops.
* ``ms_deform_attn_core_pytorch`` (reference :85-107) — the pure-torch formulation the reference keeps for ONNX /
  TorchScript tracing (``is_tracing`` branch of ``MSDeformAttn.forward``).  It is kept for that explicit branch only;
  nothing in this package falls back to it: on a missing library or a CPU tensor the ops raise.
* ``bilinear_grid_sample`` (reference :110-190) — the export-friendly stand-in for ``F.grid_sample(mode="bilinear",
  padding_mode="zeros")`` the reference's pure-torch formulation calls; same signature, written with index masks.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import alo_hip

_FWD_SCHEMA = (
    "ms_deform_attn_forward(Tensor value, Tensor spatial_shapes, Tensor level_start_index, "
    "Tensor sampling_loc, Tensor attn_weight, int im2col_step) -> Tensor"
)
_BWD_SCHEMA = (
    "ms_deform_attn_backward(Tensor value, Tensor spatial_shapes, Tensor level_start_index, "
    "Tensor sampling_loc, Tensor attn_weight, Tensor grad_output, int im2col_step) -> Tensor[]"
)
_registered = None  # keeps the torch.library.Library objects alive


def _fwd_meta(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    N, _, M, D = value.shape
    return value.new_empty((N, sampling_loc.shape[1], M * D))


def _bwd_meta(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    return [torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)]


def _cpu_stub(*args):
    raise RuntimeError("Not implemented on the CPU")


def load_ops():
    """Load libalo_hotpath.so and register the ``alonet_custom`` dispatcher ops (idempotent)."""
    global _registered
    alo_hip.lib()  # raises HotpathUnavailable (a RuntimeError) when the library cannot be loaded
    if _registered is not None:
        return
    if hasattr(torch.ops.alonet_custom, "ms_deform_attn_forward"):
        raise RuntimeError("alonet_custom::ms_deform_attn_forward is already registered by another library")
    define = torch.library.Library("alonet_custom", "DEF")
    define.define(_FWD_SCHEMA)
    define.define(_BWD_SCHEMA)
    define.impl("ms_deform_attn_forward", alo_hip.msda_forward, "CUDA")
    define.impl("ms_deform_attn_backward", alo_hip.msda_backward, "CUDA")
    define.impl("ms_deform_attn_forward", _cpu_stub, "CPU")
    define.impl("ms_deform_attn_backward", _cpu_stub, "CPU")
    define.impl("ms_deform_attn_forward", _fwd_meta, "Meta")
    define.impl("ms_deform_attn_backward", _bwd_meta, "Meta")
    _registered = define


def load_MultiScaleDeformableAttention():
    """Must run once before ``MSDeformAttnFunction`` is used (``MSDeformAttn.__init__`` calls it).

    Builds the library with hipcc first if it has not been built yet — the counterpart of the reference's
    "build on first use"; raises if neither works.
    """
    try:
        load_ops()
    except alo_hip.HotpathUnavailable:
        print("Building the gfx950 ms_deform_attn kernels (hipcc) ...")
        alo_hip.build()
        load_ops()


class MSDeformAttnFunction(Function):
    """``apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)``"""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        load_ops()
        ctx.im2col_step = im2col_step
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return torch.ops.alonet_custom.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step
        )

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, loc, attn = ctx.saved_tensors
        g_value, g_loc, g_attn = torch.ops.alonet_custom.ms_deform_attn_backward(
            value, shapes, start, loc, attn, grad_output.contiguous(), ctx.im2col_step
        )
        return g_value, None, None, g_loc, g_attn, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Pure-torch multi-scale deformable attention for the tracing/export branch (any device, any float dtype).

    ``value (N,S,M,D)``, ``value_spatial_shapes (L,2) [H,W]``, ``sampling_locations (N,Lq,M,L,P,2)`` in [0,1],
    ``attention_weights (N,Lq,M,L,P)`` -> ``(N, Lq, M*D)``.  Same map as the HIP op: bilinear, zero padding,
    ``align_corners=False`` (pixel centres at half-integers).
    """
    N, _, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [(int(h), int(w)) for h, w in value_spatial_shapes]
    per_level = value.split([h * w for h, w in sizes], dim=1)
    grids = sampling_locations * 2 - 1
    sampled = []
    for lvl, (h, w) in enumerate(sizes):
        feat = per_level[lvl].permute(0, 2, 3, 1).reshape(N * M, D, h, w)
        grid = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        sampled.append(F.grid_sample(feat, grid, mode="bilinear", padding_mode="zeros", align_corners=False))
    sampled = torch.stack(sampled, dim=-2).reshape(N * M, D, Lq, L * P)
    weights = attention_weights.permute(0, 2, 1, 3, 4).reshape(N * M, 1, Lq, L * P)
    return (sampled * weights).sum(-1).reshape(N, M * D, Lq).transpose(1, 2).contiguous()


def bilinear_grid_sample(im, grid, align_corners=False):
    """``F.grid_sample(im, grid, mode="bilinear", padding_mode="zeros", align_corners=...)`` from elementary ops
    (floor / gather / multiply-add), for exporters without a grid-sample operator.

    ``im (N, C, H, W)``, ``grid (N, Hg, Wg, 2)`` with (x, y) in [-1, 1] -> ``(N, C, Hg, Wg)``.  Taps that fall outside
    the map contribute zero (their index is clamped for the gather and their weight is masked).
    """
    n, c, h, w = im.shape
    gn, gh, gw, two = grid.shape
    assert n == gn and two == 2
    gx, gy = grid[..., 0].reshape(n, -1), grid[..., 1].reshape(n, -1)
    if align_corners:
        x, y = (gx + 1) / 2 * (w - 1), (gy + 1) / 2 * (h - 1)
    else:
        x, y = ((gx + 1) * w - 1) / 2, ((gy + 1) * h - 1) / 2
    x0, y0 = torch.floor(x), torch.floor(y)
    fx, fy = x - x0, y - y0
    flat = im.reshape(n, c, h * w)
    out = None
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = x0 + dx, y0 + dy
            inside = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
            idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long()
            tap = torch.gather(flat, 2, idx[:, None, :].expand(-1, c, -1))
            term = tap * (wx * wy * inside.to(im.dtype))[:, None, :]
            out = term if out is None else out + term
    return out.reshape(n, c, gh, gw)
