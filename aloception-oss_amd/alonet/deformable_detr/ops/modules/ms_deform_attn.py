"""``MSDeformAttn`` — the multi-scale deformable attention layer of Deformable-DETR on the gfx950 op.

Same constructor, parameter names (state-dict keys ``sampling_offsets / attention_weights / value_proj /
output_proj``), initialisation and ``forward`` contract as the reference module
(alonet/deformable_detr/ops/modules/ms_deform_attn.py:34-155).  The four linear layers stay on stock PyTorch-ROCm
(hipBLASLt); the gather is ``MSDeformAttnFunction`` -> HIP.

Reduced-precision use (``module.bfloat16()``): ``value`` is produced and gathered in bf16, but sampling locations and
attention weights are evaluated in fp32 — an 8-bit mantissa on a location in [0,1] would move samples by a third of a
pixel on a 167-wide map.  With fp32/fp64 parameters the arithmetic is the reference's, operation for operation.
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

import alo_hip

from ..functions import MSDeformAttnFunction, load_MultiScaleDeformableAttention, ms_deform_attn_core_pytorch


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return n != 0 and (n & (n - 1)) == 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        """
        :param d_model   hidden dimension
        :param n_levels  number of feature levels
        :param n_heads   number of attention heads
        :param n_points  number of sampling points per attention head per feature level
        """
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn(
                "d_model // n_heads is not a power of 2: the gfx950 kernel then leaves lanes idle "
                "(channels per head are spread over a power-of-two lane group)."
            )
        self.im2col_step = 64  # kept for API compatibility; the HIP op has no batch chunking
        self.fused_prologue = True  # inference: fold softmax + location arithmetic into the kernel (alo_msda_forward_fused)
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

        load_MultiScaleDeformableAttention()

    def _reset_parameters(self):
        # offsets start as a ring of n_heads directions, point k pushed k+1 pixels out (reference :70-88)
        constant_(self.sampling_offsets.weight.data, 0.0)
        angle = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ring = torch.stack([angle.cos(), angle.sin()], -1)
        ring = ring / ring.abs().max(-1, keepdim=True)[0]
        ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        ring = ring * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, self.n_points, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(ring.reshape(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def _merged_query_projection(self):
        """[sampling_offsets; attention_weights] as one (3*M*L*P, C) weight + bias, rebuilt when either parameter changes."""
        so, aw = self.sampling_offsets, self.attention_weights
        key = (alo_hip.tensor_version(so.weight), alo_hip.tensor_version(so.bias), alo_hip.tensor_version(aw.weight), alo_hip.tensor_version(aw.bias), so.weight.data_ptr(), aw.weight.data_ptr(),
               so.weight.dtype, so.weight.device)
        hit = self.__dict__.get("_alo_merged")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, torch.cat([so.weight, aw.weight], 0).contiguous(), torch.cat([so.bias, aw.bias], 0).contiguous())
            self.__dict__["_alo_merged"] = hit
        return hit[1], hit[2]

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, **kwargs):
        """
        :param query                    (N, Lq, C)
        :param reference_points         (N, Lq, n_levels, 2) in [0,1] (top-left (0,0), bottom-right (1,1), padding included)
                                        or (N, Lq, n_levels, 4): (cx, cy, w, h) reference boxes
        :param input_flatten            (N, sum_l H_l*W_l, C)
        :param input_spatial_shapes     (n_levels, 2) [(H_0, W_0), ...]
        :param input_level_start_index  (n_levels,)
        :param input_padding_mask       (N, sum_l H_l*W_l) bool, True on padding
        :return                         (N, Lq, C)
        """
        N, Lq, _ = query.shape
        _, S, _ = input_flatten.shape
        total = getattr(input_spatial_shapes, "_alo_total", None)  # set by DeformableTransformer: no device sync
        if total is None:
            total = int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum())
        assert total == S
        M, L, P = self.n_heads, self.n_levels, self.n_points

        if reference_points.shape[-1] not in (2, 4):
            raise ValueError(
                "Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1])
            )
        # any trainable parameter of the layer counts (partial fine-tuning): the fused inference path has no backward
        needs_grad = torch.is_grad_enabled() and (
            any(t.requires_grad for t in (query, input_flatten, reference_points))
            or any(p.requires_grad for p in self.parameters()))
        fused = "is_tracing" not in kwargs and not needs_grad and self.fused_prologue and query.is_cuda
        # inference: the four K = d_model linears go through the streaming MFMA kernel when it fits (bf16, d_model = 256)
        proj = (lambda lin, t: alo_hip.linear_auto(t, lin.weight, lin.bias)) if fused else (lambda lin, t: lin(t))
        D = self.d_model // M
        hm = (fused and D == 32 and L == 4 and P == 4 and self.value_proj.bias is not None
              and alo_hip.value_proj_head_major_supported(input_flatten, self.value_proj.weight, M))
        if hm and self.sampling_offsets.bias is not None and self.attention_weights.bias is not None:
            # both projections of the query in ONE GEMM (query read once); the attention kernel takes its offsets and logits as
            # column slices of the merged result
            w_cat, b_cat = self._merged_query_projection()
            both = alo_hip.linear_auto(query, w_cat, b_cat)
            offsets = both[..., :M * L * P * 2].view(N, Lq, M, L, P, 2)
            logits = both[..., M * L * P * 2:].view(N, Lq, M, L * P)
        else:
            offsets = proj(self.sampling_offsets, query).view(N, Lq, M, L, P, 2)
            logits = proj(self.attention_weights, query).view(N, Lq, M, L * P)
        if hm:
            # inference, DETR-family shape: value_proj, the padding mask and the head-major layout are ONE kernel
            value = alo_hip.value_proj_head_major(input_flatten, self.value_proj.weight, self.value_proj.bias,
                                                  input_padding_mask, M)
            output = alo_hip.msda_forward_fused_hm(value, input_spatial_shapes, input_level_start_index, offsets, logits,
                                                   reference_points)
            return proj(self.output_proj, output)
        value = proj(self.value_proj, input_flatten)

        if fused and alo_hip.head_major_supported(value.view(N, S, M, self.d_model // M), L, P):
            # inference, DETR-family shape: padding is zeroed while the projection's output is re-laid head-major (one pass
            # instead of masked_fill), softmax + sampling-location arithmetic happen inside the kernel's descriptor stage
            value = alo_hip.value_head_major(value.view(N, S, M, self.d_model // M), input_padding_mask)
            output = alo_hip.msda_forward_fused_hm(value, input_spatial_shapes, input_level_start_index,
                                                   offsets.contiguous(), logits.contiguous(), reference_points)
            return proj(self.output_proj, output)

        if input_padding_mask is not None:
            if needs_grad:
                value = value.masked_fill(input_padding_mask[..., None], float(0))
            else:  # the projection's output is a fresh tensor: mask it in place, no clone
                value.masked_fill_(input_padding_mask[..., None], float(0))
        value = value.view(N, S, M, self.d_model // M)
        low_precision = value.dtype in (torch.bfloat16, torch.float16)
        geo = torch.float32 if low_precision else value.dtype

        if fused:
            # inference: softmax + sampling-location arithmetic happen inside the kernel's descriptor stage
            output = alo_hip.msda_forward_fused(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                                offsets.contiguous(), logits.contiguous(), reference_points)
            return proj(self.output_proj, output)

        offsets = offsets.to(geo)
        weights = F.softmax(logits.to(geo), -1).view(N, Lq, M, L, P)
        reference_points = reference_points.to(geo)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        else:
            locations = (
                reference_points[:, :, None, :, None, :2]
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
            )

        if "is_tracing" in kwargs:  # ONNX / TorchScript export branch of the reference (:138-144)
            output = ms_deform_attn_core_pytorch(value.to(geo), input_spatial_shapes, locations, weights).to(value.dtype)
        else:
            output = MSDeformAttnFunction.apply(
                value, input_spatial_shapes, input_level_start_index, locations, weights, self.im2col_step
            )
        return self.output_proj(output)
