"""Deformable transformer (encoder over all pyramid pixels, decoder over object queries) on the gfx950 MSDA op.

Module tree, parameter names and arithmetic follow alonet/deformable_detr/deformable_transformer.py:22-633 so that a
reference checkpoint's ``transformer.*`` keys load unchanged; every attention gather goes through
``MSDeformAttn`` -> ``MSDeformAttnFunction`` -> HIP.  The two-stage variant (never built by the reference's R50
constructors) is not provided.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

import alo_hip

from .ops.modules import MSDeformAttn
from .utils import inverse_sigmoid


_GEOMETRY = {}


def _level_geometry(shapes, device):
    """int32 ``spatial_shapes`` (L,2) / ``level_start_index`` (L,) on the device — what this fork of the op reads
    (ms_deform_attn_cuda.cu:67-68) — built once per (pyramid, device): a host->device copy per forward would also make the
    forward impossible to capture in a HIP graph.  The host-side copies ride along as attributes so that shape checks
    and loops downstream need no device synchronisation."""
    key = (shapes, str(device))
    if key not in _GEOMETRY:
        sizes = [h * w for h, w in shapes]
        # built outside inference mode whatever the caller's mode: an inference tensor cached here would later fail in
        # MSDeformAttnFunction's save_for_backward when a training step reuses the same pyramid shape
        with torch.inference_mode(False):
            spatial_shapes = torch.tensor(shapes, dtype=torch.int32, device=device)
            level_start_index = torch.tensor([sum(sizes[:i]) for i in range(len(sizes))], dtype=torch.int32, device=device)
        spatial_shapes._alo_total = sum(sizes)
        spatial_shapes._alo_shapes = list(shapes)
        _GEOMETRY[key] = (spatial_shapes, level_start_index)
    return _GEOMETRY[key]


def _fused_ok(module, kwargs, *tensors):
    """The one-pass HIP epilogues (alo_add_layernorm) stand in for ``norm(x + dropout(y))`` when nothing is lost: eval mode
    (dropout is the identity), no autograd graph, CUDA, fp32 / bf16, and not the pure-torch export branch."""
    return (not module.training and "is_tracing" not in kwargs and alo_hip.fusable(*tensors)
            and alo_hip.add_layernorm_supported(tensors[0]))


def _add_norm(norm, x, residual, pos=None):
    return alo_hip.add_layernorm(x, residual, norm.weight, norm.bias, norm.eps, pos=pos)


def _self_attention(mha, qk_in, v_in):
    """Inference form of ``nn.MultiheadAttention(q = k = qk_in, v = v_in)`` on batch-first (B, L, E) tensors: the packed
    input projection as two GEMMs (q and k share their input), ``scaled_dot_product_attention`` over the heads, the output
    projection — no sequence-first transposes, projections on the short-K MFMA kernel when the shape allows."""
    B, L, E = qk_in.shape
    H = mha.num_heads
    w, b = mha.in_proj_weight, mha.in_proj_bias
    qk = alo_hip.linear_auto(qk_in, w[: 2 * E], None if b is None else b[: 2 * E])
    v = alo_hip.linear_auto(v_in, w[2 * E:], None if b is None else b[2 * E:])
    q, k = qk[..., :E], qk[..., E:]
    heads = lambda t: t.reshape(B, L, H, E // H).transpose(1, 2)  # (B, H, L, E/H)
    out = F.scaled_dot_product_attention(heads(q), heads(k), heads(v))
    out = out.transpose(1, 2).reshape(B, L, E)
    return alo_hip.linear_auto(out, mha.out_proj.weight, mha.out_proj.bias)


def _ffn(linear1, activation, linear2, x):
    """``linear2(act(linear1(x)))``; for ReLU the activation rides in the first GEMM's epilogue (alo_linear_shortk when
    d_model is 64 / 128 / 256 and the tensors are bf16, else hipBLASLt's RELU_BIAS via ``torch._addmm_activation``) instead
    of a separate pass over the (rows, d_ffn) intermediate."""
    if activation is F.relu and alo_hip.ffn256_supported(x, linear1.weight, linear2.weight):
        # d_model = 256, bf16: both layers in one kernel, the hidden activation never leaves the chip
        return alo_hip.ffn256(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias)
    if activation is F.relu and linear1.bias is not None:
        h = alo_hip.linear_auto(x, linear1.weight, linear1.bias, relu=True)
        return alo_hip.linear_auto(h, linear2.weight, linear2.bias)
    return linear2(activation(linear1(x)))


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _get_activation_fn(activation):
    fns = {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}
    if activation not in fns:
        raise RuntimeError(f"activation should be relu/gelu, not {activation}.")
    return fns[activation]


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src):
        src2 = self.linear2(self.dropout2(self.activation(self.linear1(src))))
        return self.norm2(src + self.dropout3(src2))

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None, **kwargs):
        src2 = self.self_attn(self.with_pos_embed(src, pos), reference_points, src, spatial_shapes, level_start_index,
                              padding_mask, **kwargs)
        src = self.norm1(src + self.dropout1(src2))
        return self.forward_ffn(src)

    def forward_fused(self, src, query, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None,
                      next_query=True, **kwargs):
        """Inference form of ``forward``: ``query = src + pos`` comes in ready-made, both residual + LayerNorm pairs are
        one HIP pass each, and the second one also emits the next layer's query.  -> (src', src' + pos | None)"""
        src2 = self.self_attn(query, reference_points, src, spatial_shapes, level_start_index, padding_mask, **kwargs)
        src = _add_norm(self.norm1, src2, src)
        src2 = _ffn(self.linear1, self.activation, self.linear2, src)
        if next_query and pos is not None:
            return _add_norm(self.norm2, src2, src, pos=pos)
        return _add_norm(self.norm2, src2, src), None


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device, **kwargs):
        """Every pixel centre of every level, normalised by the valid (un-padded) extent: (B, S, L, 2)."""
        per_level = []
        host_shapes = getattr(spatial_shapes, "_alo_shapes", None)  # python copy set by DeformableTransformer: no device sync
        if (host_shapes is not None and valid_ratios.is_cuda and valid_ratios.dtype == torch.float32
                and not (torch.is_grad_enabled() and valid_ratios.requires_grad) and "is_tracing" not in kwargs):
            return alo_hip.encoder_reference_points(valid_ratios, host_shapes)   # one kernel instead of ~10 per level
        for lvl in range(spatial_shapes.shape[0]):
            h, w = host_shapes[lvl] if host_shapes is not None else (int(spatial_shapes[lvl, 0]), int(spatial_shapes[lvl, 1]))
            ys = torch.arange(h, dtype=torch.float32, device=device) + 0.5
            xs = torch.arange(w, dtype=torch.float32, device=device) + 0.5
            ref_y, ref_x = torch.meshgrid(ys, xs, indexing="ij")
            ref_y = ref_y.reshape(1, -1) / (valid_ratios[:, None, lvl, 1] * h)
            ref_x = ref_x.reshape(1, -1) / (valid_ratios[:, None, lvl, 0] * w)
            per_level.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(per_level, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None, **kwargs):
        output = src
        reference_points = self.get_reference_points(spatial_shapes, valid_ratios, device=src.device, **kwargs)
        if _fused_ok(self, kwargs, output, pos) and all(hasattr(layer, "forward_fused") for layer in self.layers):
            query = output if pos is None else output + pos
            for i, layer in enumerate(self.layers):
                output, query = layer.forward_fused(output, query, pos, reference_points, spatial_shapes, level_start_index,
                                                    padding_mask, next_query=i + 1 < len(self.layers), **kwargs)
                if query is None:
                    query = output
            return output
        for layer in self.layers:
            output = layer(output, pos, reference_points, spatial_shapes, level_start_index, padding_mask, **kwargs)
        return output


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, dim_feedforward=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8,
                 n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        tgt2 = self.linear2(self.dropout3(self.activation(self.linear1(tgt))))
        return self.norm3(tgt + self.dropout4(tgt2))

    def pre_process_tgt(self, tgt, query_pos, tgt_key_padding_mask, **kwargs):
        return tgt, query_pos, tgt_key_padding_mask

    def decoder_layer_forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                              tgt_key_padding_mask=None, src_padding_mask=None, **kwargs):
        carried = tgt.__dict__.get("_alo_with_pos")  # (query_pos, tgt + query_pos) left by the previous layer's last LayerNorm kernel
        if carried is not None and carried[0] is query_pos:
            q = k = carried[1]
        else:
            q = k = self.with_pos_embed(tgt, query_pos)  # self-attention among the queries (sequence-first API)
        if _fused_ok(self, kwargs, tgt, query_pos) and tgt_key_padding_mask is None and self.self_attn._qkv_same_embed_dim:
            tgt2 = _self_attention(self.self_attn, q, tgt)
        else:
            tgt2 = self.self_attn(q.transpose(0, 1), k.transpose(0, 1), tgt.transpose(0, 1),
                                  key_padding_mask=tgt_key_padding_mask)[0].transpose(0, 1)
        if _fused_ok(self, kwargs, tgt, tgt2, query_pos):  # inference: residual + LayerNorm (+ query_pos) in one pass each
            if query_pos is None:
                tgt = query = _add_norm(self.norm2, tgt2, tgt)
            else:
                tgt, query = _add_norm(self.norm2, tgt2, tgt, pos=query_pos.expand_as(tgt))
            tgt2 = self.cross_attn(query, reference_points, src, src_spatial_shapes, level_start_index,
                                   src_padding_mask, **kwargs)
            tgt = _add_norm(self.norm1, tgt2, tgt)
            if query_pos is None or not query_pos.is_contiguous():
                return _add_norm(self.norm3, _ffn(self.linear1, self.activation, self.linear2, tgt), tgt)
            # the layer's output and (output + query_pos), the next layer's self-attention input, from the same pass
            out, out_pos = _add_norm(self.norm3, _ffn(self.linear1, self.activation, self.linear2, tgt), tgt, pos=query_pos)
            out._alo_with_pos = (query_pos, out_pos)
            return out
        tgt = self.norm2(tgt + self.dropout2(tgt2))
        tgt2 = self.cross_attn(self.with_pos_embed(tgt, query_pos), reference_points, src, src_spatial_shapes,
                               level_start_index, src_padding_mask, **kwargs)
        tgt = self.norm1(tgt + self.dropout1(tgt2))
        return self.forward_ffn(tgt)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                tgt_key_padding_mask=None, src_padding_mask=None, **kwargs):
        tgt, query_pos, tgt_key_padding_mask = self.pre_process_tgt(tgt, query_pos, tgt_key_padding_mask, **kwargs)
        return self.decoder_layer_forward(tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                                          tgt_key_padding_mask, src_padding_mask, **kwargs)


class DeformableTransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.bbox_embed = None  # set by DeformableDETR for iterative box refinement
        self.class_embed = None

    def pre_process_tgt(self, tgt, query_pos, tgt_key_padding_mask, reference_points, **kwargs):
        return tgt, query_pos, tgt_key_padding_mask, reference_points

    def decoder_forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                        query_pos=None, src_padding_mask=None, tgt_key_padding_mask=None, **kwargs):
        output = tgt
        intermediate, intermediate_refs = [], []
        ref_input = None
        for lid, layer in enumerate(self.layers):
            if ref_input is None or self.bbox_embed is not None:  # without box refinement the reference points never change
                if reference_points.shape[-1] == 4:
                    ratios = torch.cat([src_valid_ratios, src_valid_ratios], -1)
                    ref_input = reference_points[:, :, None] * ratios[:, None]
                else:
                    assert reference_points.shape[-1] == 2
                    ref_input = reference_points[:, :, None] * src_valid_ratios[:, None]
            output = layer(tgt=output, query_pos=query_pos, reference_points=ref_input, src=src,
                           src_spatial_shapes=src_spatial_shapes, level_start_index=src_level_start_index,
                           src_padding_mask=src_padding_mask, tgt_key_padding_mask=tgt_key_padding_mask, **kwargs)
            if self.bbox_embed is not None:  # iterative bounding-box refinement
                tmp = self.bbox_embed[lid](output)
                if reference_points.shape[-1] == 4:
                    new_ref = (tmp + inverse_sigmoid(reference_points)).sigmoid()
                else:
                    tmp = torch.cat([tmp[..., :2] + inverse_sigmoid(reference_points), tmp[..., 2:]], -1)
                    new_ref = tmp.sigmoid()
                reference_points = new_ref.detach()
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_refs.append(reference_points)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_refs)
        return output, reference_points

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None, tgt_key_padding_mask=None, decoder_outputs=None, **kwargs):
        decoder_outputs = {} if decoder_outputs is None else decoder_outputs
        tgt, query_pos, tgt_key_padding_mask, reference_points = self.pre_process_tgt(
            tgt, query_pos, tgt_key_padding_mask=tgt_key_padding_mask, reference_points=reference_points, **kwargs)
        output, inter_refs = self.decoder_forward(
            tgt=tgt, reference_points=reference_points, src=src, src_spatial_shapes=src_spatial_shapes,
            src_level_start_index=src_level_start_index, src_valid_ratios=src_valid_ratios, query_pos=query_pos,
            src_padding_mask=src_padding_mask, tgt_key_padding_mask=tgt_key_padding_mask, **kwargs)
        decoder_outputs["init_reference_out"] = reference_points
        decoder_outputs.update({"hs": output, "inter_references_out": inter_refs})
        return decoder_outputs


class DeformableTransformer(nn.Module):
    """Transformer with multi-scale deformable attention.  GPU only (the MSDA op has no CPU implementation)."""

    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 encoder=None, decoder=None, decoder_layer=None, encoder_layer=None, dropout=0.1, activation="relu",
                 return_intermediate_dec=False, num_feature_levels=4, dec_n_points=4, enc_n_points=4,
                 two_stage=False, two_stage_num_proposals=300):
        super().__init__()
        if two_stage:
            raise NotImplementedError("two-stage Deformable-DETR is not part of this build (unused by the R50 models)")
        self.d_model, self.nhead = d_model, nhead
        self.two_stage, self.two_stage_num_proposals = two_stage, two_stage_num_proposals
        if encoder is None:
            encoder_layer = encoder_layer or DeformableTransformerEncoderLayer(
                d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, enc_n_points)
            encoder = DeformableTransformerEncoder(encoder_layer, num_encoder_layers)
        self.encoder = encoder
        if decoder is None:
            decoder_layer = decoder_layer or DeformableTransformerDecoderLayer(
                d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, dec_n_points)
            decoder = DeformableTransformerDecoder(decoder_layer, num_decoder_layers, return_intermediate_dec)
        self.decoder = decoder
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.reference_points = nn.Linear(d_model, 2)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        xavier_uniform_(self.reference_points.weight.data, gain=1.0)
        constant_(self.reference_points.bias.data, 0.0)
        normal_(self.level_embed)

    @staticmethod
    def get_valid_ratio(mask):
        """mask (B,H,W) -> (B,2): fraction (w, h) of the map that is not padding."""
        _, H, W = mask.shape
        valid_h = torch.sum((~mask).float()[:, :, 0], 1, keepdim=True)
        valid_w = torch.sum((~mask).float()[:, 0, :], 1, keepdim=True)
        return torch.cat([valid_w / W, valid_h / H], 1)

    def forward(self, srcs, masks, pos_embeds, query_embed=None, **kwargs):
        assert query_embed is not None
        device = srcs[0].device
        src_flatten, mask_flatten, pos_flatten, shapes = [], [], [], []
        pos_encoder = kwargs.pop("pos_encoder", None)  # set by DeformableDETR when it left the encodings to this module
        ready = kwargs.pop("src_flatten", None)        # (B, S, C): the levels of ``srcs`` are already views into it
        ready_mask = kwargs.pop("mask_flatten", None)  # (B, S) bool and (B, L, 2) float32 from alo_mask_pyramid
        ready_ratios = kwargs.pop("valid_ratios", None)
        for lvl, (src, mask, pos_embed) in enumerate(zip(srcs, masks, pos_embeds)):
            _, _, h, w = src.shape
            shapes.append((h, w))
            if ready is None:
                src_flatten.append(src.flatten(2).transpose(1, 2))
            if ready_mask is None:
                mask_flatten.append(mask.flatten(1))
            if pos_embed is not None:
                pos_flatten.append(pos_embed.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1))
        src_flatten = torch.cat(src_flatten, 1) if ready is None else ready
        mask_flatten = torch.cat(mask_flatten, 1) if ready_mask is None else ready_mask
        spatial_shapes, level_start_index = _level_geometry(tuple(shapes), device)
        sizes = [h * w for h, w in shapes]
        if pos_flatten:
            pos_flatten = torch.cat(pos_flatten, 1).to(src_flatten.dtype)
        else:
            # inference: sine encoding of all levels + level embedding in one HIP pass, straight into the flattened layout
            pos_flatten = alo_hip.pos_sine_flat(mask_flatten, spatial_shapes, level_start_index, pos_encoder.dim_t(device),
                                                self.level_embed, pos_encoder.normalize, pos_encoder.center,
                                                pos_encoder.scale, src_flatten.dtype)
        valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1) if ready_ratios is None else ready_ratios

        memory = self.encoder(src_flatten, spatial_shapes, level_start_index, valid_ratios, pos_flatten, mask_flatten,
                              **kwargs)

        bs, _, c = memory.shape
        query_pos, tgt = torch.split(query_embed, c, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        tgt = tgt.unsqueeze(0).expand(bs, -1, -1)
        if _fused_ok(self, kwargs, memory, query_embed):
            # the one-pass kernels of the decoder layers want dense operands: materialise the two broadcasts once per forward
            # instead of once per layer
            query_pos, tgt = query_pos.contiguous(), tgt.contiguous()
        reference_points = self.reference_points(query_pos).sigmoid()

        out = {}
        out.update(self.decoder(tgt, reference_points, memory, spatial_shapes, level_start_index, valid_ratios,
                                query_pos=query_pos, src_padding_mask=mask_flatten, **kwargs))
        memory_t = memory.transpose(1, 2)
        splits, start = [], 0
        for (h, w), n in zip(shapes, sizes):
            splits.append(memory_t[..., start:start + n].reshape(bs, c, h, w))
            start += n
        out["memory"] = splits
        out["enc_outputs_class"] = None
        out["enc_outputs_coord_unact"] = None
        return out
