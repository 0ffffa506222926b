"""DETR transformer: post-/pre-norm encoder-decoder on ``nn.MultiheadAttention`` with positional encodings added to
queries and keys at every layer; the decoder returns the (normalised) activations of every layer.

Module tree and parameter names follow alonet/detr/transformer.py:22-440 (``encoder.layers.N.{self_attn, linear1,
linear2, norm1, norm2}``, ``decoder.layers.N.{self_attn, multihead_attn, linear1, linear2, norm1..3}``, ``decoder.norm``)
so reference checkpoints load unchanged.  No custom kernel is involved: this is the CPU-runnable plumbing model
(BASELINE configs[0]).
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _get_activation_fn(activation):
    fns = {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}
    if activation not in fns:
        raise RuntimeError(f"activation should be relu/gelu, not {activation}.")
    return fns[activation]


def _with_pos(tensor, pos):
    return tensor if pos is None else tensor + pos


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before

    def with_pos_embed(self, tensor, pos):
        return _with_pos(tensor, pos)

    def _ffn(self, x):
        return self.linear2(self.dropout(self.activation(self.linear1(x))))

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos=None, **kwargs):
        if self.normalize_before:
            src2 = self.norm1(src)
            q = k = _with_pos(src2, pos)
            src = src + self.dropout1(self.self_attn(q, k, value=src2, attn_mask=src_mask,
                                                     key_padding_mask=src_key_padding_mask)[0])
            return src + self.dropout2(self._ffn(self.norm2(src)))
        q = k = _with_pos(src, pos)
        src = self.norm1(src + self.dropout1(self.self_attn(q, k, value=src, attn_mask=src_mask,
                                                            key_padding_mask=src_key_padding_mask)[0]))
        return self.norm2(src + self.dropout2(self._ffn(src)))


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask=None, src_key_padding_mask=None, pos=None, **kwargs):
        out = src
        for layer in self.layers:
            out = layer(out, src_mask=mask, src_key_padding_mask=src_key_padding_mask, pos=pos, **kwargs)
        return out if self.norm is None else self.norm(out)


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, n_heads=8, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before

    def with_pos_embed(self, tensor, pos):
        return _with_pos(tensor, pos)

    def pre_process_tgt(self, tgt, query_pos, tgt_key_padding_mask, **kwargs):
        return tgt, query_pos, tgt_key_padding_mask

    def _ffn(self, x):
        return self.linear2(self.dropout(self.activation(self.linear1(x))))

    def decoder_layer_forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                              memory_key_padding_mask=None, pos=None, query_pos=None, **kwargs):
        def self_block(x):
            q = k = _with_pos(x, query_pos)
            return self.self_attn(q, k, value=x, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]

        def cross_block(x):
            return self.multihead_attn(query=_with_pos(x, query_pos), key=_with_pos(memory, pos), value=memory,
                                       attn_mask=memory_mask, key_padding_mask=memory_key_padding_mask)[0]

        if self.normalize_before:
            tgt = tgt + self.dropout1(self_block(self.norm1(tgt)))
            tgt = tgt + self.dropout2(cross_block(self.norm2(tgt)))
            return tgt + self.dropout3(self._ffn(self.norm3(tgt)))
        tgt = self.norm1(tgt + self.dropout1(self_block(tgt)))
        tgt = self.norm2(tgt + self.dropout2(cross_block(tgt)))
        return self.norm3(tgt + self.dropout3(self._ffn(tgt)))

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None, **kwargs):
        tgt, query_pos, tgt_key_padding_mask = self.pre_process_tgt(tgt, query_pos, tgt_key_padding_mask, **kwargs)
        return self.decoder_layer_forward(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask,
                                          memory_key_padding_mask, pos, query_pos, **kwargs)


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def pre_process_tgt(self, tgt, query_pos, tgt_key_padding_mask, **kwargs):
        return tgt, query_pos, tgt_key_padding_mask

    def decoder_forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                        memory_key_padding_mask=None, pos=None, query_pos=None, **kwargs):
        out = tgt
        intermediate = []
        for layer in self.layers:
            out = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                        tgt_key_padding_mask=tgt_key_padding_mask, memory_key_padding_mask=memory_key_padding_mask,
                        pos=pos, query_pos=query_pos, **kwargs)
            if self.return_intermediate:
                intermediate.append(self.norm(out))
        if self.norm is not None:
            out = self.norm(out)
        if self.return_intermediate:
            return torch.stack(intermediate)
        return out.unsqueeze(0)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None, decoder_outputs=None, **kwargs):
        decoder_outputs = {} if decoder_outputs is None else decoder_outputs
        tgt, query_pos, tgt_key_padding_mask = self.pre_process_tgt(tgt, query_pos, tgt_key_padding_mask=None, **kwargs)
        decoder_outputs["hs"] = self.decoder_forward(
            tgt, memory, tgt_mask=tgt_mask, memory_mask=memory_mask, tgt_key_padding_mask=tgt_key_padding_mask,
            memory_key_padding_mask=memory_key_padding_mask, pos=pos, query_pos=query_pos, **kwargs)
        return decoder_outputs


class Transformer(nn.Module):
    def __init__(self, d_model=512, nhead=8, encoder=None, decoder=None, decoder_layer=None, encoder_layer=None,
                 num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        if encoder is None:
            encoder_layer = encoder_layer or TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout,
                                                                     activation, normalize_before)
            encoder = TransformerEncoder(encoder_layer, num_encoder_layers,
                                         nn.LayerNorm(d_model) if normalize_before else None)
        self.encoder = encoder
        if decoder is None:
            decoder_layer = decoder_layer or TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout,
                                                                     activation, normalize_before)
            decoder = TransformerDecoder(decoder_layer, num_decoder_layers, nn.LayerNorm(d_model),
                                         return_intermediate=return_intermediate_dec)
        self.decoder = decoder
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model = d_model
        self.nhead = nhead

    def forward(self, src, mask, query_embed, pos_embed, **kwargs):
        """src (B,C,H,W), mask (B,H,W) bool, query_embed (Q,C), pos_embed (B,C,H,W) -> dict(hs (layers,B,Q,C), memory)."""
        bs, c, h, w = src.shape
        src = src.flatten(2).permute(2, 0, 1)
        pos_embed = pos_embed.flatten(2).permute(2, 0, 1)
        query_embed = query_embed.unsqueeze(1).repeat(1, bs, 1)
        mask = mask.flatten(1)
        memory = self.encoder(src, src_key_padding_mask=mask, pos=pos_embed, **kwargs)
        out = {}
        out.update(self.decoder(torch.zeros_like(query_embed), memory, memory_key_padding_mask=mask, pos=pos_embed,
                                query_pos=query_embed, **kwargs))
        out["memory"] = memory.permute(1, 2, 0).view(bs, c, h, w)
        out["hs"] = out["hs"].transpose(1, 2)
        return out
