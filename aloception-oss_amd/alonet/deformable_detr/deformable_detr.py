"""Deformable-DETR object detector (https://arxiv.org/abs/2010.04159) on the gfx950 multi-scale deformable attention.

Same constructor arguments, state-dict layout, ``forward(frames) -> dict`` and ``inference(forward_out) -> [BoundingBoxes2D]``
as the reference (alonet/deformable_detr/deformable_detr.py:32-760).  Backbone convolutions, linear layers and
``nn.MultiheadAttention`` run on stock PyTorch-ROCm; the 12 deformable-attention gathers per forward run on the HIP op.
TensorRT / TorchScript tracing mode is not provided.
"""
import copy
import math

import threading

import torch
import torch.nn.functional as F
from torch import nn

import alo_hip
import aloscene
from alonet.common import load_weights
from alonet.detr.backbone import _resize_mask, conv1x1_as_gemm
from alonet.detr.misc import assert_and_export_onnx
from alonet.transformers import MLP, PositionEmbeddingSine

from .backbone import Backbone
from .deformable_transformer import (
    DeformableTransformer,
    DeformableTransformerDecoder,
    DeformableTransformerDecoderLayer,
)
from .utils import inverse_sigmoid

INPUT_MEAN_STD = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class DeformableDETR(nn.Module):
    """
    Parameters
    ----------
    backbone : nn.Module            ``Joiner(Backbone, PositionEmbeddingSine)``
    transformer : nn.Module         ``DeformableTransformer``
    num_classes : int
    num_queries : int               detection slots (300)
    num_feature_levels : int        pyramid levels sampled by the attention (4)
    aux_loss : bool                 also return the predictions of the intermediate decoder layers
    with_box_refine : bool          iterative bounding-box refinement
    weights : str                   checkpoint path or registered name
    device : torch.device           defaults to cuda (the deformable attention op has no CPU implementation)
    activation_fn : "sigmoid" | "softmax"   (softmax adds a background class)
    """

    INPUT_MEAN_STD = INPUT_MEAN_STD

    def __init__(self, backbone, transformer, num_classes, num_queries=300, num_feature_levels=4, aux_loss=True,
                 with_box_refine=False, return_dec_outputs=False, return_enc_outputs=False, return_bb_outputs=False,
                 weights=None, device=torch.device("cuda"), activation_fn="sigmoid", return_intermediate_dec=True,
                 strict_load_weights=True, tracing=False, include_preprocessing=False):
        super().__init__()
        if tracing:
            raise NotImplementedError("tracing / ONNX export mode is not part of this build")
        if activation_fn not in ("sigmoid", "softmax"):
            raise Exception(f"activation_fn = {activation_fn} must be one of this two values: 'sigmoid' or 'softmax'.")
        self.num_feature_levels = num_feature_levels
        self.backbone = backbone
        self.num_queries = num_queries
        self.return_intermediate_dec = return_intermediate_dec
        self.hidden_dim = hidden = transformer.d_model
        self.return_dec_outputs = return_dec_outputs
        self.return_enc_outputs = return_enc_outputs
        self.return_bb_outputs = return_bb_outputs
        self.activation_fn = activation_fn
        self.background_class = num_classes if activation_fn == "softmax" else None
        num_classes += 1 if activation_fn == "softmax" else 0

        def proj(in_ch, **conv):
            return nn.Sequential(nn.Conv2d(in_ch, hidden, **conv), nn.GroupNorm(32, hidden))

        if num_feature_levels > 1:
            n_backbone = len(backbone.strides) - 1  # stride-4 stage is not used by the detector
            projs = [proj(backbone.num_channels[i], kernel_size=1) for i in range(1, n_backbone + 1)]
            in_ch = backbone.num_channels[n_backbone]
            for _ in range(num_feature_levels - n_backbone):
                projs.append(proj(in_ch, kernel_size=3, stride=2, padding=1))
                in_ch = hidden
            self.input_proj = nn.ModuleList(projs)
        else:
            self.input_proj = nn.ModuleList([proj(backbone.num_channels[0], kernel_size=1)])
        self.query_embed = nn.Embedding(num_queries, hidden * 2)
        self.transformer = transformer
        self.class_embed = nn.Linear(hidden, num_classes)
        self.bbox_embed = MLP(hidden, hidden, 4, 3)
        self.aux_loss = aux_loss
        self.with_box_refine = with_box_refine

        prior_prob = 0.01
        self.class_embed.bias.data = torch.ones(num_classes) * -math.log((1 - prior_prob) / prior_prob)
        nn.init.constant_(self.bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(self.bbox_embed.layers[-1].bias.data, 0)
        for p in self.input_proj:
            nn.init.xavier_uniform_(p[0].weight, gain=1)
            nn.init.constant_(p[0].bias, 0)

        self.num_decoder_layers = num_pred = transformer.decoder.num_layers
        if with_box_refine:
            self.class_embed = _get_clones(self.class_embed, num_pred)
            self.bbox_embed = _get_clones(self.bbox_embed, num_pred)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            self.transformer.decoder.bbox_embed = self.bbox_embed
        else:
            nn.init.constant_(self.bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([self.class_embed for _ in range(num_pred)])  # one shared head
            self.bbox_embed = nn.ModuleList([self.bbox_embed for _ in range(num_pred)])
            self.transformer.decoder.bbox_embed = None

        self.device = device
        if device is not None:
            self.to(device)
        if weights is not None:
            load_weights(self, weights, device, strict_load_weights=strict_load_weights)

    # ---- forward ----------------------------------------------------------------------------------------------------
    @assert_and_export_onnx(check_mean_std=True, input_mean_std=INPUT_MEAN_STD)
    def forward(self, frames, **kwargs):
        """frames: batched ``aloscene.Frame`` (B,3,H,W), resnet-normalised, with ``frames.mask`` (1 on padding).

        Returns a dict: ``pred_logits`` (B, num_queries, num_classes), ``pred_boxes`` (B, num_queries, 4) as relative
        (xc, yc, w, h), ``activation_fn``, and optionally ``aux_outputs`` / ``dec_outputs`` / ``enc_outputs`` /
        ``bb_lvl*_{src,mask,pos}_outputs``.
        """
        if "is_tracing" not in kwargs:  # the pure-torch export branch may run anywhere; the HIP op may not
            assert next(self.parameters()).is_cuda, "DeformableDETR cannot run on CPU (due to MSdeformable op)"
        frame_masks = frames.mask.as_tensor()
        # the stride-4 positional encoding is only ever an output (bb_lvl0_pos_outputs), never an input of the transformer;
        # at inference the others are produced by the transformer itself, flattened, in one pass (alo_pos_sine_flat)
        # PanopticHead switches return_bb_outputs on for the backbone FEATURES (and the last level's mask); unless it hands the
        # detector's outputs on to its caller (return_detr_outputs) nobody reads the per-level positional encodings / other masks,
        # and the detector may keep its inference fast path (`_alo_lean_bb_outputs`, set by PanopticHead)
        lean = getattr(self, "_alo_lean_bb_outputs", False)
        lazy_pos = ((not self.return_bb_outputs or lean) and "is_tracing" not in kwargs and not self.training
                    and isinstance(self.backbone[1], PositionEmbeddingSine) and self.backbone[1].num_pos_feats % 4 == 0
                    and alo_hip.fusable(frames.as_tensor(), self.transformer.level_embed))
        skip = tuple(range(len(self.backbone.num_channels))) if lazy_pos else (() if self.return_bb_outputs else (0,))
        # inference: the per-level padding masks come from one kernel below, not from one F.interpolate per backbone stage
        features, pos = self.backbone(frames, skip_pos_levels=skip, **(dict(kwargs, skip_masks=True) if lazy_pos else kwargs))

        srcs, masks = [], []
        flat = self._flat_sources(features) if lazy_pos else None
        if lazy_pos and flat is None:   # rare shapes: the stock mask resize after all
            features = [(x, _resize_mask(frame_masks.float(), x.shape[-2:]).to(torch.bool)) for x, _ in features]
        if flat is not None:
            # every level's projection is normalised straight into its slot of the encoder's flattened source; masks of all
            # levels (bilinear resize for the backbone stages, nearest for the extra level, as the stock path) + valid ratios: one call
            flat, slots = flat
            n_bb = len(features) - 1
            shapes = [hw for _, hw in slots]
            mask_flat, valid_ratios = alo_hip.mask_pyramid(frame_masks, shapes, nearest_levels=range(n_bb, len(slots)))
            for lvl, (start, (h, w)) in enumerate(slots):
                x = features[min(lvl + 1, n_bb)][0]
                rows = flat[:, start:start + h * w]
                self._project(lvl, x, out_rows=rows)
                srcs.append(rows.view(rows.shape[0], h, w, -1).permute(0, 3, 1, 2))
                if lvl >= n_bb:
                    pos.append(None)
                masks.append(mask_flat[:, start:start + h * w].view(-1, h, w))
            kwargs = dict(kwargs, src_flatten=flat, mask_flatten=mask_flat, valid_ratios=valid_ratios)
        else:
            for lvl, (src, mask) in enumerate(features[1:]):
                srcs.append(self._project(lvl, src))
                masks.append(mask[:, 0])
            for lvl in range(len(srcs), self.num_feature_levels):  # extra, coarser levels
                src = self._project(lvl, features[-1][0] if lvl == len(features) - 1 else srcs[-1])
                mask = F.interpolate(frame_masks.float(), size=src.shape[-2:]).to(torch.bool)
                pos.append(None if lazy_pos else self.backbone[1]((src, mask)).to(src.dtype))
                srcs.append(src)
                masks.append(mask[:, 0])
        if lazy_pos:
            kwargs = dict(kwargs, pos_encoder=self.backbone[1])

        transformer_out = self.transformer(srcs, masks, pos[1:], self.query_embed.weight, **kwargs)
        if self.return_bb_outputs:
            features[-1] = (srcs[-2], masks[-2])
        return self.forward_heads(transformer_out, bb_outputs=(features, pos[:-1]))

    def _project_fast(self, lvl, x):
        """Which fast path covers ``input_proj[lvl]`` on ``x``: "gemm" (1x1), "conv3x3" or None."""
        conv = self.input_proj[lvl][0]
        if (alo_hip.fusable(x, conv.weight) and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
                and conv.groups == 1 and conv.dilation == (1, 1) and conv.padding_mode == "zeros"):
            if conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.out_channels % 4 == 0:
                return "gemm"
            if conv.kernel_size == (3, 3) and alo_hip.conv3x3_supported(x, conv.weight, conv.stride, conv.padding):
                return "conv3x3"
        return None

    def _project(self, lvl, x, out_rows=None):
        """``input_proj[lvl](x)`` (convolution + GroupNorm).  At inference on channels-last bf16 maps the 1x1 projections are
        GEMMs over the NHWC rows and the 3x3 / stride 2 one is the implicit-GEMM kernel (MIOpen's choices for these shapes are
        3-5x slower); with ``out_rows`` (B, h*w, C) the GroupNorm writes its channels-last rows there (alo_groupnorm_rows)."""
        conv, norm = self.input_proj[lvl][0], self.input_proj[lvl][1]
        kind = self._project_fast(lvl, x)
        if kind is None:
            assert out_rows is None
            return self.input_proj[lvl](x)
        if kind == "gemm":
            y = conv1x1_as_gemm(x, conv.weight, conv.bias, conv.stride)
        else:
            y = alo_hip.conv3x3(x, conv.weight, conv.bias, False, conv.stride)
        if out_rows is None:
            return norm(y)
        n, c, h, w = y.shape
        alo_hip.groupnorm_rows(y.permute(0, 2, 3, 1).reshape(n, h * w, c), norm.weight, norm.bias, norm.num_groups, norm.eps,
                               out=out_rows)
        return out_rows

    def _flat_sources(self, features):
        """(flat (B, S, C) buffer, [(start, (h, w)) per level]) when every level can be projected + normalised straight into the
        encoder's flattened source, else None."""
        n_bb = len(features) - 1
        if self.num_feature_levels > n_bb + 1:   # a second extra level would read the first one out of the flat buffer
            return None
        slots, start = [], 0
        for lvl in range(self.num_feature_levels):
            x = features[min(lvl + 1, n_bb)][0]
            conv, norm = self.input_proj[lvl][0], self.input_proj[lvl][1]
            if self._project_fast(lvl, x) is None or not isinstance(norm, nn.GroupNorm) or not norm.affine:
                return None
            h = (x.shape[2] + 2 * conv.padding[0] - conv.kernel_size[0]) // conv.stride[0] + 1
            w = (x.shape[3] + 2 * conv.padding[1] - conv.kernel_size[1]) // conv.stride[1] + 1
            probe = x.new_empty((1, 1, conv.out_channels))
            if not alo_hip.groupnorm_rows_supported(probe, norm.weight, norm.num_groups):
                return None
            slots.append((start, (h, w)))
            start += h * w
        x0 = features[1][0]
        return x0.new_empty((x0.shape[0], start, self.input_proj[0][0].out_channels)), slots

    @staticmethod
    def _shared(heads):
        """True when every decoder level uses the SAME head module (no box refinement): the levels can go through it at once."""
        return all(m is heads[0] for m in heads)

    def forward_position_heads(self, transformer_outputs):
        hs = transformer_outputs["hs"]
        init_ref, inter_refs = transformer_outputs["init_reference_out"], transformer_outputs["inter_references_out"]

        def boxes(tmp, reference):
            if reference.shape[-1] == 4:
                tmp = tmp + reference
            else:
                assert reference.shape[-1] == 2
                tmp = torch.cat([tmp[..., :2] + reference, tmp[..., 2:]], -1)
            return tmp.sigmoid()

        if self._shared(self.bbox_embed) and hs.dim() == 4 and inter_refs.dim() == 4 and hs.shape[0] > 1:
            # one pass over all levels instead of ~14 small kernels per level (level l refines reference l - 1)
            refs = torch.cat([init_ref.unsqueeze(0).to(inter_refs.dtype), inter_refs[:hs.shape[0] - 1]], 0)
            return list(boxes(self.bbox_embed[0](hs), inverse_sigmoid(refs)).unbind(0))
        return [boxes(self.bbox_embed[lvl](hs[lvl]), inverse_sigmoid(init_ref if lvl == 0 else inter_refs[lvl - 1]))
                for lvl in range(hs.shape[0])]

    def forward_class_heads(self, transformer_outputs):
        hs = transformer_outputs["hs"]
        if self._shared(self.class_embed) and hs.dim() == 4:
            return self.class_embed[0](hs)
        return torch.stack([self.class_embed[lvl](hs[lvl]) for lvl in range(hs.shape[0])])

    def forward_heads(self, transformer_outputs, bb_outputs=None, **kwargs):
        outputs_class = self.forward_class_heads(transformer_outputs)
        outputs_coord = self.forward_position_heads(transformer_outputs)
        last = self.num_decoder_layers - 1 if transformer_outputs["hs"].shape[0] > 1 else 0
        out = {"pred_logits": outputs_class[last], "pred_boxes": outputs_coord[last],
               "activation_fn": self.activation_fn}
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b, "activation_fn": self.activation_fn}
                                  for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
        if self.return_dec_outputs:
            out["dec_outputs"] = transformer_outputs["hs"]
        if self.return_enc_outputs:
            out["enc_outputs"] = transformer_outputs["memory"][-2]
        if self.return_bb_outputs:
            features, pos = bb_outputs
            for lvl, (src, mask) in enumerate(features):
                out[f"bb_lvl{lvl}_src_outputs"] = src
                out[f"bb_lvl{lvl}_mask_outputs"] = mask
                out[f"bb_lvl{lvl}_pos_outputs"] = pos[lvl]
        if not torch.is_grad_enabled() and out["pred_logits"].is_cuda and not out["pred_logits"].is_inference():
            # inference(): scores, labels and boxes of the last level in ONE fp32 tensor, made while the forward's launches are still
            # in flight (inside the HIP graph when the forward is replayed) — inference() then needs a single device-to-host copy
            # instead of six small launches and four copies behind the forward (0.24 -> 0.1 ms of an 9.2 ms step).  The pack rides
            # on the ``pred_logits`` tensor OBJECT: the output dictionary has exactly the reference's keys.  (Inference tensors —
            # ``torch.inference_mode()`` — carry no version counter to tie the pack to; they take inference()'s step-by-step chain.)
            self._pack_detections(out["pred_logits"], out["pred_boxes"], self.activation_fn)
        return out

    @staticmethod
    def _pack_detections(logits, boxes, activation_fn):
        """Attach ``logits._alo_detections = (packed (B, Q, 6) fp32 = [score, label, cx, cy, w, h], boxes, version counters of both,
        activation)``: exactly the values inference() derives (reference deformable_detr.py:508-530: softmax / sigmoid, max over
        the classes)."""
        probs = F.softmax(logits.float(), -1) if activation_fn == "softmax" else logits.float().sigmoid()
        scores, labels = probs.max(-1)
        packed = torch.cat([scores.unsqueeze(-1), labels.unsqueeze(-1).to(torch.float32), boxes.float()], -1)
        logits._alo_detections = (packed, boxes, logits._version, boxes._version, activation_fn)

    @staticmethod
    def _packed_detections(logits, boxes, activation_fn):
        """The pack the forward left on ``logits`` if it still describes THESE tensors (same objects, untouched since), else None."""
        ready = getattr(logits, "_alo_detections", None)
        if ready is None or logits.is_inference() or boxes.is_inference():
            return None
        packed, of_boxes, v_logits, v_boxes, of_activation = ready
        if of_boxes is boxes and v_logits == logits._version and v_boxes == boxes._version and of_activation == activation_fn:
            return packed
        return None

    # ---- post-processing ----------------------------------------------------------------------------------------------
    def _to_host_pinned(self, packed):
        """The step's one device-to-host hand-over (B x Q x 6 floats) through a page-locked buffer kept with the model: the copy is
        a DMA straight into it (a pageable ``.cpu()`` stages through the runtime's own bounce buffer under a process-wide lock —
        with eight ranks on one host at 9 ms per step that is the first place weak scaling is lost).  One buffer per host THREAD
        (two threads calling inference() on one model must not overwrite each other's detections: round-5 advisor finding), and what
        is returned is a copy of it (B x Q x 6 floats: nothing next to the DMA), so a ``get_outs_filter`` override that keeps views of
        its input never aliases the next call's detections."""
        if not packed.is_cuda:
            return packed
        slot = self.__dict__.setdefault("_alo_host_detections", threading.local())
        buf = getattr(slot, "buf", None)
        if buf is None or buf.shape != packed.shape or buf.dtype != packed.dtype:
            buf = slot.buf = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
        buf.copy_(packed, non_blocking=True)
        torch.cuda.current_stream(packed.device).synchronize()
        return buf.clone()

    def get_outs_labels(self, m_outputs=None, activation_fn=None):
        assert m_outputs is not None
        activation_fn = m_outputs.get("activation_fn") or activation_fn or self.activation_fn
        logits = m_outputs["pred_logits"]
        probs = F.softmax(logits, -1) if activation_fn == "softmax" else logits.sigmoid()
        scores, labels = probs.max(-1)
        return labels, scores

    def get_outs_filter(self, outs_scores=None, outs_labels=None, m_outputs=None, threshold=None, activation_fn=None):
        activation_fn = activation_fn or self.activation_fn
        if outs_scores is None or outs_labels is None:
            outs_labels, outs_scores = self.get_outs_labels(m_outputs, activation_fn=activation_fn)
        if torch.is_tensor(outs_scores) and torch.is_tensor(outs_labels):  # one comparison for the batch, not one per image
            if activation_fn == "softmax":
                keep = outs_labels != self.background_class
                keep = keep if threshold is None else keep & (outs_scores > threshold)
            else:
                keep = outs_scores > (0.2 if threshold is None else threshold)
            return list(keep.unbind(0))
        filters = []
        for scores, labels in zip(outs_scores, outs_labels):
            if activation_fn == "softmax":
                keep = labels != self.background_class
                filters.append(keep if threshold is None else keep & (scores > threshold))
            else:
                filters.append(scores > (0.2 if threshold is None else threshold))
        return filters

    @torch.no_grad()
    def inference(self, forward_out, threshold=0.2, filters=None, **kwargs):
        """Forward outputs -> one ``aloscene.BoundingBoxes2D`` (relative xcyc, with ``Labels`` + scores) per image."""
        logits, boxes_all = forward_out["pred_logits"], forward_out["pred_boxes"]
        activation_fn = forward_out.get("activation_fn") or self.activation_fn
        packed = self._packed_detections(logits, boxes_all, activation_fn)
        if packed is not None:
            # the forward already packed (score, label, box) of THESE tensors: one copy to the host, one selection for the whole
            # batch, then views per image
            host = self._to_host_pinned(packed)
            if filters is None:
                filters = self.get_outs_filter(outs_scores=host[..., 0], outs_labels=host[..., 1].long(), threshold=threshold,
                                               activation_fn=activation_fn, **kwargs)
            keep_all = (filters if torch.is_tensor(filters) else torch.stack(list(filters))).cpu()   # host already unless the caller's
            counts = keep_all.sum(1).tolist()
            sel = host[keep_all]                                                                        # (kept, 6)
            scores, labels, boxes = sel[:, 0].contiguous(), sel[:, 1].contiguous(), sel[:, 2:].contiguous()
            return [aloscene.BoundingBoxes2D(b, boxes_format="xcyc", absolute=False, names=("N", None),
                                             labels=aloscene.Labels(lab, encoding="id", scores=sc, names=("N",)))
                    for sc, lab, b in zip(scores.split(counts), labels.split(counts), boxes.split(counts))]
        probs = F.softmax(logits.float(), -1) if activation_fn == "softmax" else logits.float().sigmoid()
        scores_all, labels_all = probs.max(-1)
        if filters is None:
            filters = self.get_outs_filter(outs_scores=scores_all, outs_labels=labels_all, threshold=threshold,
                                           activation_fn=activation_fn, **kwargs)
        # one device -> host transfer per tensor; the per-image boolean selection then runs on the host (on the device every
        # `x[keep]` is a nonzero + a synchronisation: 3 per image)
        keep_all = torch.stack(list(filters)) if not torch.is_tensor(filters) else filters
        scores_all, labels_all, boxes_all, keep_all = (t.cpu() for t in (scores_all, labels_all, boxes_all.float(), keep_all))
        preds = []
        for scores, labels, boxes, keep in zip(scores_all, labels_all, boxes_all, keep_all):
            lab = aloscene.Labels(labels[keep].type(torch.float32), encoding="id", scores=scores[keep], names=("N",))
            preds.append(aloscene.BoundingBoxes2D(boxes[keep], boxes_format="xcyc", absolute=False,
                                                  names=("N", None), labels=lab))
        return preds

    # ---- builders (same names as the reference so subclasses can override them) ----------------------------------------
    def build_positional_encoding(self, hidden_dim=256):
        return PositionEmbeddingSine(hidden_dim // 2, normalize=True, center=True)

    def build_backbone(self, backbone_name="resnet50", train_backbone=True, return_interm_layers=True, dilation=False):
        return Backbone(backbone_name, train_backbone, return_interm_layers, dilation)

    def build_decoder_layer(self, hidden_dim=256, dropout=0.1, nheads=8, dim_feedforward=1024, num_feature_levels=4,
                            dec_n_points=4):
        return DeformableTransformerDecoderLayer(d_model=hidden_dim, dim_feedforward=dim_feedforward, dropout=dropout,
                                                 activation="relu", n_levels=num_feature_levels, n_heads=nheads,
                                                 n_points=dec_n_points)

    def build_decoder(self, dec_layers=6, return_intermediate_dec=True, hidden_dim=256, num_feature_levels=4):
        layer = self.build_decoder_layer(hidden_dim=hidden_dim, num_feature_levels=num_feature_levels)
        return DeformableTransformerDecoder(layer, dec_layers, return_intermediate_dec)

    def build_transformer(self, hidden_dim=256, dropout=0.1, nheads=8, dim_feedforward=1024, enc_layers=6, dec_layers=6,
                          num_feature_levels=4, dec_n_points=4, enc_n_points=4, return_intermediate_dec=True):
        decoder = self.build_decoder(dec_layers=dec_layers, return_intermediate_dec=return_intermediate_dec,
                                     hidden_dim=hidden_dim, num_feature_levels=num_feature_levels)
        return DeformableTransformer(decoder=decoder, d_model=hidden_dim, dropout=dropout, nhead=nheads,
                                     dim_feedforward=dim_feedforward, num_encoder_layers=enc_layers,
                                     num_decoder_layers=dec_layers, num_feature_levels=num_feature_levels,
                                     dec_n_points=dec_n_points, enc_n_points=enc_n_points,
                                     return_intermediate_dec=return_intermediate_dec)
