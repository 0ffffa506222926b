"""DETR (https://arxiv.org/abs/2005.12872): ResNet backbone + vanilla transformer + class / box heads.

Same constructor, state-dict layout, ``forward(frames) -> dict`` and ``inference`` as alonet/detr/detr.py:32-560.
Runs on the CPU (default ``device``) or the GPU; no custom kernel.  TensorRT / tracing mode is not provided.
"""
import torch
import torch.nn.functional as F
from torch import nn

import aloscene
from alonet.common import load_weights
from alonet.transformers import MLP, PositionEmbeddingSine

from .backbone import Backbone
from .misc import assert_and_export_onnx
from .transformer import Transformer, TransformerDecoder, TransformerDecoderLayer

INPUT_MEAN_STD = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))


class Detr(nn.Module):
    INPUT_MEAN_STD = INPUT_MEAN_STD

    def __init__(self, backbone, transformer, num_classes, num_queries, background_class=None, aux_loss=True,
                 weights=None, return_dec_outputs=False, return_enc_outputs=False, return_bb_outputs=False,
                 device=torch.device("cpu"), strict_load_weights=True, tracing=False):
        super().__init__()
        if tracing:
            raise NotImplementedError("tracing / ONNX export mode is not part of this build")
        self.backbone = backbone
        self.num_queries = num_queries
        self.hidden_dim = transformer.d_model
        self.query_embed = nn.Embedding(num_queries, self.hidden_dim)
        self.input_proj = nn.Conv2d(backbone.num_channels, self.hidden_dim, kernel_size=1)
        self.transformer = transformer
        self.num_decoder_layers = transformer.decoder.num_layers
        self.return_dec_outputs = return_dec_outputs
        self.return_enc_outputs = return_enc_outputs
        self.return_bb_outputs = return_bb_outputs
        # softmax classification: one extra "no object" class, by default the last id
        self.background_class = num_classes if background_class is None else background_class
        self.num_classes = num_classes + 1
        self.class_embed = self.build_class_embed()
        self.bbox_embed = self.build_bbox_embed()
        self.aux_loss = aux_loss
        if device is not None:
            self.to(device)
        if weights is not None:
            load_weights(self, weights, device, strict_load_weights=strict_load_weights)
        self.device = device

    @assert_and_export_onnx(check_mean_std=True, input_mean_std=INPUT_MEAN_STD)
    def forward(self, frames, **kwargs):
        """frames: batched, resnet-normalised ``aloscene.Frame`` with a padding mask ->
        dict(pred_logits (B,Q,num_classes+1), pred_boxes (B,Q,4) relative xcyc[, aux_outputs, dec/enc/bb outputs])."""
        features, pos = self.backbone(frames, **kwargs)
        src, mask = features[-1]
        mask = mask[:, 0].to(torch.bool)
        proj = self.input_proj(src)
        out = self.transformer(proj, mask, self.query_embed.weight, pos[-1], **kwargs)
        if self.return_bb_outputs:
            features[-1] = (proj, mask)
        return self.forward_heads(out, bb_outputs=(features, pos))

    def forward_position_heads(self, transformer_outputs):
        return self.bbox_embed(transformer_outputs["hs"]).sigmoid()

    def forward_class_heads(self, transformer_outputs):
        return self.class_embed(transformer_outputs["hs"])

    def forward_heads(self, transformer_outputs, bb_outputs=None, **kwargs):
        outputs_class = self.forward_class_heads(transformer_outputs)
        outputs_coord = self.forward_position_heads(transformer_outputs)
        out = {"pred_logits": outputs_class[-1], "pred_boxes": outputs_coord[-1]}
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b}
                                  for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
        if self.return_dec_outputs:
            out["dec_outputs"] = transformer_outputs["hs"]
        if self.return_enc_outputs:
            out["enc_outputs"] = transformer_outputs["memory"]
        if self.return_bb_outputs:
            features, pos = bb_outputs
            for lvl, (src, mask) in enumerate(features):
                out[f"bb_lvl{lvl}_src_outputs"] = src
                out[f"bb_lvl{lvl}_mask_outputs"] = mask
                out[f"bb_lvl{lvl}_pos_outputs"] = pos[lvl]
        return out

    # ---- post-processing ----------------------------------------------------------------------------------------------
    def get_outs_labels(self, m_outputs):
        scores, labels = F.softmax(m_outputs["pred_logits"], -1).max(-1)
        return labels, scores

    def get_outs_filter(self, outs_scores=None, outs_labels=None, m_outputs=None, background_class=None, threshold=None):
        background_class = background_class or self.background_class
        if outs_scores is None or outs_labels is None:
            outs_labels, outs_scores = self.get_outs_labels(m_outputs)
        filters = []
        for scores, labels in zip(outs_scores, outs_labels):
            keep = labels != background_class
            filters.append(keep if threshold is None else keep & (scores > threshold))
        return filters

    @torch.no_grad()
    def inference(self, forward_out, filters=None, background_class=None, threshold=None):
        """Forward outputs -> one ``aloscene.BoundingBoxes2D`` (relative xcyc, Labels with scores) per image."""
        scores_all, labels_all = F.softmax(forward_out["pred_logits"].float(), -1).max(-1)
        if filters is None:
            filters = self.get_outs_filter(outs_scores=scores_all, outs_labels=labels_all,
                                           background_class=background_class, threshold=threshold)
        preds = []
        for scores, labels, boxes, keep in zip(scores_all, labels_all, forward_out["pred_boxes"], filters):
            lab = aloscene.Labels(labels[keep].type(torch.float32), encoding="id", scores=scores[keep], names=("N",))
            preds.append(aloscene.BoundingBoxes2D(boxes[keep].float().cpu(), boxes_format="xcyc", absolute=False,
                                                  names=("N", None), labels=lab))
        return preds

    # ---- builders -------------------------------------------------------------------------------------------------------
    def build_class_embed(self):
        return nn.Linear(self.hidden_dim, self.num_classes)

    def build_bbox_embed(self):
        return MLP(self.hidden_dim, self.hidden_dim, 4, 3)

    def build_positional_encoding(self, hidden_dim=256, position_embedding="sin", center=False):
        if position_embedding not in ("v2", "sin", "sine"):
            raise NotImplementedError(f"not supported {position_embedding}")
        return PositionEmbeddingSine(hidden_dim // 2, normalize=True, center=center)

    def build_backbone(self, backbone_name, train_backbone, return_interm_layers, dilation):
        return Backbone(backbone_name, train_backbone, return_interm_layers, dilation)

    def build_decoder_layer(self, hidden_dim=256, dropout=0.1, nheads=8, dim_feedforward=2048, normalize_before=False):
        return TransformerDecoderLayer(hidden_dim, nheads, dim_feedforward, dropout, normalize_before=normalize_before)

    def build_decoder(self, hidden_dim=256, num_decoder_layers=6, **layer_kwargs):
        layer = self.build_decoder_layer(hidden_dim=hidden_dim, **layer_kwargs)
        return TransformerDecoder(layer, num_decoder_layers, nn.LayerNorm(hidden_dim), return_intermediate=True)

    def build_transformer(self, hidden_dim=256, dropout=0.1, nheads=8, dim_feedforward=2048, num_encoder_layers=6,
                          num_decoder_layers=6, normalize_before=False):
        decoder = self.build_decoder(hidden_dim=hidden_dim, num_decoder_layers=num_decoder_layers, dropout=dropout,
                                     nheads=nheads, dim_feedforward=dim_feedforward, normalize_before=normalize_before)
        return Transformer(d_model=hidden_dim, dropout=dropout, nhead=nheads, dim_feedforward=dim_feedforward,
                           num_encoder_layers=num_encoder_layers, num_decoder_layers=num_decoder_layers,
                           normalize_before=normalize_before, return_intermediate_dec=True, decoder=decoder)
