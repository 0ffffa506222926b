"""PanopticHead: per-query segmentation masks on top of a DETR-family detector (DETR or Deformable-DETR).

Reference: alonet/detr_panoptic/detr_panoptic.py:21-311.  ``forward`` runs the detector with its decoder / encoder /
backbone outputs exposed, keeps the confident queries, builds their multi-head attention maps over the stride-32
encoder memory (``MHAttentionMap``) and decodes them to mask logits at stride 4 (``FPNstyleCNN``); ``inference`` turns
them into ``aloscene.Mask`` objects aligned with the predicted boxes.  State-dict keys: ``detr.*``, ``bbox_attention.*``,
``mask_head.*``.  With Deformable-DETR as the detector every attention gather inside ``detr`` is the HIP kernel.
"""
import torch
import torch.nn.functional as F
from torch import nn

import alo_hip
import aloscene
from alonet.common import load_weights
from alonet.detr.misc import assert_and_export_onnx

from .nn import FPNstyleCNN, MHAttentionMap
from .utils import get_mask_queries

INPUT_MEAN_STD = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))


class PanopticHead(nn.Module):
    INPUT_MEAN_STD = INPUT_MEAN_STD

    def __init__(self, DETR_module, freeze_detr=True, aux_loss=None, return_pred_outputs=True,
                 return_detr_outputs=False, device=torch.device("cpu"), weights=None, fpn_list=(1024, 512, 256),
                 strict_load_weights=True, tracing=False):
        super().__init__()
        if tracing:
            raise NotImplementedError("tracing / ONNX export mode is not part of this build")
        self.detr = DETR_module
        self.detr.return_dec_outputs = True
        self.detr.return_enc_outputs = True
        self.detr.return_bb_outputs = True
        self.return_detr_outputs = return_detr_outputs
        self.return_pred_outputs = return_pred_outputs
        self.detr.aux_loss = aux_loss if aux_loss is not None else self.detr.aux_loss
        if freeze_detr:
            for p in self.parameters():
                p.requires_grad_(False)
        hidden_dim, nheads = DETR_module.transformer.d_model, DETR_module.transformer.nhead
        self.bbox_attention = MHAttentionMap(hidden_dim, hidden_dim, nheads, dropout=0.1)
        self.mask_head = FPNstyleCNN(hidden_dim + nheads, list(fpn_list), hidden_dim)
        if device is not None:
            self.to(device)
        self.device = device
        if weights is not None:
            load_weights(self, weights, device, strict_load_weights=strict_load_weights)

    @assert_and_export_onnx(check_mean_std=True, input_mean_std=INPUT_MEAN_STD)
    def forward(self, frames, get_filter_fn=None, **kwargs):
        """-> dict(pred_masks (B, kept queries, H/4, W/4) logits, pred_masks_info[, pred_logits, pred_boxes, detector outputs])."""
        detr_out = self.detr_forward(frames, **kwargs)
        proj_src, mask = detr_out["bb_lvl3_src_outputs"], detr_out["bb_lvl3_mask_outputs"]
        bs = proj_src.shape[0]
        get_filter_fn = get_filter_fn or get_mask_queries
        dec_outputs, filters = get_filter_fn(frames=frames, m_outputs=detr_out, model=self.detr, **kwargs)
        bbox_mask = self.bbox_attention(dec_outputs, detr_out["enc_outputs"], mask=mask)
        fpns = [detr_out[f"bb_lvl{i}_src_outputs"] for i in (2, 1, 0)]
        seg = self.mask_head(proj_src, bbox_mask.to(proj_src.dtype), fpns)
        seg = seg.view(bs, bbox_mask.shape[1], seg.shape[-2], seg.shape[-1])
        out = self.forward_head(seg, detr_outputs=detr_out)
        out["pred_masks_info"] = {"frame_size": tuple(frames.shape[-2:]), "filters": filters}
        return out

    def forward_head(self, pred_masks, detr_outputs, **kwargs):
        out = {"pred_masks": pred_masks}
        if self.return_pred_outputs:
            out.update({"pred_logits": detr_outputs["pred_logits"], "pred_boxes": detr_outputs["pred_boxes"]})
        if self.return_detr_outputs:
            out.update({k: v for k, v in detr_outputs.items() if k not in ("pred_logits", "pred_boxes")})
        if "activation_fn" in detr_outputs:
            out.setdefault("activation_fn", detr_outputs["activation_fn"])
        return out

    def detr_forward(self, frames, **kwargs):
        # The head reads the backbone features, the last level's mask, the decoder and encoder outputs — not the per-level positional
        # encodings or the other levels' masks.  Unless the detector's outputs are handed on to the caller (return_detr_outputs) a
        # Deformable-DETR detector may therefore keep its fused inference path for THIS call (deformable_detr.py forward).
        self.detr._alo_lean_bb_outputs = not self.return_detr_outputs
        try:
            return self.detr(frames, **kwargs)
        finally:
            self.detr._alo_lean_bb_outputs = False

    @torch.no_grad()
    def inference(self, forward_out, maskth=0.5, filters=None, frame_size=None, **kwargs):
        """-> (list of BoundingBoxes2D, list of aloscene.Mask (N,H,W) one-hot over the kept queries, per image)."""
        info = forward_out.get("pred_masks_info")
        b_filters = filters or self.detr.get_outs_filter(m_outputs=forward_out, **kwargs)
        m_filters = filters or (info.get("filters") if isinstance(info, dict) else None)
        if m_filters is None:
            if forward_out["pred_masks"].shape[1] == forward_out["pred_boxes"].shape[1]:
                b, nq = forward_out["pred_boxes"].shape[:2]
                m_filters = [forward_out["pred_boxes"].new_ones(nq).bool()] * b
            else:
                m_filters = b_filters
        frame_size = frame_size or (info.get("frame_size") if isinstance(info, dict) else None)
        frame_size = tuple(frame_size or forward_out["pred_masks"].shape[-2:])
        pred_boxes = self.detr.inference(forward_out, filters=b_filters, **kwargs)
        masks_all = forward_out["pred_masks"].float()
        # on the GPU the whole chain below (up-sample, sigmoid, threshold, arg-max one-hot) is one kernel writing the int64 masks
        fused = masks_all.is_cuda and masks_all.numel() > 0 and maskth >= 0 and alo_hip.is_available()
        if fused:
            onehot_all = alo_hip.panoptic_onehot(masks_all, frame_size, maskth)
        elif masks_all.numel() > 0:
            masks_all = F.interpolate(masks_all, size=frame_size, mode="bilinear", align_corners=False)
        else:
            masks_all = masks_all.view(masks_all.shape[0], 0, *frame_size)
        if not fused:
            masks_all = F.threshold(masks_all.sigmoid(), maskth, 0.0)
        pred_masks = []
        zero = torch.zeros(*frame_size, device=masks_all.device, dtype=torch.long)
        # which mask row (if any) belongs to each kept box: worked out on the host from ONE transfer of the two filters — the
        # per-box `(ib == kept).nonzero().item()` of the straightforward loop is three device synchronisations per box
        b_host = torch.stack(list(b_filters)).cpu() if len(b_filters) else None
        m_host = torch.stack(list(m_filters)).cpu() if len(m_filters) else None
        for img, (boxes, masks, b_filter, m_filter) in enumerate(zip(pred_boxes, onehot_all if fused else masks_all, b_filters, m_filters)):
            if not fused:
                nothing = (~masks.bool()).all(dim=0, keepdim=True)  # pixels where no query passes the threshold
                onehot = torch.zeros_like(masks)
                if onehot.numel():
                    onehot.scatter_(0, masks.argmax(dim=0, keepdim=True), 1)
                masks = onehot.long() * (~nothing)
            kept = torch.where(m_host[img])[0].tolist()
            row_of = {q: r for r, q in enumerate(kept)}
            rows = [row_of.get(q, -1) for q in torch.where(b_host[img])[0].tolist()]
            rows = [r if r < len(masks) else -1 for r in rows]
            if not rows:
                masks = zero[[]].view(0, *frame_size)
            elif all(r >= 0 for r in rows):
                masks = masks[torch.tensor(rows, device=masks.device)] if rows != list(range(len(masks))) else masks
            else:
                masks = torch.stack([masks[r] if r >= 0 else zero for r in rows], dim=0)
            pred_masks.append(aloscene.Mask(masks, names=("N", "H", "W"), labels=boxes.labels))
        return pred_boxes, pred_masks
