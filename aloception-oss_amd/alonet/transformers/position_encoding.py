"""Sine positional encoding over the un-padded area of a feature map.

Reference: alonet/transformers/position_encoding.py:9-72 — cumulative count of valid pixels along y and x, optionally
centred (-0.5) and normalised to [0, 2*pi], expanded on ``num_pos_feats`` sin/cos frequencies per axis; output is
(B, 2*num_pos_feats, H, W) with the y block first.
"""
import math

import torch
from torch import nn


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None, center=False):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self.center = center

    def dim_t(self, device):
        """``temperature ** (2 * (i // 2) / num_pos_feats)`` for i < num_pos_feats (fp32, cached per device)."""
        cache = self.__dict__.setdefault("_dim_t", {})
        key = str(device)
        if key not in cache:
            idx = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
            cache[key] = self.temperature ** (2 * torch.div(idx, 2, rounding_mode="floor") / self.num_pos_feats)
        return cache[key]

    def forward(self, ftmap_mask):
        """``ftmap_mask = (feature_map (B,C,H,W), mask (B,1,H,W))``, mask true/1 on padding."""
        ft_maps, mask = ftmap_mask
        valid = ~(mask[:, 0].to(torch.bool))
        y_embed = valid.cumsum(1, dtype=torch.float32)
        x_embed = valid.cumsum(2, dtype=torch.float32)
        if self.normalize:
            if self.center:
                y_embed = y_embed - 0.5
                x_embed = x_embed - 0.5
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        dim_t = self.dim_t(ft_maps.device)

        def expand(embed):  # even frequencies -> sin, odd -> cos, interleaved
            p = embed[..., None] / dim_t
            return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=4).flatten(3)

        return torch.cat((expand(y_embed), expand(x_embed)), dim=3).permute(0, 3, 1, 2)
