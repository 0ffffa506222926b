"""Mask head: a small convolutional decoder with GroupNorm that up-samples FPN-style through three backbone levels.

Reference: alonet/detr_panoptic/nn/FPNstyle.py:14-84 (same layer names ``lay1..5, gn1..5, out_lay, adapter1..3``).
Input: the projected stride-32 feature repeated per query and concatenated with that query's attention maps; output one
mask logit map per (image, query) at the resolution of the finest FPN level given.
"""
import torch
import torch.nn.functional as F
from torch import nn


def _expand(tensor, length):
    return tensor.unsqueeze(1).repeat(1, length, 1, 1, 1).flatten(0, 1)


class FPNstyleCNN(nn.Module):
    def __init__(self, dim, fpn_dims, context_dim):
        super().__init__()
        inter = [dim, context_dim // 2, context_dim // 4, context_dim // 8, context_dim // 16, context_dim // 64]
        self.lay1 = nn.Conv2d(dim, dim, 3, padding=1)
        self.gn1 = nn.GroupNorm(8, dim)
        self.lay2 = nn.Conv2d(dim, inter[1], 3, padding=1)
        self.gn2 = nn.GroupNorm(8, inter[1])
        self.lay3 = nn.Conv2d(inter[1], inter[2], 3, padding=1)
        self.gn3 = nn.GroupNorm(8, inter[2])
        self.lay4 = nn.Conv2d(inter[2], inter[3], 3, padding=1)
        self.gn4 = nn.GroupNorm(8, inter[3])
        self.lay5 = nn.Conv2d(inter[3], inter[4], 3, padding=1)
        self.gn5 = nn.GroupNorm(8, inter[4])
        self.out_lay = nn.Conv2d(inter[4], 1, 3, padding=1)
        self.dim = dim
        self.adapter1 = nn.Conv2d(fpn_dims[0], inter[1], 1)
        self.adapter2 = nn.Conv2d(fpn_dims[1], inter[2], 1)
        self.adapter3 = nn.Conv2d(fpn_dims[2], inter[3], 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x, bbox_mask, fpns):
        """x (B,C,H,W), bbox_mask (B,Q,heads,H,W), fpns: three (B,c_i,h_i,w_i) maps, coarse to fine -> (B*Q,1,h,w)."""
        x = torch.cat([_expand(x, bbox_mask.shape[1]), bbox_mask.flatten(0, 1)], 1)
        x = F.relu(self.gn1(self.lay1(x)))
        x = F.relu(self.gn2(self.lay2(x)))
        for adapter, lay, gn, fpn in ((self.adapter1, self.lay3, self.gn3, fpns[0]),
                                      (self.adapter2, self.lay4, self.gn4, fpns[1]),
                                      (self.adapter3, self.lay5, self.gn5, fpns[2])):
            cur = adapter(fpn)
            if cur.size(0) != x.size(0):
                cur = _expand(cur, x.size(0) // cur.size(0))
            x = cur + F.interpolate(x, size=cur.shape[-2:], mode="nearest")
            x = F.relu(gn(lay(x)))
        return self.out_lay(x)
