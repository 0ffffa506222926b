"""Mask head: a small convolutional decoder with GroupNorm that up-samples FPN-style through three backbone levels.

Reference: alonet/detr_panoptic/nn/FPNstyle.py:14-84 (same layer names ``lay1..5, gn1..5, out_lay, adapter1..3``).
Input: the projected stride-32 feature repeated per query and concatenated with that query's attention maps; output one
mask logit map per (image, query) at the resolution of the finest FPN level given.
"""
import torch
import torch.nn.functional as F
from torch import nn

import alo_hip
from alonet.detr.backbone import conv1x1_as_gemm


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

    # ---- inference (bf16, CUDA, no autograd): the two widest convolutions on alo_conv3x3_nhwc ---------------------------------
    # lay1 / lay2 run over B*Q maps of 264 channels at stride 32; MIOpen needs 2.4 + 0.6 ms for them (B*Q = 128), the implicit-GEMM
    # kernel 0.33 + 0.13 ms with the channel counts zero-padded to multiples of 64 (weights padded once, cached).
    @staticmethod
    def _pad64(c):
        return (c + 63) // 64 * 64

    def _padded_weights(self, lay):
        w, b = lay.weight, lay.bias
        key = (alo_hip.tensor_version(w), w.data_ptr(), None if b is None else alo_hip.tensor_version(b))
        hit = lay.__dict__.get("_alo_padded")
        if hit is None or hit[0] != key:
            cout, cin = w.shape[:2]
            wp = torch.zeros((self._pad64(cout), self._pad64(cin), 3, 3), dtype=w.dtype, device=w.device)
            wp[:cout, :cin] = w.detach()
            bp = torch.zeros(self._pad64(cout), dtype=w.dtype, device=w.device)
            if b is not None:
                bp[:cout] = b.detach()
            hit = (key, wp.contiguous(memory_format=torch.channels_last), bp)
            lay.__dict__["_alo_padded"] = hit
        return hit[1], hit[2]

    def _wide_layers_fast(self, x, bbox_mask):
        """relu(gn2(lay2(relu(gn1(lay1(cat(expand(x), bbox_mask))))))) with both convolutions on the implicit-GEMM kernel."""
        b, c, h, w = x.shape
        nq, heads = bbox_mask.shape[1], bbox_mask.shape[2]
        cin = c + heads
        xp = torch.zeros((b * nq, self._pad64(cin), h, w), dtype=x.dtype, device=x.device).contiguous(memory_format=torch.channels_last)
        xp.view(b, nq, -1, h, w)[:, :, :c] = x.unsqueeze(1)
        xp[:, c:cin] = bbox_mask.flatten(0, 1)
        w1, b1 = self._padded_weights(self.lay1)
        y = F.relu_(self.gn1(alo_hip.conv3x3(xp, w1, b1)[:, :self.lay1.out_channels]))
        yp = torch.zeros_like(xp) if self._pad64(y.shape[1]) == xp.shape[1] else torch.zeros(
            (y.shape[0], self._pad64(y.shape[1]), h, w), dtype=y.dtype, device=y.device).contiguous(memory_format=torch.channels_last)
        yp[:, :y.shape[1]] = y
        w2, b2 = self._padded_weights(self.lay2)
        return self._norm_relu(self.gn2, alo_hip.conv3x3(yp, w2, b2)[:, :self.lay2.out_channels])

    @staticmethod
    def _norm_relu(gn, y, fast=True):
        """relu(gn(y)); channels-last in and out on the GroupNorm kernel of this library when it covers the layer (2, 4 or 8k
        channels per group) — ATen's works on NCHW, a layout copy either side on a channels-last activation."""
        if fast and isinstance(gn, nn.GroupNorm) and alo_hip.groupnorm_nhwc_supported(y, gn):   # (a fine-tuning variant may swap in BatchNorm2d)
            return alo_hip.groupnorm_nhwc(y, gn, relu=True)
        return F.relu(gn(y))

    def forward(self, x, bbox_mask, fpns):
        """x (B,C,H,W), bbox_mask (B,Q,heads,H,W), fpns: three (B,c_i,h_i,w_i) maps, coarse to fine -> (B*Q,1,h,w)."""
        fast = (x.is_cuda and x.dtype == torch.bfloat16 and self.lay1.weight.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and bbox_mask.dtype == x.dtype and bbox_mask.numel() > 0)
        if fast:
            x = self._wide_layers_fast(x, bbox_mask)
        else:
            x = torch.cat([_expand(x, bbox_mask.shape[1]), bbox_mask.flatten(0, 1)], 1)
            x = F.relu(self.gn1(self.lay1(x)))
            x = F.relu(self.gn2(self.lay2(x)))
        for adapter, lay, gn, fpn in ((self.adapter1, self.lay3, self.gn3, fpns[0]),
                                      (self.adapter2, self.lay4, self.gn4, fpns[1]),
                                      (self.adapter3, self.lay5, self.gn5, fpns[2])):
            if (fast and fpn.is_contiguous(memory_format=torch.channels_last) and adapter.kernel_size == (1, 1)
                    and adapter.stride == (1, 1) and adapter.padding == (0, 0) and adapter.groups == 1 and adapter.out_channels % 4 == 0):
                # 1x1 adapter over the NHWC rows as a GEMM (bias in its epilogue): no MIOpen kernel — and no MIOpen search on the
                # first call of a shape — anywhere in the head
                cur = conv1x1_as_gemm(fpn, adapter.weight, adapter.bias)
            else:
                cur = adapter(fpn)
            if fast and cur.size(0) and x.size(0) % cur.size(0) == 0 and cur.shape[1] % 8 == 0:
                # inference: the per-query copy of the adapter output, the up-sampled copy of x and the add are ONE pass over the
                # stage's largest tensor (B*Q maps at this level's resolution: 547 MB at stride 4 for 8 frames x 16 queries), and the
                # GroupNorm + ReLU behind the convolution stays channels-last
                x = alo_hip.upsample_add(x.contiguous(memory_format=torch.channels_last), cur.contiguous(memory_format=torch.channels_last))
            else:
                if cur.size(0) != x.size(0):
                    cur = _expand(cur, x.size(0) // cur.size(0))
                x = cur + F.interpolate(x, size=cur.shape[-2:], mode="nearest")
            x = self._norm_relu(gn, self._conv(lay, x, fast), fast)
        return self._conv(self.out_lay, x, fast)

    @staticmethod
    def _conv(lay, x, fast):
        """lay(x); at inference the few-channel 3x3 convolutions of the fine levels run on alo_conv3x3_small_nhwc (memory-bound layers
        on which the library's kernels reach 0.4-0.9 TB/s) and the 128 -> 64 one on the implicit-GEMM kernel of the backbone."""
        if fast and alo_hip.conv3x3_small_supported(x, lay):
            return alo_hip.conv3x3_small(x, lay)
        if (fast and lay.kernel_size == (3, 3) and lay.groups == 1 and lay.padding_mode == "zeros"
                and alo_hip.conv3x3_supported(x, lay.weight, lay.stride, lay.padding, lay.dilation)):
            return alo_hip.conv3x3(x, lay.weight, lay.bias, False, lay.stride)
        return lay(x)
