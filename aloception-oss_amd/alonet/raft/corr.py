"""RAFT correlation block on the gfx950 kernels: all-pairs volume + pyramid on the fp16 matrix pipe at fp32 accuracy (scaled
two-term operand split, three products — csrc/corr.hip), windowed lookup as a gather kernel.

Drop-in for the reference's ``CorrBlock`` (alonet/raft/corr.py:12-60) through RAFT's ``corr_block=`` constructor
hook (alonet/raft/raft.py:47-60,168,185): ``CorrBlock(fmap1, fmap2, num_levels=4, radius=4)`` builds
``corr_pyramid`` (list of ``(B*H*W, 1, h_l, w_l)`` float32 tensors) and ``corr_fn(coords)`` returns the
``(B, num_levels*(2r+1)^2, H, W)`` float32 window features.  ``AlternateCorrBlock`` is not provided: the reference's
version needs the absent third-party ``alt_cuda_corr`` extension and is unreachable (corr.py:5-9,86).

Differentiability.  The reference's block is plain autograd-able torch code, so RAFT can be fine-tuned through it.  The HIP
kernels are forward kernels; under autograd they run inside ``torch.autograd.Function`` s whose BACKWARD re-evaluates the torch
formulation below (``pyramid_torch`` / ``lookup_torch`` — matmul, ``avg_pool2d``, ``grid_sample``) on the device and
differentiates that: gradients are those of the reference's own graph, the forward values are the kernels'.  The backward
holds a second copy of the volume while it runs (training crops are small; BASELINE.json trains no RAFT config).
"""
import math

import torch
import torch.nn.functional as F

import alo_hip

from .utils.utils import bilinear_sampler


# ---- the torch formulation: what the reference computes, op for op (used for gradients, and as ``TorchCorrBlock``) ----------------
def pyramid_torch(fmap1, fmap2, num_levels=4):
    """corr.py:13-27,52-60: ``<fmap1[b,:,i], fmap2[b,:,j]> / sqrt(C)`` as ``(B*H*W, 1, H, W)`` + ``num_levels - 1`` 2x2 means."""
    B, C, H, W = fmap1.shape
    vol = torch.matmul(fmap1.reshape(B, C, H * W).transpose(1, 2), fmap2.reshape(B, C, H * W)) / math.sqrt(C)
    pyramid = [vol.reshape(B * H * W, 1, H, W)]
    for _ in range(num_levels - 1):
        pyramid.append(F.avg_pool2d(pyramid[-1], 2, stride=2))
    return pyramid


def lookup_torch(pyramid, coords, radius):
    """corr.py:29-50: a (2r+1)^2 window around ``coords / 2^l`` on every level; the window's FIRST axis moves x (the reference
    stacks ``meshgrid(dy, dx)`` onto (x, y) coordinates), bilinear with corners aligned and zeros outside."""
    B, _, H, W = coords.shape
    r = radius
    centre = coords.permute(0, 2, 3, 1).reshape(B * H * W, 1, 1, 2)
    steps = torch.linspace(-r, r, 2 * r + 1, device=coords.device, dtype=coords.dtype)
    window = torch.stack(torch.meshgrid(steps, steps, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    feats = [bilinear_sampler(vol, centre / 2 ** lvl + window).view(B, H, W, -1) for lvl, vol in enumerate(pyramid)]
    return torch.cat(feats, dim=-1).permute(0, 3, 1, 2).contiguous().float()


class TorchCorrBlock:
    """The same block on stock torch ops only (any device, differentiable end to end): ``RAFT(corr_block=TorchCorrBlock)``."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        self.corr_pyramid = pyramid_torch(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        return lookup_torch(self.corr_pyramid, coords, self.radius)

    @staticmethod
    def corr(fmap1, fmap2):
        B, _, H, W = fmap1.shape
        return pyramid_torch(fmap1, fmap2, 1)[0].view(B, H, W, 1, H, W)


# ---- HIP forward, torch-formulation backward -----------------------------------------------------------------------------------
def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class _BuildFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap1, fmap2, num_levels):
        ctx.save_for_backward(fmap1, fmap2)
        ctx.num_levels = num_levels
        return tuple(alo_hip.corr_build(fmap1, fmap2, num_levels))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        fmap1, fmap2 = ctx.saved_tensors
        with torch.enable_grad():
            a = fmap1.detach().requires_grad_(ctx.needs_input_grad[0])
            b = fmap2.detach().requires_grad_(ctx.needs_input_grad[1])
            pyramid = pyramid_torch(a, b, ctx.num_levels)
            pairs = [(p, g) for p, g in zip(pyramid, grads) if g is not None]
            wrt = [t for t in (a, b) if t.requires_grad]
            got = torch.autograd.grad([p for p, _ in pairs], wrt, [g for _, g in pairs], allow_unused=True) if pairs and wrt else ()
        it = iter(got)
        return (next(it) if a.requires_grad and pairs else None, next(it) if b.requires_grad and pairs else None, None)


class _LookupFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, radius, *pyramid):
        ctx.save_for_backward(coords, *pyramid)
        ctx.radius = radius
        return alo_hip.corr_lookup(list(pyramid), coords, radius)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        coords, *pyramid = ctx.saved_tensors
        with torch.enable_grad():
            c = coords.detach().requires_grad_(ctx.needs_input_grad[0])
            pyr = [p.detach().requires_grad_(ctx.needs_input_grad[2 + i]) for i, p in enumerate(pyramid)]
            wrt = [t for t in [c] + pyr if t.requires_grad]
            got = torch.autograd.grad(lookup_torch(pyr, c, ctx.radius), wrt, grad_out, allow_unused=True) if wrt else ()
        it = iter(got)
        return (next(it) if c.requires_grad else None, None) + tuple(next(it) if p.requires_grad else None for p in pyr)


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        if _needs_grad(fmap1, fmap2):
            self.corr_pyramid = list(_BuildFunction.apply(fmap1.contiguous(), fmap2.contiguous(), num_levels))
        else:
            self.corr_pyramid = alo_hip.corr_build(fmap1, fmap2, num_levels)

    def lookup_conv1x1(self, coords, weight, bias, relu=True):
        """``act(conv1x1(self(coords)))`` without materialising the window features (the motion encoder's ``convc1``);
        None when the fused kernel does not cover the configuration or a gradient is wanted (the caller then convolves
        ``self(coords)``).  A C-ABI extra (``alo_corr_lookup_conv1x1``), NOT used by ``RAFT.forward``: measured 0.23 ms against
        0.20 ms for lookup + convolution, so the model keeps the two-kernel form."""
        if _needs_grad(coords, weight, bias, *self.corr_pyramid):
            return None
        if not alo_hip.corr_lookup_conv1x1_supported(self.corr_pyramid, weight, self.radius):
            return None
        return alo_hip.corr_lookup_conv1x1(self.corr_pyramid, coords.float(), weight, bias, self.radius, relu)

    def __call__(self, coords):
        coords = coords.float()
        if _needs_grad(coords, *self.corr_pyramid):
            return _LookupFunction.apply(coords.contiguous(), self.radius, *self.corr_pyramid)
        return alo_hip.corr_lookup(self.corr_pyramid, coords, self.radius)

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation only: (B,C,H,W) x2 -> (B,H,W,1,H,W), scaled by 1/sqrt(C)."""
        B, _, H, W = fmap1.shape
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        if _needs_grad(fmap1, fmap2):
            (vol,) = _BuildFunction.apply(fmap1.contiguous(), fmap2.contiguous(), 1)
        else:
            (vol,) = alo_hip.corr_build(fmap1, fmap2, 1)
        return vol.view(B, H, W, 1, H, W)
