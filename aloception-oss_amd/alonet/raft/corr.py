"""RAFT correlation block on the gfx950 kernels (fp32 MFMA all-pairs volume + pyramid, gather lookup).

Drop-in for the reference's ``CorrBlock`` (alonet/raft/corr.py:12-60) through RAFT's ``corr_block=`` constructor
hook (alonet/raft/raft.py:47-60,168,185): ``CorrBlock(fmap1, fmap2, num_levels=4, radius=4)`` builds
``corr_pyramid`` (list of ``(B*H*W, 1, h_l, w_l)`` float32 tensors) and ``corr_fn(coords)`` returns the
``(B, num_levels*(2r+1)^2, H, W)`` float32 window features.  ``AlternateCorrBlock`` is not provided: the reference's
version needs the absent third-party ``alt_cuda_corr`` extension and is unreachable (corr.py:5-9,86).
"""
import torch

import alo_hip


def _no_backward(*tensors):
    """The HIP correlation kernels are forward-only: say so instead of silently cutting the graph (the reference's
    CorrBlock is differentiable torch code, so RAFT's feature encoder would otherwise stop receiving gradients)."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        raise RuntimeError(
            "alonet.raft.CorrBlock (HIP) has no backward: the correlation volume / lookup kernels are inference-only. "
            "Run RAFT under torch.no_grad(), or pass a differentiable corr_block= to the model for training.")


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        _no_backward(fmap1, fmap2)
        self.num_levels = num_levels
        self.radius = radius
        self.corr_pyramid = alo_hip.corr_build(fmap1.float(), fmap2.float(), num_levels)

    def lookup_conv1x1(self, coords, weight, bias, relu=True):
        """``act(conv1x1(self(coords)))`` without materialising the window features (the motion encoder's ``convc1``);
        None when the fused kernel does not cover the configuration (the caller then convolves ``self(coords)``)."""
        _no_backward(coords, weight)
        if not alo_hip.corr_lookup_conv1x1_supported(self.corr_pyramid, weight, self.radius):
            return None
        return alo_hip.corr_lookup_conv1x1(self.corr_pyramid, coords.float(), weight, bias, self.radius, relu)

    def __call__(self, coords):
        _no_backward(coords)
        return alo_hip.corr_lookup(self.corr_pyramid, coords.float(), self.radius)

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation only: (B,C,H,W) x2 -> (B,H,W,1,H,W), scaled by 1/sqrt(C)."""
        _no_backward(fmap1, fmap2)
        B, _, H, W = fmap1.shape
        (vol,) = alo_hip.corr_build(fmap1.float(), fmap2.float(), 1)
        return vol.view(B, H, W, 1, H, W)
