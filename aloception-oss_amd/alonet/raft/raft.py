"""RAFT optical flow (https://arxiv.org/abs/2003.12039) with the gfx950 correlation kernels.

``RAFT.forward(frame1, frame2, iters=12, flow_init=None, only_last=False)`` and ``.inference`` keep the reference's
contract (alonet/raft/raft.py:134-205): frames are ``minmax_sym``-normalised ``aloscene.Frame`` batches whose H and W
are multiples of 8; the result is a list (one dict per iteration) with ``flow`` (1/8 resolution), ``hidden_state``,
``up_mask``, ``delta_flow`` and — for every entry, or only the last with ``only_last`` — ``up_flow`` (full resolution,
convex up-sampling).  The correlation volume is built once by ``corr_block`` (default: the HIP ``CorrBlock``) and
looked up once per iteration.
"""
import torch
import torch.nn.functional as F
from torch import nn

from aloscene import Flow
from alonet.common import load_weights

from .corr import CorrBlock
from .extractor import BasicEncoder
from .update import BasicUpdateBlock
from .utils.utils import coords_grid, upflow8


class RAFTBase(nn.Module):
    """Sub-classes define ``hidden_dim, context_dim, corr_levels, corr_radius, out_plane`` (class attributes)."""

    hidden_dim = context_dim = corr_levels = corr_radius = out_plane = None

    def __init__(self, fnet, cnet, update_block, weights=None, corr_block=CorrBlock, device=torch.device("cpu")):
        super().__init__()
        missing = [a for a in ("hidden_dim", "context_dim", "corr_levels", "corr_radius", "out_plane")
                   if getattr(self, a) is None]
        if missing:
            raise NotImplementedError(f"{type(self).__name__} must define the class attributes {missing}")
        self.fnet, self.cnet, self.update_block = fnet, cnet, update_block
        self.corr_block = corr_block
        if weights is not None:
            load_weights(self, weights, device)

    @property
    def hdim(self):
        return self.hidden_dim

    @property
    def cdim(self):
        return self.context_dim

    def build_fnet(self, encoder_cls=BasicEncoder, output_dim=256):
        return encoder_cls(output_dim=output_dim, norm_fn="instance", dropout=self.dropout)

    def build_cnet(self, encoder_cls=BasicEncoder):
        return encoder_cls(output_dim=self.hdim + self.cdim, norm_fn="batch", dropout=self.dropout)

    def build_update_block(self, update_cls=BasicUpdateBlock):
        return update_cls(self.corr_levels, self.corr_radius, hidden_dim=self.hdim, out_planes=self.out_plane)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def initialize_flow(self, img):
        """flow = coords1 - coords0, both (N, 2, H/8, W/8) pixel grids."""
        N, _, H, W = img.shape
        grid = coords_grid(N, H // 8, W // 8, device=img.device)
        return grid, grid.clone()

    def upsample_flow(self, flow, mask):
        """(N, 2, H, W) -> (N, 2, 8H, 8W): each fine pixel is a softmax-weighted (convex) mix of its 3x3 coarse
        neighbourhood; bilinear when no mask is predicted."""
        if mask is None:
            return upflow8(flow)
        N, _, H, W = flow.shape
        mask = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
        nbrs = F.unfold(8 * flow, [3, 3], padding=1).view(N, self.out_plane, 9, 1, 1, H, W)
        up = torch.sum(mask * nbrs, dim=2).permute(0, 1, 4, 2, 5, 3)
        return up.reshape(N, self.out_plane, 8 * H, 8 * W)

    def forward_heads(self, m_outputs, only_last=False):
        for out in (m_outputs[-1:] if only_last else m_outputs):
            out["up_flow"] = self.upsample_flow(out["flow"], out["up_mask"])
        return m_outputs

    def forward(self, frame1, frame2, iters=12, flow_init=None, only_last=False):
        assert frame1.normalization == "minmax_sym"
        assert frame2.normalization == "minmax_sym"
        frame1, frame2 = frame1.as_tensor(), frame2.as_tensor()

        fmap1, fmap2 = self.fnet([frame1, frame2])
        corr_fn = self.corr_block(fmap1.float(), fmap2.float(), radius=self.corr_radius)

        net, inp = torch.split(self.cnet(frame1), [self.hdim, self.cdim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)

        coords0, coords1 = self.initialize_flow(frame1)
        if flow_init is not None:
            coords1 = coords1 + flow_init

        m_outputs = []
        for _ in range(iters):
            coords1 = coords1.detach()
            flow = coords1 - coords0
            corr = corr_fn(coords1).to(net.dtype)
            net, up_mask, delta_flow = self.update_block(net, inp, corr, flow.to(net.dtype))
            coords1 = coords1 + delta_flow.float()
            m_outputs.append({"flow": coords1 - coords0, "hidden_state": net, "up_mask": up_mask,
                              "delta_flow": delta_flow})
        return self.forward_heads(m_outputs, only_last=only_last)

    @torch.no_grad()
    def inference(self, m_outputs, only_last=False):
        def wrap(out):
            return Flow(out["up_flow"], names=("B", "C", "H", "W"))

        return wrap(m_outputs[-1]) if only_last else [wrap(o) for o in m_outputs]


class RAFT(RAFTBase):
    """RAFT (basic): 256-d feature encoder, 128+128 context encoder, 4-level radius-4 correlation, SepConvGRU."""

    hidden_dim = 128
    context_dim = 128
    corr_levels = 4
    corr_radius = 4
    out_plane = 2

    def __init__(self, dropout=0, **kwargs):
        self.dropout = dropout
        fnet = self.build_fnet(encoder_cls=BasicEncoder, output_dim=256)
        cnet = self.build_cnet(encoder_cls=BasicEncoder)
        update_block = self.build_update_block(update_cls=BasicUpdateBlock)
        super().__init__(fnet, cnet, update_block, **kwargs)
