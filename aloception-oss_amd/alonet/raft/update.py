"""RAFT update operator: motion encoder + (separable) convolutional GRU + flow and up-sampling-mask heads.

Module names follow alonet/raft/update.py (``encoder.{convc1,convc2,convf1,convf2,conv}``, ``gru.conv{z,r,q}{1,2}``,
``flow_head.{conv1,conv2}``, ``mask.{0,2}``).  ``convc1`` consumes the (B, 324, H/8, W/8) output of the correlation
lookup kernel.
"""
import torch
import torch.nn.functional as F
from torch import nn

import alo_hip


def _fusable(*tensors):
    """Inference on the GPU in fp32 with H*W a multiple of 4: the elementwise glue between MIOpen's convolutions (bias,
    ReLU, the GRU gates) runs as one HIP pass per stage instead of one PyTorch kernel per operation."""
    t = tensors[0]
    return alo_hip.fusable(*tensors) and t.dtype == torch.float32 and (t.shape[-1] * t.shape[-2]) % 4 == 0


def _conv_act(conv, x, relu=True):
    """``act(conv(x))``: the convolution runs without bias, bias + ReLU are one in-place pass."""
    out = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    if not out.is_contiguous():
        out = out.contiguous()
    return alo_hip.bias_act_nchw_(out, conv.bias, relu)


def _cached(module, name, key_tensors, build):
    """Derived inference-time tensors (merged / rescaled weights) cached on the module, keyed on parameter versions."""
    key = tuple((t.data_ptr(), alo_hip.tensor_version(t)) for t in key_tensors)
    hit = module.__dict__.get(name)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, build())
        module.__dict__[name] = hit
    return hit[1]


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256, out_planes=2):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, out_planes, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x)))


def _gru_step(h, x, convz, convr, convq):
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(convz(hx))
    r = torch.sigmoid(convr(hx))
    q = torch.tanh(convq(torch.cat([r * h, x], dim=1)))
    return (1 - z) * h + z * q


class ConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)

    def forward(self, h, x):
        return _gru_step(h, x, self.convz, self.convr, self.convq)


class SepConvGRU(nn.Module):
    """GRU with a 1x5 (horizontal) pass followed by a 5x1 (vertical) pass."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        c = hidden_dim + input_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))

    def forward(self, h, x):
        if _fusable(h, x, self.convz1.weight):
            return self._forward_fused(h, x)
        h = _gru_step(h, x, self.convz1, self.convr1, self.convq1)
        return _gru_step(h, x, self.convz2, self.convr2, self.convq2)

    def _forward_fused(self, h, x):
        """Same arithmetic, different plumbing: the z and r convolutions of a pass are ONE convolution with 2C outputs,
        [h | x] and [r*h | x] live in two persistent buffers (no torch.cat), and sigmoid / tanh / gating / biases are two
        HIP passes per half (alo_gru_gate, alo_gru_update) instead of ~13 elementwise kernels."""
        B, C, H, W = h.shape
        hx = torch.empty((B, C + x.shape[1], H, W), dtype=h.dtype, device=h.device)
        rhx = torch.empty_like(hx)
        hx[:, :C].copy_(h)
        hx[:, C:].copy_(x)
        rhx[:, C:].copy_(x)
        net = torch.empty_like(h)
        halves = ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2))
        for i, (cz, cr, cq) in enumerate(halves):
            wzr, bzr = _cached(self, f"_zr{i}", (cz.weight, cr.weight, cz.bias, cr.bias),
                               lambda: (torch.cat([cz.weight, cr.weight], 0).contiguous(),
                                        torch.cat([cz.bias, cr.bias], 0).contiguous()))
            zr = F.conv2d(hx, wzr, None, cz.stride, cz.padding)
            alo_hip.gru_gate_(zr, bzr, hx, rhx, C)
            q = F.conv2d(rhx, cq.weight, None, cq.stride, cq.padding)
            alo_hip.gru_update_(q, cq.bias, zr, hx, C, net if i == 1 else None)
        return net


class SmallMotionEncoder(nn.Module):
    def __init__(self, corr_levels, corr_radius, out_planes=2):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** out_planes
        self.convc1 = nn.Conv2d(cor_planes, 96, 1)
        self.convf1 = nn.Conv2d(out_planes, 64, 7, padding=3)
        self.convf2 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv = nn.Conv2d(128, 80, 3, padding=1)

    def forward(self, flow, corr):
        cor = F.relu(self.convc1(corr))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        out = F.relu(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


class BasicMotionEncoder(nn.Module):
    def __init__(self, corr_levels, corr_radius, out_planes=2):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** out_planes
        self.convc1 = nn.Conv2d(cor_planes, 256, 1)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(out_planes, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - out_planes, 3, padding=1)

    def forward(self, flow, corr):
        if _fusable(flow, corr, self.convc1.weight):
            cor = _conv_act(self.convc2, _conv_act(self.convc1, corr))
            flo = _conv_act(self.convf2, _conv_act(self.convf1, flow))
            out = _conv_act(self.conv, torch.cat([cor, flo], dim=1))
            return torch.cat([out, flow], dim=1)
        cor = F.relu(self.convc2(F.relu(self.convc1(corr))))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        out = F.relu(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


class SmallUpdateBlock(nn.Module):
    def __init__(self, corr_levels, corr_radius, hidden_dim=96, out_planes=2):
        super().__init__()
        self.encoder = SmallMotionEncoder(corr_levels, corr_radius, out_planes=out_planes)
        self.gru = ConvGRU(hidden_dim=hidden_dim, input_dim=hidden_dim + 49)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=128, out_planes=out_planes)

    def forward(self, net, inp, corr, flow):
        inp = torch.cat([inp, self.encoder(flow, corr)], dim=1)
        net = self.gru(net, inp)
        return net, None, self.flow_head(net)


class BasicUpdateBlock(nn.Module):
    def __init__(self, corr_levels, corr_radius, hidden_dim=128, input_dim=128, out_planes=2):
        super().__init__()
        self.encoder = BasicMotionEncoder(corr_levels, corr_radius, out_planes=out_planes)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256, out_planes=out_planes)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1))

    def forward(self, net, inp, corr, flow, upsample=True):
        inp = torch.cat([inp, self.encoder(flow, corr)], dim=1)
        net = self.gru(net, inp)
        if _fusable(net, self.mask[0].weight):
            fh, m0, m2 = self.flow_head, self.mask[0], self.mask[2]
            delta_flow = fh.conv2(_conv_act(fh.conv1, net))
            # the 0.25 of the original RAFT (gradient balancing) is folded into the last convolution's weight and bias
            w2, b2 = _cached(self, "_mask_quarter", (m2.weight, m2.bias), lambda: (0.25 * m2.weight, 0.25 * m2.bias))
            up_mask = F.conv2d(_conv_act(m0, net), w2, b2, m2.stride, m2.padding)
            return net, up_mask, delta_flow
        delta_flow = self.flow_head(net)
        return net, 0.25 * self.mask(net), delta_flow  # 0.25: gradient balancing of the original RAFT
