"""RAFT update operator: motion encoder + (separable) convolutional GRU + flow and up-sampling-mask heads.

Module names follow alonet/raft/update.py (``encoder.{convc1,convc2,convf1,convf2,conv}``, ``gru.conv{z,r,q}{1,2}``,
``flow_head.{conv1,conv2}``, ``mask.{0,2}``).  ``convc1`` consumes the (B, 324, H/8, W/8) output of the correlation
lookup kernel.
"""
import torch
import torch.nn.functional as F
from torch import nn


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
        h = _gru_step(h, x, self.convz1, self.convr1, self.convq1)
        return _gru_step(h, x, self.convz2, self.convr2, self.convq2)


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
        delta_flow = self.flow_head(net)
        return net, 0.25 * self.mask(net), delta_flow  # 0.25: gradient balancing of the original RAFT
