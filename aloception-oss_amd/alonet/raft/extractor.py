"""RAFT feature / context encoders (stock PyTorch-ROCm convolutions).

``BasicEncoder``: 7x7 stride-2 stem, three residual stages (64, 96, 128 channels; strides 1, 2, 2) and a 1x1 output
convolution -> features at 1/8 resolution.  Module names match alonet/raft/extractor.py:113-187 (``conv1, norm1,
layer{1,2,3}.{0,1}.{conv1,conv2,norm1,norm2,downsample.{0,1}}, conv2``) so RAFT checkpoints load unchanged.
``SmallEncoder`` is the bottleneck variant used by RAFT-small.
"""
import torch
from torch import nn


def _norm(kind, channels, groups=None):
    if kind == "group":
        return nn.GroupNorm(num_groups=groups or channels // 8, num_channels=channels)
    if kind == "batch":
        return nn.BatchNorm2d(channels)
    if kind == "instance":
        return nn.InstanceNorm2d(channels)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(f"unknown norm_fn {kind!r}")


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes, planes // 8)
        self.norm2 = _norm(norm_fn, planes, planes // 8)
        self.downsample = None
        if stride != 1:
            self.norm3 = _norm(norm_fn, planes, planes // 8)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class BottleneckBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        mid = planes // 4
        self.conv1 = nn.Conv2d(in_planes, mid, 1)
        self.conv2 = nn.Conv2d(mid, mid, 3, padding=1, stride=stride)
        self.conv3 = nn.Conv2d(mid, planes, 1)
        self.relu = nn.ReLU(inplace=True)
        groups = planes // 8
        self.norm1 = _norm(norm_fn, mid, groups)
        self.norm2 = _norm(norm_fn, mid, groups)
        self.norm3 = _norm(norm_fn, planes, groups)
        self.downsample = None
        if stride != 1:
            self.norm4 = _norm(norm_fn, planes, groups)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm4)

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        y = self.relu(self.norm3(self.conv3(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class _Encoder(nn.Module):
    block = None
    widths = None

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        w0, w1, w2 = self.widths
        self.norm1 = _norm(norm_fn, w0, 8)
        self.conv1 = nn.Conv2d(3, w0, 7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = w0
        self.layer1 = self._make_layer(w0, stride=1)
        self.layer2 = self._make_layer(w1, stride=2)
        self.layer3 = self._make_layer(w2, stride=2)
        self.conv2 = nn.Conv2d(w2, output_dim, 1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        layers = (self.block(self.in_planes, dim, self.norm_fn, stride=stride), self.block(dim, dim, self.norm_fn, 1))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def forward(self, x):
        pair = isinstance(x, (tuple, list))  # two frames share one pass: stacked on the batch axis
        if pair:
            n = x[0].shape[0]
            x = torch.cat(x, dim=0)
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        return torch.split(x, [n, n], dim=0) if pair else x


class BasicEncoder(_Encoder):
    block = ResidualBlock
    widths = (64, 96, 128)


class SmallEncoder(_Encoder):
    block = BottleneckBlock
    widths = (32, 64, 96)
