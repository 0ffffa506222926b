"""ResNet backbone with frozen batch-norm + positional encoding, on stock PyTorch-ROCm (MIOpen convolutions).

Re-states alonet/detr/backbone.py:50-182.  torchvision is not available, so the ResNet-50/101 body is written out here
with torchvision's module names (``conv1, bn1, layer1..4.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}``): a
reference checkpoint's ``backbone.0.body.*`` keys load unchanged.  Stride sits on the 3x3 convolution of each
bottleneck (torchvision's "v1.5" layout).
"""
import torch
import torch.nn.functional as F
from torch import nn

import alo_hip

from .misc import assert_and_export_onnx


class FrozenBatchNorm2d(nn.Module):
    """Batch-norm with fixed statistics and affine parameters (buffers, so they are never trained); eps = 1e-5."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def scale_shift(self):
        scale = self.weight * (self.running_var + 1e-5).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + shift.reshape(1, -1, 1, 1).to(x.dtype)


def _weight_2d(weight):
    """(Cout, Cin, 1, 1) -> (Cout, Cin), the SAME view object every call: the packed copy the streaming kernels make rides on it."""
    w2 = weight.__dict__.get("_alo_2d") if hasattr(weight, "__dict__") else None
    if w2 is None or alo_hip.tensor_version(w2) != alo_hip.tensor_version(weight) or w2.data_ptr() != weight.data_ptr():
        w2 = weight.reshape(weight.shape[0], -1)
        try:
            weight._alo_2d = w2
        except AttributeError:
            pass
    return w2


def conv1x1_as_gemm(x, weight, bias, stride=(1, 1), relu=False, residual=None):
    """1x1 convolution of a channels-last ``x`` (N,Cin,H,W) as ``rows @ W^T`` -> channels-last (N,Cout,H',W'): the streaming
    MFMA kernels (bias, identity and ReLU in their epilogue) when the shape allows it, otherwise hipBLASLt with the bias / ReLU
    epilogue (``torch._addmm_activation``).  A strided convolution first gathers the kept pixels (a quarter of the map).
    ``residual``: channels-last (N,Cout,H',W'), added before the activation."""
    if tuple(stride) != (1, 1):
        if (stride[0] == stride[1] and residual is None and not torch.is_grad_enabled()
                and alo_hip.conv1x1_strided_supported(x, _weight_2d(weight))):
            # the streaming GEMM addresses the kept pixels itself: no gathered copy of a quarter of the map
            return alo_hip.conv1x1_strided(x, _weight_2d(weight), bias, stride[0], relu)
        x = x[:, :, ::stride[0], ::stride[1]]
    n, cin, h, w_ = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin)  # a view for stride 1 (NHWC rows are contiguous), one gather otherwise
    res_rows = None if residual is None else residual.permute(0, 2, 3, 1).reshape(-1, weight.shape[0])
    out = alo_hip.linear_auto(rows, _weight_2d(weight), bias, relu, residual=res_rows)
    return out.view(n, h, w_, -1).permute(0, 3, 1, 2)


def folded_conv_bn(conv, bn):
    """``(w', b')`` with ``bn(conv(x)) == conv'(x)``: ``w' = w * scale[:, None, None, None]`` (channels-last), ``b' = shift``.
    In eval mode the folded tensors are cached (keyed on the parameter versions); in training mode they are rebuilt every
    call so autograd still reaches ``conv.weight``."""
    key = (conv.weight.data_ptr(), alo_hip.tensor_version(conv.weight), conv.weight.dtype, bn.weight.data_ptr(), alo_hip.tensor_version(bn.weight),
           alo_hip.tensor_version(bn.running_var), alo_hip.tensor_version(bn.running_mean), alo_hip.tensor_version(bn.bias))
    cached = conv.__dict__.get("_folded")
    if conv.training or torch.is_grad_enabled() and conv.weight.requires_grad or cached is None or cached[0] != key:
        scale, shift = bn.scale_shift()
        w = (conv.weight.float() * scale.float().reshape(-1, 1, 1, 1)).to(conv.weight.dtype)
        w = w.contiguous(memory_format=torch.channels_last)
        b = shift.to(conv.weight.dtype)
        if not conv.training and not (torch.is_grad_enabled() and conv.weight.requires_grad):
            conv.__dict__["_folded"] = (key, w.detach(), b.detach())
        return w, b
    return cached[1], cached[2]


def conv_bn(x, conv, bn, relu=False, residual=None):
    """``act(bn(conv(x)) [+ residual])`` with the frozen batch-norm folded into the convolution's weight and bias.

    A FrozenBatchNorm2d is a fixed per-channel affine map, so ``bn(conv(x)) == conv'(x)`` with
    ``w' = w * scale[:, None, None, None]`` and ``b' = shift``: two full passes over the activation (multiply, add)
    disappear.  In eval mode the folded tensors are cached (keyed on the parameter versions); in training mode they
    are rebuilt every call so autograd still reaches ``conv.weight``.  The modules and their state-dict keys are
    untouched.
    """
    if isinstance(bn, FrozenBatchNorm2d) and conv.bias is None:
        w, b = folded_conv_bn(conv, bn)
        if alo_hip.fusable(x, w) and x.is_contiguous(memory_format=torch.channels_last):
            # inference on the GPU, NHWC activations
            if conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and w.shape[0] % 4 == 0:
                # a 1x1 convolution over NHWC rows IS a matrix product: hipBLASLt runs it 1.5-6x faster than MIOpen's
                # implicit-GEMM kernels at these shapes and takes bias + ReLU in its epilogue
                rows_probe = x.permute(0, 2, 3, 1)
                w2 = _weight_2d(w)
                if residual is None or (tuple(conv.stride) == (1, 1) and residual.is_contiguous(memory_format=torch.channels_last)
                                        and (alo_hip.linear_shortk_supported(rows_probe, w2)
                                             or alo_hip.linear_packed_supported(rows_probe, w2))):
                    # bias + identity + ReLU ride in the epilogue of the streaming GEMM
                    return conv1x1_as_gemm(x, w, b, conv.stride, relu=relu, residual=residual)
                out = conv1x1_as_gemm(x, w, None, conv.stride)
                return alo_hip.bias_act_(out, b, residual, relu)
            if residual is None and alo_hip.conv3x3_supported(x, w, conv.stride, conv.padding, conv.dilation, conv.groups):
                # the bottleneck's 3x3: implicit GEMM on MFMA, bias + ReLU in its epilogue (2-3.5x MIOpen at these shapes)
                return alo_hip.conv3x3(x, w, b, relu, conv.stride)
            if relu or residual is not None:
                # bias (+ identity) + ReLU are ONE in-place pass over the convolution output instead of MIOpen's separate
                # bias kernel followed by add / relu kernels
                out = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
                if out.is_contiguous(memory_format=torch.channels_last) and out.shape[1] % 4 == 0:
                    return alo_hip.bias_act_(out, b, residual, relu)
                out = out + b.view(1, -1, 1, 1)
            else:
                out = F.conv2d(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups)
        else:
            out = F.conv2d(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups)
    else:
        out = bn(conv(x))
    if residual is not None:
        out = out + residual
    return F.relu_(out) if relu else out


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, norm_layer=FrozenBatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else conv_bn(x, self.downsample[0], self.downsample[1])
        out = conv_bn(x, self.conv1, self.bn1, relu=True)
        out = conv_bn(out, self.conv2, self.bn2, relu=True)
        return conv_bn(out, self.conv3, self.bn3, relu=True, residual=identity)


_DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


class ResNetBody(nn.Module):
    """conv1 .. layer4 of a torchvision ResNet; ``forward`` returns the outputs of the requested stages."""

    def __init__(self, name="resnet50", replace_stride_with_dilation=(False, False, False),
                 norm_layer=FrozenBatchNorm2d, return_layers=None):
        super().__init__()
        if name not in _DEPTHS:
            raise ValueError(f"backbone {name!r} is not available (have {sorted(_DEPTHS)})")
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        depths = _DEPTHS[name]
        self.layer1 = self._stage(64, depths[0], 1, False, norm_layer)
        self.layer2 = self._stage(128, depths[1], 2, replace_stride_with_dilation[0], norm_layer)
        self.layer3 = self._stage(256, depths[2], 2, replace_stride_with_dilation[1], norm_layer)
        self.layer4 = self._stage(512, depths[3], 2, replace_stride_with_dilation[2], norm_layer)
        self.return_layers = dict(return_layers or {"layer4": "0"})
        for m in self.modules():  # torchvision's default initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _stage(self, planes, blocks, stride, dilate, norm_layer):
        prev_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                norm_layer(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, prev_dilation, downsample, norm_layer)]
        self.inplanes = planes * Bottleneck.expansion
        layers += [Bottleneck(self.inplanes, planes, 1, self.dilation, None, norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _stem_fusable(self, x):
        c, p = self.conv1, self.maxpool
        return (alo_hip.stem_conv_pool_supported(x, c.weight) and isinstance(self.bn1, FrozenBatchNorm2d) and c.bias is None
                and c.stride == (2, 2) and c.padding == (3, 3) and c.dilation == (1, 1) and c.groups == 1
                and isinstance(p, nn.MaxPool2d) and p.kernel_size == 3 and p.stride == 2 and p.padding == 1
                and p.dilation == 1 and not p.ceil_mode)

    def forward(self, x):
        out = {}
        # NHWC activations: MIOpen's bf16/fp32 implicit-GEMM kernels are NHWC-native; fed NCHW they wrap every
        # convolution in a pair of layout-transpose kernels
        if self._stem_fusable(x):
            # conv1 + bn1 + ReLU + max-pool in one kernel, straight from the NCHW image: the half-resolution 64-channel map
            # (the largest activation of the network) never reaches memory
            w, b = folded_conv_bn(self.conv1, self.bn1)
            x = alo_hip.stem_conv_pool(x, w, b)
        else:
            x = x.contiguous(memory_format=torch.channels_last)
            x = self.maxpool(conv_bn(x, self.conv1, self.bn1, relu=True))
        for name in ("layer1", "layer2", "layer3", "layer4"):
            x = getattr(self, name)(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def _resize_mask(mask, size):
    """torchvision ``resize`` of the float padding mask (bilinear, no antialias), as the reference applies before the
    ``.to(bool)`` cast (detr/backbone.py:127-128): any pixel touched by padding becomes padding."""
    return F.interpolate(mask, size=size, mode="bilinear", align_corners=False)


class BackboneBase(nn.Module):
    def __init__(self, backbone, train_backbone, num_channels, return_interm_layers, **kwargs):
        super().__init__()
        for name, parameter in backbone.named_parameters():
            if not train_backbone or ("layer2" not in name and "layer3" not in name and "layer4" not in name):
                parameter.requires_grad_(False)
        if return_interm_layers:
            backbone.return_layers = {"layer1": "0", "layer2": "1", "layer3": "2", "layer4": "3"}
        else:
            backbone.return_layers = {"layer4": "0"}
        self.body = backbone
        self.num_channels = num_channels

    def forward(self, frames, skip_masks=False, **kwargs):
        """``skip_masks``: the caller derives the per-level padding masks itself (alo_mask_pyramid); ``None`` is returned
        in their place."""
        xs = self.body(frames.as_tensor())
        if skip_masks:
            return {name: (x, None) for name, x in xs.items()}
        frame_masks = frames.mask.as_tensor()
        out = {}
        for name, x in xs.items():
            out[name] = (x, _resize_mask(frame_masks.float(), x.shape[-2:]).to(torch.bool))
        return out


class Backbone(BackboneBase):
    """ResNet backbone with frozen BatchNorm (weights are random unless a checkpoint is loaded: no download here)."""

    def __init__(self, name, train_backbone, return_interm_layers, dilation, **kwargs):
        body = ResNetBody(name, replace_stride_with_dilation=(False, False, dilation), norm_layer=FrozenBatchNorm2d)
        super().__init__(body, train_backbone, 2048, return_interm_layers, **kwargs)


class Joiner(nn.Sequential):
    """``[backbone, position_embedding]``; ``forward`` -> (list of (feature, mask), list of positional encodings)."""

    def __init__(self, backbone, position_embedding, tracing=None):
        super().__init__(backbone, position_embedding)

    @assert_and_export_onnx()
    def forward(self, frames, skip_pos_levels=(), **kwargs):
        """``skip_pos_levels``: indices of returned stages whose positional encoding the caller will not read (``None`` is
        returned in their place; the stride-4 map of Deformable-DETR costs as much as all the others together)."""
        out, pos = [], []
        for i, (_, x) in enumerate(self[0](frames, **kwargs).items()):
            out.append(x)
            pos.append(None if i in skip_pos_levels else self[1](x).to(x[0].dtype))
        return out, pos
