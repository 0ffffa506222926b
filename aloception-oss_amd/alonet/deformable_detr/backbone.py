"""Backbone of Deformable-DETR: the DETR ResNet returning all four stages (reference: alonet/deformable_detr/backbone.py).

Stage ``layer1`` (stride 4) is returned for the panoptic head but ignored by the detector; the transformer consumes
strides 8/16/32 plus one extra stride-64 level made by ``DeformableDETR.input_proj[3]``.
"""
from alonet.detr.backbone import BackboneBase as _DetrBackboneBase
from alonet.detr.backbone import FrozenBatchNorm2d, ResNetBody
from alonet.detr.backbone import Joiner as _DetrJoiner


class BackboneBase(_DetrBackboneBase):
    def __init__(self, backbone, train_backbone, return_interm_layers, **kwargs):
        super().__init__(backbone, train_backbone, num_channels=2048, return_interm_layers=True, **kwargs)
        if return_interm_layers:
            backbone.return_layers = {"layer1": "0", "layer2": "1", "layer3": "2", "layer4": "3"}
            self.strides = [4, 8, 16, 32]
            self.num_channels = [256, 512, 1024, 2048]
        else:
            backbone.return_layers = {"layer4": "0"}
            self.strides = [32]
            self.num_channels = [2048]


class Backbone(BackboneBase):
    def __init__(self, name, train_backbone, return_interm_layers, dilation, **kwargs):
        assert name not in ("resnet18", "resnet34"), "number of channels are hard coded"
        body = ResNetBody(name, replace_stride_with_dilation=(False, False, dilation), norm_layer=FrozenBatchNorm2d)
        super().__init__(body, train_backbone, return_interm_layers, **kwargs)


class Joiner(_DetrJoiner):
    def __init__(self, backbone, position_embedding, tracing=None):
        super().__init__(backbone, position_embedding, tracing)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels
