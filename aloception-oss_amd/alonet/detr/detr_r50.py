"""DETR with the ResNet-50 configuration of the paper (reference: alonet/detr/detr_r50.py:10-52)."""
from .backbone import Joiner
from .detr import Detr


class DetrR50(Detr):
    def __init__(self, *args, num_classes=91, background_class=91, **kwargs):
        position_embedding = self.build_positional_encoding(hidden_dim=256, position_embedding="sin")
        backbone = self.build_backbone("resnet50", train_backbone=True, return_interm_layers=True, dilation=False)
        num_channels = backbone.num_channels
        backbone = Joiner(backbone, position_embedding)
        backbone.num_channels = num_channels
        transformer = self.build_transformer(hidden_dim=256, dropout=0.1, nheads=8, dim_feedforward=2048,
                                             num_encoder_layers=6, num_decoder_layers=6, normalize_before=False)
        super().__init__(backbone, transformer, *args, num_classes=num_classes, num_queries=100,
                         background_class=background_class, **kwargs)
