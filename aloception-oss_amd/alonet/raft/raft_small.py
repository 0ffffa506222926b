"""RAFT-small: bottleneck encoders (128-d features, 96 + 64 context), radius-3 correlation, plain ConvGRU, bilinear
up-sampling (no mask head).  Same class attributes and constructor as alonet/raft/raft_small.py:8-22; ``out_plane``
(2, a flow field) is stated here — the reference's class leaves that abstract attribute unset."""
from .extractor import SmallEncoder
from .raft import RAFTBase
from .update import SmallUpdateBlock


class RAFTSmall(RAFTBase):
    hidden_dim = 96
    context_dim = 64
    corr_levels = 4
    corr_radius = 3
    out_plane = 2

    def __init__(self, dropout=0, **kwargs):
        self.dropout = dropout
        fnet = self.build_fnet(encoder_cls=SmallEncoder, output_dim=128)
        cnet = self.build_cnet(encoder_cls=SmallEncoder)
        update_block = self.build_update_block(update_cls=SmallUpdateBlock)
        super().__init__(fnet, cnet, update_block, **kwargs)
