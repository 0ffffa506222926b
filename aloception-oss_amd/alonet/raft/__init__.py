from .corr import CorrBlock
from .raft import RAFT, RAFTBase
from .raft_small import RAFTSmall

__all__ = ["CorrBlock", "RAFT", "RAFTBase", "RAFTSmall"]
