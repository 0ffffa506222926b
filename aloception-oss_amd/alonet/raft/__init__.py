from .corr import CorrBlock
from .raft import RAFT, RAFTBase

__all__ = ["CorrBlock", "RAFT", "RAFTBase"]
