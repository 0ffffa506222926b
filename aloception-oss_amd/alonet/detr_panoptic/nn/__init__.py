from .MHAttention import MHAttentionMap
from .FPNstyle import FPNstyleCNN

__all__ = ["MHAttentionMap", "FPNstyleCNN"]
