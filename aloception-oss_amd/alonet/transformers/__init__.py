from .mlp import MLP
from .position_encoding import PositionEmbeddingSine

__all__ = ["MLP", "PositionEmbeddingSine"]
