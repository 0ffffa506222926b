from .hip_graph import GraphedForward
from .weights import load_weights

__all__ = ["load_weights", "GraphedForward"]
