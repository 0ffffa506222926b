from .weights import load_weights

__all__ = ["load_weights"]
