from .ms_deform_attn import MSDeformAttn

__all__ = ["MSDeformAttn"]
