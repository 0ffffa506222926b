from .ms_deform_attn_func import (
    MSDeformAttnFunction,
    bilinear_grid_sample,
    load_MultiScaleDeformableAttention,
    load_ops,
    ms_deform_attn_core_pytorch,
)

__all__ = ["MSDeformAttnFunction", "ms_deform_attn_core_pytorch", "bilinear_grid_sample",
           "load_MultiScaleDeformableAttention", "load_ops"]
