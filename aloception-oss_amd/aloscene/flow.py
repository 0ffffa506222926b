"""``Flow``: dense optical flow field, channel 0 = horizontal, 1 = vertical displacement in pixels."""
from .tensors import SpatialAugmentedTensor


class Flow(SpatialAugmentedTensor):
    @staticmethod
    def __new__(cls, x, *args, names=("C", "H", "W"), occlusion=None, **kwargs):
        obj = super().__new__(cls, x, *args, names=names, **kwargs)
        obj.add_child("occlusion", occlusion)
        return obj
