"""``Mask``: binary / float spatial mask (padding masks of batched frames, segmentation outputs)."""
from .tensors import SpatialAugmentedTensor


class Mask(SpatialAugmentedTensor):
    @staticmethod
    def __new__(cls, x, *args, names=("N", "H", "W"), labels=None, **kwargs):
        obj = super().__new__(cls, x, *args, names=names, **kwargs)
        obj.add_child("labels", labels)
        return obj
