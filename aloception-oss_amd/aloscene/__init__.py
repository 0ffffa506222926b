"""aloscene — the slice of aloception's data model that the dense-vision hot path touches.

``Frame`` (normalisations, ``batch_list`` -> zero-padded batch + ``mask`` child, ``as_tensor``, ``to``), ``Mask``,
``Flow``, ``Labels`` and ``BoundingBoxes2D`` keep the constructor arguments and attribute names the reference's models
read (reference: aloscene/frame.py, aloscene/tensors/spatial_augmented_tensor.py:274-419).
Everything else of the reference package (3-D boxes, depth, disparity, camera calibration, renderer, file IO) is out
of scope: the hot path never reaches it.
"""
from .tensors import AugmentedTensor, SpatialAugmentedTensor
from .mask import Mask
from .flow import Flow
from .labels import Labels
from .bounding_boxes_2d import BoundingBoxes2D
from .frame import Frame

__all__ = ["AugmentedTensor", "SpatialAugmentedTensor", "Frame", "Mask", "Flow", "Labels", "BoundingBoxes2D"]
