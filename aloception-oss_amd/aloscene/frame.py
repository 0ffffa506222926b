"""``Frame``: an image (or batch of images) with its normalisation state and attached labels.

Constructor and normalisation API follow the reference (aloscene/frame.py:91-142,386-548): accepted normalisations
are ``"255"``, ``"01"``, ``"minmax_sym"`` ([-1, 1]) and ``"resnet"`` (ImageNet mean/std); every ``norm_*`` converts
from whatever the current state is.  ``Frame.batch_list`` pads to the largest frame and attaches the padding ``mask``
(spatial_augmented_tensor.py:322-419).  Loading from a file path is not supported here (no image codecs on the path).
"""
import torch

from .tensors import SpatialAugmentedTensor

_RESNET_MEAN_STD = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
_CHILDREN = ("boxes2d", "boxes3d", "labels", "flow", "segmentation", "disparity", "points2d", "points3d", "depth")


class Frame(SpatialAugmentedTensor):
    @staticmethod
    def __new__(cls, x, *args, normalization="255", mean_std=None, names=("C", "H", "W"), **kwargs):
        if isinstance(x, str):
            raise NotImplementedError("loading a Frame from a file path is outside the hot-path scope; pass a tensor")
        children = {k: kwargs.pop(k, None) for k in _CHILDREN}
        if normalization not in {"01", "255", "minmax_sym", "resnet"}:
            raise AssertionError(f"{normalization} norm is not yet supported")
        obj = super().__new__(cls, x, *args, names=names, **kwargs)
        for key, child in children.items():
            obj.add_child(key, child)
        obj.add_property("normalization", normalization)
        obj.add_property("_resnet_mean_std", _RESNET_MEAN_STD)
        obj.add_property("mean_std", _RESNET_MEAN_STD if normalization == "resnet" else mean_std)
        return obj

    # ---- normalisation ----------------------------------------------------------------------------------------------
    def _channel_stats(self, mean_std):
        shape = [1] * self.dim()
        shape[self.names.index("C")] = 3
        mean = torch.tensor(mean_std[0], device=self.device, dtype=torch.float32).view(shape)
        std = torch.tensor(mean_std[1], device=self.device, dtype=torch.float32).view(shape)
        return mean, std

    def _pad_fill(self, shape):
        """Padding of ``batch_list`` / ``pad`` (reference frame.py:555-600): a BLACK pixel in the frame's own normalisation —
        0 for "01" / "255", -1 for "minmax_sym", ``(0 - mean) / std`` for a mean/std state such as "resnet"."""
        fill = torch.zeros(shape, dtype=self.dtype, device=self.device)
        if self.normalization == "minmax_sym":
            return fill - 1
        if self.normalization not in ("01", "255") and self.mean_std is not None:
            view = [1] * len(shape)
            view[self.names.index("C") + (len(shape) - self.dim())] = 3
            mean = torch.tensor(self.mean_std[0], device=self.device).view(view)
            std = torch.tensor(self.mean_std[1], device=self.device).view(view)
            return (fill - mean) / std
        return fill

    def _restate(self, data, normalization, mean_std=None):
        out = data.as_subclass(Frame)._inherit(self)
        out._props["normalization"] = normalization
        out._props["mean_std"] = mean_std
        return out

    def _to_01(self):
        t = self.as_tensor()
        if self.normalization == "01":
            return t.clone()
        if self.normalization == "255":
            return t.div(255)
        if self.normalization == "minmax_sym":
            return (t + 1.0) / 2.0
        if self.mean_std is not None:
            mean, std = self._channel_stats(self.mean_std)
            return t * std + mean
        raise Exception(f"Can't convert from {self.normalization} to norm01")

    def norm01(self):
        return self._restate(self._to_01(), "01")

    def norm255(self):
        if self.normalization == "255":
            return self._restate(self.as_tensor().clone(), "255")
        if self.normalization == "minmax_sym":
            return self._restate((self.as_tensor() + 1.0) * 255.0 / 2.0, "255")
        return self._restate(self._to_01().mul(255), "255")

    def norm_minmax_sym(self):
        if self.normalization == "minmax_sym":
            return self._restate(self.as_tensor().clone(), "minmax_sym")
        if self.normalization == "255":
            return self._restate(2 * (self.as_tensor() / 255.0) - 1.0, "minmax_sym")
        return self._restate(2 * self._to_01() - 1.0, "minmax_sym")

    def mean_std_norm(self, mean, std, name):
        if self.mean_std is not None and tuple(self.mean_std[0]) == tuple(mean) and tuple(self.mean_std[1]) == tuple(std):
            return self._restate(self.as_tensor().clone(), name, (mean, std))
        m, s = self._channel_stats((mean, std))
        return self._restate((self._to_01() - m) / s, name, (mean, std))

    def norm_resnet(self):
        return self.mean_std_norm(_RESNET_MEAN_STD[0], _RESNET_MEAN_STD[1], "resnet")

    def norm_as(self, target):
        if target.normalization == "01":
            return self.norm01()
        if target.normalization == "255":
            return self.norm255()
        if target.normalization == "minmax_sym":
            return self.norm_minmax_sym()
        if target.mean_std is not None:
            return self.mean_std_norm(target.mean_std[0], target.mean_std[1], target.normalization)
        raise Exception(f"Can't convert the tensor normalization to the target normalization: {target.normalization}")
