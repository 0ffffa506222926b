"""``BoundingBoxes2D``: (N, 4) boxes with a format tag (``xcyc`` / ``xyxy`` / ``yxyx``) and relative/absolute flag."""
import torch

from .tensors import AugmentedTensor

_FORMATS = ("xcyc", "xyxy", "yxyx")


class BoundingBoxes2D(AugmentedTensor):
    @staticmethod
    def __new__(cls, x, boxes_format, absolute, labels=None, frame_size=None, names=("N", None), *args, **kwargs):
        if boxes_format not in _FORMATS:
            raise ValueError(f"boxes_format must be one of {_FORMATS}, got {boxes_format!r}")
        if absolute and frame_size is None:
            raise ValueError("absolute boxes need frame_size=(H, W)")
        obj = super().__new__(cls, x, *args, names=names, **kwargs)
        obj.add_property("boxes_format", boxes_format)
        obj.add_property("absolute", absolute)
        obj.add_property("frame_size", frame_size)
        obj.add_child("labels", labels)
        return obj

    @property
    def absolute(self):  # explicit: torch.Tensor.absolute (alias of abs) would shadow the stored property
        return self._props["absolute"]

    def _as(self, fmt):
        t = self.as_tensor()
        cur = self.boxes_format
        if cur == fmt:
            return t
        if cur == "xcyc":
            xc, yc, w, h = t.unbind(-1)
            xyxy = torch.stack([xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2], -1)
        elif cur == "yxyx":
            xyxy = t[..., [1, 0, 3, 2]]
        else:
            xyxy = t
        if fmt == "xyxy":
            return xyxy
        if fmt == "yxyx":
            return xyxy[..., [1, 0, 3, 2]]
        x1, y1, x2, y2 = xyxy.unbind(-1)
        return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)

    def _rewrap(self, data, **changes):
        out = data.as_subclass(type(self))._inherit(self)
        out._props.update(changes)
        return out

    def xcyc(self):
        return self._rewrap(self._as("xcyc"), boxes_format="xcyc")

    def xyxy(self):
        return self._rewrap(self._as("xyxy"), boxes_format="xyxy")

    def yxyx(self):
        return self._rewrap(self._as("yxyx"), boxes_format="yxyx")

    def _scale(self, size, inverse):
        h, w = size
        per_fmt = {"xcyc": (w, h, w, h), "xyxy": (w, h, w, h), "yxyx": (h, w, h, w)}[self.boxes_format]
        s = torch.tensor(per_fmt, dtype=self.dtype, device=self.device)
        return self.as_tensor() / s if inverse else self.as_tensor() * s

    def abs_pos(self, frame_size):
        """Absolute pixel coordinates for a frame of ``frame_size = (H, W)``."""
        if self.absolute:
            return self.rel_pos().abs_pos(frame_size) if tuple(self.frame_size) != tuple(frame_size) else self
        return self._rewrap(self._scale(frame_size, inverse=False), absolute=True, frame_size=tuple(frame_size))

    def rel_pos(self):
        if not self.absolute:
            return self
        return self._rewrap(self._scale(self.frame_size, inverse=True), absolute=False, frame_size=None)

    def area(self):
        x1, y1, x2, y2 = self._as("xyxy").unbind(-1)
        return (x2 - x1) * (y2 - y1)

    def remove_padding(self):
        """Boxes relative to the un-padded frame (identity here: boxes are stored relative to their own frame)."""
        return self

    def _pair(self, other):
        a = self._as("xyxy")
        if other.absolute != self.absolute:
            other = other.abs_pos(self.frame_size) if self.absolute else other.rel_pos()
        return a, other._as("xyxy")

    def iou_with(self, boxes2, ret_union=False):
        """Pairwise IoU matrix (N, M) with another set of boxes."""
        a, b = self._pair(boxes2)
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[None, :, :2])
        rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        union = area_a[:, None] + area_b[None, :] - inter
        iou = inter / union
        return (iou, union) if ret_union else iou

    def giou_with(self, boxes2):
        """Pairwise generalised IoU (https://giou.stanford.edu/) matrix (N, M)."""
        a, b = self._pair(boxes2)
        assert (a[:, 2:] >= a[:, :2]).all(), f"degenerate boxes {a}"
        assert (b[:, 2:] >= b[:, :2]).all(), f"degenerate boxes {b}"
        iou, union = self.iou_with(boxes2, ret_union=True)
        lt = torch.min(a[:, None, :2], b[None, :, :2])
        rb = torch.max(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        hull = wh[..., 0] * wh[..., 1]
        return iou - (hull - union) / hull
