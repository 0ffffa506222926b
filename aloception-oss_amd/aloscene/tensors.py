"""``AugmentedTensor``: a ``torch.Tensor`` subclass carrying dimension names, properties and child labels.

A compact re-statement of the behaviour the hot-path models rely on (reference:
aloscene/tensors/augmented_tensor.py:29-120,369-397,750-762): names are given at construction, children (``mask``,
``labels`` ...) and properties (``normalization`` ...) follow the tensor through ``to`` / ``clone`` / element-wise
math, and ``as_tensor()`` hands back a plain ``torch.Tensor``.  Dimension names are kept as a Python tuple rather than
torch's prototype named-tensor feature, which keeps every stock PyTorch-ROCm operator usable on the data.
"""
import numpy as np
import torch

_PASSTHROUGH = {"to", "clone", "detach", "contiguous", "float", "half", "bfloat16", "double", "cpu", "cuda",
                "add", "sub", "mul", "div", "true_divide", "neg", "__add__", "__sub__", "__mul__", "__truediv__",
                "__radd__", "__rsub__", "__rmul__", "__rtruediv__", "__neg__", "clamp", "type", "requires_grad_"}


class AugmentedTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, x, *args, names=None, device=None, dtype=None, **kwargs):
        if names is None:
            raise ValueError("AugmentedTensor needs `names=` (one name per dimension, e.g. ('C','H','W'))")
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        elif not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x)
        x = x.as_subclass(torch.Tensor)
        if device is not None or dtype is not None:
            x = x.to(device=device, dtype=dtype)
        if len(names) != x.dim():
            raise ValueError(f"{len(names)} names {tuple(names)} for a {x.dim()}-d tensor")
        obj = torch.Tensor._make_subclass(cls, x, x.requires_grad)
        obj._names = tuple(names)
        obj._props = {}
        obj._children = {}
        return obj

    def __init__(self, *args, **kwargs):
        super().__init__()

    # ---- metadata ---------------------------------------------------------------------------------------------------
    @property
    def names(self):
        return getattr(self, "_names", (None,) * self.dim())

    def add_property(self, name, value):
        self._props[name] = value

    def add_child(self, name, value):
        self._children[name] = value

    def __getattr__(self, name):  # only called when normal lookup fails
        for store in ("_props", "_children"):
            d = self.__dict__.get(store)
            if d is not None and name in d:
                return d[name]
        raise AttributeError(f"{type(self).__name__} has no attribute {name!r}")

    def __setattr__(self, name, value):
        d = self.__dict__
        if "_props" in d and name in d["_props"]:
            d["_props"][name] = value
        elif "_children" in d and name in d["_children"]:
            d["_children"][name] = value
        else:
            super().__setattr__(name, value)

    def _inherit(self, other, names=None):
        """Copy names / properties / children of ``other`` onto this (freshly wrapped) tensor."""
        self._names = tuple(names) if names is not None else other.names
        self._props = dict(getattr(other, "_props", {}))
        self._children = dict(getattr(other, "_children", {}))
        return self

    def as_tensor(self):
        """Plain ``torch.Tensor`` view of the data (names and children dropped)."""
        return self.as_subclass(torch.Tensor)

    # ---- propagation through torch ops --------------------------------------------------------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        name = getattr(func, "__name__", "")
        src = next((a for a in args if isinstance(a, AugmentedTensor)), None)
        if src is None or name not in _PASSTHROUGH or not isinstance(out, torch.Tensor) or out.shape != src.shape:
            if isinstance(out, AugmentedTensor):
                out = out.as_subclass(torch.Tensor)
            return out
        res = out.as_subclass(type(src))
        res._inherit(src)
        if name in ("to", "cpu", "cuda"):  # children travel with the data (reference :369-397)
            moved = {}
            def move(c):
                if not isinstance(c, torch.Tensor):
                    return c
                if name == "to":
                    dev_args = [a for a in args[1:] if isinstance(a, (str, torch.device))]
                    dev = kwargs.get("device", dev_args[0] if dev_args else None)
                    return c.to(dev) if dev is not None else c
                return getattr(c, name)()

            for key, child in res._children.items():
                if isinstance(child, (list, tuple)):
                    moved[key] = [move(c) for c in child]
                elif isinstance(child, torch.Tensor) and name == "to":
                    dev_args = [a for a in args[1:] if isinstance(a, (str, torch.device))]
                    dev = kwargs.get("device", dev_args[0] if dev_args else None)
                    moved[key] = child.to(dev) if dev is not None else child
                elif isinstance(child, torch.Tensor):
                    moved[key] = getattr(child, name)()
                else:
                    moved[key] = child
            res._children = moved
        return res

    def __repr__(self):
        return f"{type(self).__name__}(names={self.names}, shape={tuple(self.shape)}, dtype={self.dtype}, device={self.device})"


class SpatialAugmentedTensor(AugmentedTensor):
    """An augmented tensor with "H" and "W" dimensions (and optionally "B", "C")."""

    @staticmethod
    def __new__(cls, x, *args, names=None, **kwargs):
        obj = super().__new__(cls, x, *args, names=names, **kwargs)
        obj.add_child("mask", None)
        return obj

    @property
    def H(self):
        return self.shape[self.names.index("H")]

    @property
    def W(self):
        return self.shape[self.names.index("W")]

    @property
    def HW(self):
        return (self.H, self.W)

    def append_mask(self, mask):
        self._children["mask"] = mask

    def _pad_fill(self, shape):
        """A plain tensor of ``shape`` holding what padding this tensor writes (zeros here; ``Frame`` overrides it)."""
        return torch.zeros(shape, dtype=self.dtype, device=self.device)

    def batch(self, dim=0):
        """Add a leading "B" dimension (no-op when the tensor already has one)."""
        if "B" in self.names:
            return self
        out = self.as_tensor().unsqueeze(dim).as_subclass(type(self))
        names = list(self.names)
        names.insert(dim, "B")
        out._inherit(self, names)
        if isinstance(out._children.get("mask"), SpatialAugmentedTensor):
            out._children["mask"] = out._children["mask"].batch(dim)
        return out

    @staticmethod
    def batch_list(tensors):
        """Stack spatial tensors of possibly different sizes into one padded batch.

        The result has size (max H, max W) and a ``mask`` child (a ``Mask`` with a single channel, float32) holding 1 on the
        padded area and 0 on real pixels — reference: spatial_augmented_tensor.py:322-419.  The padding carries the value
        the tensor's own ``pad`` would write (``_pad_fill``): 0 for plain tensors, and for a ``Frame`` the value of a black
        pixel in its normalisation (frame.py:555-600; pinned on the reference's Frame by G16).
        """
        from .mask import Mask

        tensors = [t for t in tensors if t is not None]
        assert len(tensors) >= 1
        batched = [t.batch() for t in tensors]
        first = batched[0]
        names = first.names
        ih, iw, ic = names.index("H"), names.index("W"), names.index("C")
        max_h = max(t.H for t in batched)
        max_w = max(t.W for t in batched)
        shape = list(first.shape)
        shape[0], shape[ih], shape[iw] = len(batched), max_h, max_w
        mshape = list(shape)
        mshape[ic] = 1
        data = first._pad_fill(shape)
        mask = torch.ones(mshape, dtype=torch.float32, device=first.device)
        for b, t in enumerate(batched):
            region = [slice(None)] * len(shape)
            region[0], region[ih], region[iw] = b, slice(0, t.H), slice(0, t.W)
            sub = list(region)
            del sub[0]
            data[tuple(region)] = t.as_tensor()[0]
            mregion = list(region)
            mregion[ic] = slice(None)
            mask[tuple(mregion)] = 0
        out = data.as_subclass(type(first))
        out._inherit(first, names)
        # per-frame labels (boxes2d, labels, ...) become one entry per batch item, as in the reference
        for key in list(out._children):
            if key == "mask":
                continue
            per_item = [t._children.get(key) for t in batched]
            out._children[key] = per_item if any(c is not None for c in per_item) else None
        out._children["mask"] = Mask(mask, names=names)
        return out
