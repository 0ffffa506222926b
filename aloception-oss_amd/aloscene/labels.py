"""``Labels``: per-object class labels with optional scores."""
from .tensors import AugmentedTensor


class Labels(AugmentedTensor):
    @staticmethod
    def __new__(cls, x, *args, encoding="id", labels_names=None, scores=None, names=("N",), **kwargs):
        if encoding not in ("id", "one-hot"):
            raise ValueError(f"unknown labels encoding {encoding!r}")
        obj = super().__new__(cls, x, *args, names=names, **kwargs)
        obj.add_property("encoding", encoding)
        obj.add_property("labels_names", labels_names)
        obj.add_child("scores", scores)
        return obj
