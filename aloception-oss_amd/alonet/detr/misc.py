"""Input contract of the DETR-family ``forward`` methods (reference: alonet/detr/misc.py:9-33)."""
from functools import wraps

from aloscene import Frame


def assert_and_export_onnx(check_mean_std=False, input_mean_std=None):
    """Decorator: a list of Frames is batched with ``Frame.batch_list``; the batch must be a resnet-normalised
    ("B","C","H","W") Frame carrying a padding mask.  The reference's ONNX/TorchScript tracing mode (plain tensors)
    is a TensorRT-export feature and is not part of this build."""

    def decorator(forward):
        @wraps(forward)
        def wrapper(instance, frames, *args, **kwargs):
            if isinstance(frames, list):
                frames = Frame.batch_list(frames)
            assert isinstance(frames, Frame), "expected an aloscene.Frame (or a list of Frames)"
            assert frames.normalization == "resnet"
            assert frames.names == ("B", "C", "H", "W")
            assert frames.mask is not None
            assert frames.mask.names == ("B", "C", "H", "W")
            if check_mean_std and input_mean_std is not None:
                assert tuple(frames.mean_std[0]) == tuple(input_mean_std[0])
                assert tuple(frames.mean_std[1]) == tuple(input_mean_std[1])
            return forward(instance, frames, *args, **kwargs)

        return wrapper

    return decorator
