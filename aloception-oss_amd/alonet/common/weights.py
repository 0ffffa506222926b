"""Checkpoint loading with the reference's state-dict conventions (alonet/common/weights.py:34-97).

``.pth`` files hold either ``{"model": state_dict}`` or a bare state dict; Lightning ``.ckpt`` files hold
``{"state_dict": ...}`` whose keys carry a leading ``model.``.  Named weights (``"deformable-detr-r50"``,
``"raft-things"`` ...) are looked up in ``~/.aloception/weights/<name>/<name>.pth``; the reference downloads them on
first use, which this offline build does not do.
"""
import os

import torch


def _resolve(weights):
    if weights.endswith(".pth") or weights.endswith(".ckpt"):
        return weights
    path = os.path.join(os.path.expanduser("~"), ".aloception", "weights", weights, weights + ".pth")
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"named weights '{weights}' not found at {path}; place the checkpoint there (no download in this build)")
    return path


def load_weights(model, weights, device=None, strict_load_weights=True):
    path = _resolve(weights)
    blob = torch.load(path, map_location=device or "cpu")
    if path.endswith(".ckpt"):
        state = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in blob["state_dict"].items()}
    else:
        state = blob["model"] if isinstance(blob, dict) and "model" in blob else blob
    model.load_state_dict(state, strict=strict_load_weights)
    import alo_hip

    alo_hip.invalidate_caches(model)  # packed / folded / merged inference-time copies of the old weights
    print(f"Weights loaded from {path}")
    return model
