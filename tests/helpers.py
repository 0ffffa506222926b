"""Shared generators for the parity tests (seeded, numpy)."""
import numpy as np


def level_start(shapes):
    shapes = np.asarray(shapes, np.int64)
    return np.concatenate([[0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1]]).astype(np.int32)


def msda_case(seed, N, M, D, Lq, shapes, P, dtype=np.float32, loc_range=(-0.1, 1.1)):
    rng = np.random.default_rng(seed)
    shapes = np.asarray(shapes, np.int32)
    L = len(shapes)
    S = int((shapes[:, 0].astype(np.int64) * shapes[:, 1]).sum())
    value = rng.standard_normal((N, S, M, D)).astype(dtype)
    loc = rng.uniform(loc_range[0], loc_range[1], (N, Lq, M, L, P, 2)).astype(dtype)
    logits = rng.standard_normal((N, Lq, M, L * P))
    attn = np.exp(logits - logits.max(-1, keepdims=True))
    attn = (attn / attn.sum(-1, keepdims=True)).reshape(N, Lq, M, L, P).astype(dtype)
    grad_out = rng.standard_normal((N, Lq, M * D)).astype(dtype)
    return dict(value=value, shapes=shapes, level_start=level_start(shapes), loc=loc, attn=attn, grad_out=grad_out)


DETR_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]  # 800 x 1333 frame, strides 8/16/32/64  (S = 22223)


def formula_state_dict(state_dict, seed=0):
    """Deterministic weights derived from each tensor's NAME and SHAPE only.

    Applied to the reference's module (when the golden vectors are generated) and to ours (in the tests): both models
    then hold identical parameters without a multi-megabyte checkpoint in the repository, and independently of the
    order in which either implementation constructs its sub-modules.
    """
    import zlib

    import torch

    out = {}
    for key in sorted(state_dict):
        ref = state_dict[key]
        if not ref.dtype.is_floating_point:
            out[key] = ref.clone()
            continue
        gen = torch.Generator().manual_seed((zlib.crc32(key.encode()) + seed) % (2 ** 31))
        noise = torch.randn(ref.shape, generator=gen, dtype=torch.float64)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "running_var":
            val = 1.0 + 0.1 * noise.abs()
        elif leaf == "running_mean":
            val = 0.05 * noise
        elif ref.dim() <= 1:
            is_norm_scale = leaf == "weight"
            val = (1.0 + 0.1 * noise) if is_norm_scale else 0.05 * noise
        else:
            fan_in = ref[0].numel()
            val = noise * (1.0 / fan_in) ** 0.5
        out[key] = val.to(ref.dtype)
    return out


def stub_pyramid(channels=(8, 12, 16, 24)):
    """A seeded four-stage convolution pyramid (strides 4 / 8 / 16 / 32, ceil sizes like a ResNet's) in the role of the ResNet
    body: ``forward(x) -> OrderedDict {"0": stride 4, ..., "3": stride 32}``.  The model-level fixtures (G14 / G15,
    tests/golden/make_golden_models.py) put the SAME module under the reference's ``BackboneBase`` / ``Joiner`` and under this
    repository's, so everything above the convolution stack is compared with the reference: mask resize, positional encodings,
    projections, transformer, heads, ``inference()``."""
    from collections import OrderedDict

    import torch.nn.functional as F
    from torch import nn

    class StubPyramid(nn.Module):
        def __init__(self):
            super().__init__()
            c0, c1, c2, c3 = channels
            self.layer1 = nn.Sequential(nn.Conv2d(3, c0, 3, stride=2, padding=1), nn.ReLU(), nn.Conv2d(c0, c0, 3, stride=2, padding=1))
            self.layer2 = nn.Conv2d(c0, c1, 3, stride=2, padding=1)
            self.layer3 = nn.Conv2d(c1, c2, 3, stride=2, padding=1)
            self.layer4 = nn.Conv2d(c2, c3, 3, stride=2, padding=1)

        def forward(self, x):
            out = OrderedDict()
            for i, name in enumerate(("layer1", "layer2", "layer3", "layer4")):
                x = F.relu(getattr(self, name)(x))
                out[str(i)] = x
            return out

    return StubPyramid()


def tied_formula_state_dict(model, seed=0, scale=None):
    """``formula_state_dict`` of ``model`` where keys that alias ONE parameter (Deformable-DETR's shared detection heads:
    ``class_embed.0 … .5`` are the same module) all carry the values of the alphabetically first alias, so the loaded weights do not depend on the
    order in which ``load_state_dict`` walks the aliases.  ``scale``: {key suffix: factor} applied on top (G15 widens
    ``query_embed.weight`` so that the queries of the vanilla DETR decoder do not come out as near-copies of each other)."""
    sd = formula_state_dict(model.state_dict(), seed)
    for suffix, factor in (scale or {}).items():
        for key in sd:
            if key.endswith(suffix):
                sd[key] = sd[key] * factor
    groups = {}
    for key, ref in model.state_dict(keep_vars=True).items():
        if ref.numel():
            groups.setdefault((ref.data_ptr(), tuple(ref.shape)), []).append(key)
    for keys in groups.values():
        owner = min(keys)                      # independent of the order in which either implementation registers its modules
        for key in keys:
            sd[key] = sd[owner].clone()
    return sd


def g17_inputs(g):
    """Re-draws the inputs of the G17 fixture (the reference's TensorRT-plugin test case for the MSDA kernel: 24 MB of uniform noise,
    kept as a seed) with torch's CPU generator, in the generator's call order, and checks them against the fixture's sha256 digests.
    Returns (value, loc, attn) as float32 tensors, or None when this torch build draws a different stream."""
    import hashlib

    import torch

    N, M, D, Lq, L, P = (int(v) for v in g["dims"])
    S = int(np.prod(g["shapes"].astype(np.int64), axis=1).sum())
    gen = torch.Generator().manual_seed(int(g["seed"]))
    value = torch.rand(N, S, M, D, generator=gen)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=gen)
    attn = torch.rand(N, Lq, M, L, P, generator=gen) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    for t, key in ((value, "sha_value"), (loc, "sha_loc"), (attn, "sha_attn")):
        if hashlib.sha256(t.numpy().tobytes()).digest() != g[key].tobytes():
            return None
    return value, loc, attn
