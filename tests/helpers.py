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
