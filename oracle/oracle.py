"""ORACLE — TEST INFRASTRUCTURE ONLY (ctypes front-end of oracle/libalo_oracle.so).

May be imported by tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py, and by
nothing under ``aloception-oss_amd/``.  Every function takes/returns numpy arrays; the C sources cite the
reference lines they restate (oracle/msda_oracle.c, oracle/corr_oracle.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libalo_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (``make -C oracle``)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _real(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", np.float64
    if dtype == np.float32:
        return "f32", np.float32
    raise TypeError(f"oracle computes in float32 or float64, got {dtype}")


def _msda_args(value, shapes, level_start, loc, attn):
    suf, dt = _real(value.dtype)
    value = np.ascontiguousarray(value, dt)
    loc = np.ascontiguousarray(loc, dt)
    attn = np.ascontiguousarray(attn, dt)
    shapes = np.ascontiguousarray(shapes, np.int32)
    level_start = np.ascontiguousarray(level_start, np.int32)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    assert shapes.shape == (L, 2) and level_start.shape == (L,) and attn.shape == (N, Lq, M, L, P)
    assert int((shapes[:, 0].astype(np.int64) * shapes[:, 1]).sum()) == S
    return suf, dt, value, shapes, level_start, loc, attn, (N, S, M, D, L, Lq, P)


def msda_forward(value, shapes, level_start, loc, attn):
    """-> out (N, Lq, M*D), dtype of ``value`` (float32 or float64)."""
    suf, dt, value, shapes, level_start, loc, attn, dims = _msda_args(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dt)
    getattr(lib(), f"oracle_msda_forward_{suf}")(
        _p(value), _p(shapes), _p(level_start), _p(loc), _p(attn), _p(out), *map(ctypes.c_int, dims)
    )
    return out


def msda_backward(value, shapes, level_start, loc, attn, grad_out):
    """-> (grad_value, grad_loc, grad_attn)."""
    suf, dt, value, shapes, level_start, loc, attn, dims = _msda_args(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dt).reshape(N, Lq, M * D)
    gv, gl, ga = np.empty_like(value), np.empty_like(loc), np.empty_like(attn)
    getattr(lib(), f"oracle_msda_backward_{suf}")(
        _p(value), _p(shapes), _p(level_start), _p(loc), _p(attn), _p(grad_out), _p(gv), _p(gl), _p(ga),
        *map(ctypes.c_int, dims),
    )
    return gv, gl, ga


def pyramid_shapes(H, W, num_levels=4):
    hw = [(H, W)]
    for _ in range(num_levels - 1):
        hw.append((hw[-1][0] // 2, hw[-1][1] // 2))
    return hw


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """fmaps (B,C,H,W) float32 -> list of ``num_levels`` arrays (B*H*W, 1, h_l, w_l) (corr.py:13-27)."""
    f1 = np.ascontiguousarray(fmap1, np.float32)
    f2 = np.ascontiguousarray(fmap2, np.float32)
    B, C, H, W = f1.shape
    HW = H * W
    lvl = np.empty((B * HW, 1, H, W), np.float32)
    lib().oracle_corr_volume(_p(f1), _p(f2), _p(lvl), B, C, HW)
    pyr = [lvl]
    for (h, w) in pyramid_shapes(H, W, num_levels)[1:]:
        prev = pyr[-1]
        nxt = np.empty((B * HW, 1, h, w), np.float32)
        lib().oracle_avg_pool2(_p(prev), _p(nxt), ctypes.c_long(B * HW), prev.shape[2], prev.shape[3])
        pyr.append(nxt)
    return pyr


def corr_lookup(pyr, coords, radius=4):
    """pyr from :func:`corr_pyramid`, coords (B,2,H,W) -> (B, L*(2r+1)^2, H, W) float32 (corr.py:29-50)."""
    coords = np.ascontiguousarray(coords, np.float32)
    B, _, H, W = coords.shape
    L = len(pyr)
    pyr = [np.ascontiguousarray(p, np.float32) for p in pyr]
    ptrs = (ctypes.c_void_p * L)(*[p.ctypes.data for p in pyr])
    hw = np.array([[p.shape[2], p.shape[3]] for p in pyr], np.int32)
    out = np.empty((B, L * (2 * radius + 1) ** 2, H, W), np.float32)
    lib().oracle_corr_lookup(ptrs, _p(hw), _p(coords), _p(out), B, H, W, radius, L)
    return out
