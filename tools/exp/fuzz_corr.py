"""Randomised differential run of alo_corr_build / alo_corr_lookup against the C oracle, and of alo_corr_lookup_backward against
autograd through the torch formulation (a development tool; the committed parity
tests are in tests/).  Random batch sizes, channel counts, map sizes (odd ones too), magnitudes, level counts, radii, coordinate spreads.

    python tools/exp/fuzz_corr.py [--seconds 120] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("aloception-oss_amd", "oracle", "tests"):  # noqa: the product package for the kernels AND the torch formulation
    sys.path.insert(0, os.path.join(ROOT, p))
import alo_hip  # noqa: E402
import oracle as O  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def run_case(rng):
    L = int(rng.integers(1, 5))
    lo = 2 ** L   # the coarsest level must be at least 2 x 2 (alo_corr_lookup rejects 1-pixel levels)
    H, W = int(rng.integers(lo, 49)), int(rng.integers(lo, 65))
    B, C = int(rng.integers(1, 4)), int(rng.choice([3, 8, 16, 37, 64, 128, 256]))
    r = int(rng.integers(0, 6))
    sa, sb = 10.0 ** rng.uniform(-6, 6), 10.0 ** rng.uniform(-6, 6)
    f1 = (rng.standard_normal((B, C, H, W)) * sa).astype(np.float32)
    f2 = (rng.standard_normal((B, C, H, W)) * sb).astype(np.float32)
    ref_pyr = O.corr_pyramid(f1, f2, L)
    levels = alo_hip.corr_build(dev(f1), dev(f2), L)
    # fp32-class accuracy relative to the size of the sums: |f1|.|f2| per entry / sqrt(C)
    scale = float(np.sqrt((f1.astype(np.float64) ** 2).sum(1).max()) * np.sqrt((f2.astype(np.float64) ** 2).sum(1).max()) / np.sqrt(C))
    for lvl in range(L):
        got = levels[lvl].cpu().numpy()
        if got.shape != ref_pyr[lvl].shape:
            return f"level {lvl} shape {got.shape} vs {ref_pyr[lvl].shape}", (B, C, H, W, L, r)
        d = np.abs(got - ref_pyr[lvl]).max()
        if not d <= 4e-6 * scale:
            return f"build level {lvl}: {d:.4g} at scale {scale:.4g}", (B, C, H, W, L, r, sa, sb)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    grid = np.broadcast_to(np.stack([xs, ys], 0)[None].astype(np.float32), (B, 2, H, W))
    spread = float(rng.choice([0.0, 0.7, 3.0, 20.0, 200.0]))
    coords = (grid + rng.standard_normal(grid.shape) * spread).astype(np.float32)
    out = alo_hip.corr_lookup(levels, dev(coords), r).cpu().numpy()
    ref = O.corr_lookup([lv.cpu().numpy() for lv in levels], coords, r)
    vmax = max(float(np.abs(ref).max()), 1e-30)
    d = np.abs(out - ref).max()
    if not d <= 4e-6 * max(vmax, float(np.abs(levels[0].cpu().numpy()).max())):
        return f"lookup: {d:.4g} of {vmax:.4g} (spread {spread})", (B, C, H, W, L, r)
    # the lookup's adjoint: <lookup(P), G> == <P, lookup_backward(G)> on a unit-scale pyramid, and against autograd through the torch
    # formulation of the reference's block (bilinear_sampler = grid_sample)
    from alonet.raft.corr import lookup_torch

    pyr = [torch.randn_like(lv) for lv in levels]
    gout = torch.randn(B, L * (2 * r + 1) ** 2, H, W, device=DEV)
    cdev = dev(coords)
    grads = alo_hip.corr_lookup_backward([torch.zeros_like(p) for p in pyr], cdev, gout, r)
    lhs = float((alo_hip.corr_lookup(pyr, cdev, r).double() * gout.double()).sum())
    rhs = float(sum((p.double() * g.double()).sum() for p, g in zip(pyr, grads)))
    if not abs(lhs - rhs) <= 2e-4 * max(1.0, abs(lhs), float(gout.numel()) ** 0.5):
        return f"adjoint identity: {lhs:.6g} vs {rhs:.6g} (spread {spread})", (B, C, H, W, L, r)
    leaves = [p.clone().requires_grad_(True) for p in pyr]
    cleaf = cdev.clone().requires_grad_(True)
    *want, want_c = torch.autograd.grad(lookup_torch(leaves, cleaf, r), leaves + [cleaf], gout)
    got_c = alo_hip.corr_lookup_backward_coords(pyr, cdev, gout, r)
    safe = torch.isfinite(want_c)
    for lvl in range(L):   # away from the interpolant's kinks (integer positions on any level)
        pos = cdev / 2 ** lvl
        safe &= ~((pos - pos.round()).abs() < 2e-3 * torch.clamp(pos.abs(), min=1.0)).any(dim=1, keepdim=True)
    d = float(((got_c - torch.nan_to_num(want_c)).abs() * safe).max())
    if not d <= 2e-4 * max(1.0, float((torch.nan_to_num(want_c).abs() * safe).max())):
        return f"coordinate gradient: {d:.4g} (spread {spread})", (B, C, H, W, L, r)
    for lvl, (g, w_) in enumerate(zip(grads, want)):
        d = float((g - w_).abs().max())
        if not d <= 3e-5 * max(1.0, float(w_.abs().max())):
            return f"lookup backward level {lvl}: {d:.4g} (spread {spread})", (B, C, H, W, L, r)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    O.build()
    rng = np.random.default_rng(a.seed)
    t0, n = time.time(), 0
    while time.time() - t0 < a.seconds:
        msg, what = run_case(rng)
        n += 1
        if msg:
            print(f"FAIL case {n}: {msg}; {what}")
            return 1
    print(f"{n} cases, no failure")
    return 0


if __name__ == "__main__":
    sys.exit(main())
