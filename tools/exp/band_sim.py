#!/usr/bin/env python
"""How often would a level-2 rolling band serve a WHOLE run of the resident forward (16 consecutive queries x 4 points of one head)?

VERDICT round 5, item 4: pyramids whose level 2 does not fit in LDS (1280 x 1920: 40 x 60 pixels x 64 B = 154 KB) could keep level 3
resident and a band of level-2 rows per workgroup (a horizontal stripe of the image + a halo).  The resident kernel chooses LDS or
buffer path per LEVEL GROUP of a run (msda.hip: 4 samples x 16 pairs go through one issue / consume sequence), so a band helps a run
only if EVERY one of its 64 level-2 samples stays inside the band.  This script counts that fraction on the four sampling
distributions of tools/kbench.py (numpy, CPU) for a 1280 x 1920 frame, stripes of H2 / 16 rows, halo rows 4 / 6 / 8 / 12, together
with the LDS a band of that height needs (level 3 resident beside it: 38.6 KB; 90 KB available)."""
import numpy as np

SHAPES = [(160, 240), (80, 120), (40, 60), (20, 30)]
SIGMA = (1.5, 2.0, 2.5, 3.0)
rng = np.random.default_rng(0)
M, P, L2 = 8, 4, 2
H2, W2 = SHAPES[L2]
stripes = 16


def offsets(kind, nq):
    ang = np.arange(M) * (2 * np.pi / M)
    ring = np.stack([np.cos(ang), np.sin(ang)], -1)
    ring = ring / np.abs(ring).max(-1, keepdims=True)
    mean = ring[:, None, :] * np.arange(1, P + 1)[None, :, None]            # (M, P, 2) pixels of the sampled level
    if kind == "ring":
        return mean[None] + rng.uniform(-0.5, 0.5, (nq, M, P, 2))
    if kind == "trained":
        t3 = rng.standard_t(3, (nq, M, P, 2)) / np.sqrt(3.0)
        return mean[None] + SIGMA[L2] * t3
    if kind == "survey":
        return rng.uniform(-0.05, 0.05, (nq, M, P, 2)) * np.array([W2, H2])
    return None


for kind in ("ring", "survey", "trained"):
    line = []
    for halo in (4, 6, 8, 12):
        inside_runs, runs = 0, 0
        for (h, w) in SHAPES:                       # queries of every level, in runs of 16 consecutive pixels
            ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
            ref_y = (ys.reshape(-1) + 0.5) / h * H2 - 0.5          # the query's own position in level-2 pixels
            nq = ref_y.size // 16 * 16
            off = offsets(kind, nq)
            y = ref_y[:nq, None, None] + off[..., 1]                # (nq, M, P) sample rows on level 2
            lo, hi = np.floor(y), np.floor(y) + 1
            stripe = (ys.reshape(-1)[:nq] * stripes // h)           # the workgroup that owns the query's stripe
            first = stripe.reshape(-1, 16)[:, :1].repeat(16, 1).reshape(-1)   # a run belongs to the stripe of its first query
            b0 = np.floor(first / stripes * H2) - halo
            b1 = np.ceil((first + 1) / stripes * H2) + halo
            ok = ((lo >= b0[:, None, None]) & (hi <= b1[:, None, None] - 1)) | (hi < 0) | (lo > H2 - 1)
            ok_run = ok.reshape(-1, 16, M, P).all(axis=(1, 3))     # (runs, M): all 64 samples of a (run, head)
            inside_runs += ok_run.sum()
            runs += ok_run.size
        rows = int(np.ceil(H2 / stripes)) + 2 * halo + 1
        line.append(f"halo {halo:2d}: {inside_runs / runs:6.1%} of the runs, band {rows} rows = {rows * (W2 * 64 + 8) / 1024:5.1f} KB")
    print(f"{kind:8s} " + " | ".join(line))
