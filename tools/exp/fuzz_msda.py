"""Randomised differential run of the MSDA entry points against the C oracle (a development tool; the committed parity tests are in
tests/).  Random pyramids, batch sizes, query counts, location spreads (inside, on and far outside the maps), masks, reference points.

    python tools/exp/fuzz_msda.py [--seconds 120] [--seed 0]

Checks per case: (1) generic forward fp32 vs oracle; (2) fused head-major bf16 forward, plain vs LDS-resident (ALWAYS) bit-equal and
vs the oracle on the bf16-rounded inputs; (3) backward fp32 vs oracle (the wide kernel for encoder-shaped launches, 4x4 tiles otherwise);
(4) backward with bf16 values vs fp32 on the rounded operands; (5) backward at D = 64 vs oracle.  Prints the first failure.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("aloception-oss_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import alo_hip  # noqa: E402
import oracle as O  # noqa: E402
from helpers import level_start  # noqa: E402

DEV = "cuda:0"


def dev(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x.to(dtype) if dtype is not None else x


def random_pyramid(rng):
    big = rng.random() < 0.08   # now and then a pyramid of detection size (windows spanning several passes, AUTO-resident launches)
    h, w = (int(rng.integers(70, 130)), int(rng.integers(90, 180))) if big else (int(rng.integers(6, 70)), int(rng.integers(6, 90)))
    out = []
    for _ in range(4):
        out.append((h, w))
        h, w = max(1, (h + int(rng.integers(0, 2))) // 2), max(1, (w + int(rng.integers(0, 2))) // 2)
    return out


def case(rng):
    shapes_l = random_pyramid(rng)
    N = int(rng.integers(1, 4))
    S = sum(h * w for h, w in shapes_l)
    encoder = rng.random() < 0.6
    Lq = S if encoder else int(rng.integers(1, 700))
    M, D, L, P = 8, 32, 4, 4
    spread = float(rng.choice([0.5, 2.0, 4.0, 9.0, 30.0]))
    refs = []
    for (h, w) in shapes_l:
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        refs.append(np.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1))
    ref_grid = np.concatenate(refs, 0)
    if encoder:
        ref = np.broadcast_to(ref_grid[None, :, None, :], (N, S, L, 2)).copy()
    else:
        ref = rng.uniform(-0.15, 1.15, (N, Lq, L, 2))
    ref = ref.astype(np.float32)
    offsets = (rng.standard_normal((N, Lq, M, L, P, 2)) * spread).astype(np.float32)
    logits = (rng.standard_normal((N, Lq, M, L * P)) * 2).astype(np.float32)
    value = rng.standard_normal((N, S, M, D)).astype(np.float32)
    mask = rng.random((N, S)) < (0.15 if rng.random() < 0.5 else 0.0)
    go = rng.standard_normal((N, Lq, M * D)).astype(np.float32)
    return dict(shapes_l=shapes_l, N=N, S=S, Lq=Lq, ref=ref, offsets=offsets, logits=logits, value=value, mask=mask, go=go, spread=spread,
                encoder=encoder)


def run_case(c):
    shapes_l, N, S, Lq = c["shapes_l"], c["N"], c["S"], c["Lq"]
    shapes_np = np.asarray(shapes_l, np.int32)
    start_np = level_start(shapes_np)
    shapes = dev(shapes_np)
    shapes._alo_shapes = [tuple(hw) for hw in shapes_l]
    start = dev(start_np)
    # --- bf16 fused head-major: plain vs resident, vs oracle on the rounded inputs
    vb = dev(c["value"]).bfloat16()
    ob, lb = dev(c["offsets"]).bfloat16(), dev(c["logits"]).bfloat16()
    ref = dev(c["ref"])
    mask = dev(c["mask"])
    vhm = alo_hip.value_head_major(vb, mask)
    plain = alo_hip.msda_forward_fused_hm(vhm, shapes, start, ob, lb, ref, resident=False)
    res = alo_hip.msda_forward_fused_hm(vhm, shapes, start, ob, lb, ref, resident="always")
    if not torch.equal(plain, res):
        return "resident != plain"
    if not torch.equal(plain, alo_hip.msda_forward_fused_hm(vhm, shapes, start, ob, lb, ref)):
        return "resident (auto) != plain"
    attn = torch.softmax(lb.float(), -1).view(N, Lq, 8, 4, 4)
    normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1).float()
    loc = ref[:, :, None, :, None, :] + ob.float() / normalizer[None, None, None, :, None, :]
    vmasked = vb.float().masked_fill(mask[..., None, None], 0)
    exact = O.msda_forward(vmasked.double().cpu().numpy(), shapes_np, start_np, loc.double().cpu().numpy(), attn.double().cpu().numpy())
    err = np.abs(plain.double().cpu().numpy() - exact)
    # locations are computed in fp32 by the kernel and in fp32 here: samples within 1e-4 px of a pixel edge may flip a corner
    if not np.all(err <= np.abs(exact) * 2.0 ** -8 + 1e-4):
        return f"bf16 forward vs oracle: max err {err.max():.4g}"
    # --- fp32 generic forward + backward vs oracle
    loc32, attn32 = loc.contiguous(), attn.contiguous()
    v32 = vmasked.contiguous()
    out = alo_hip.msda_forward(v32, shapes, start, loc32, attn32, 64)
    err = np.abs(out.double().cpu().numpy() - exact)
    if not np.all(err <= 2e-4 * max(1.0, np.abs(exact).max())):
        return f"fp32 forward vs oracle: max err {err.max():.4g}"
    go = dev(c["go"])
    gv, gl, ga = (x.cpu().numpy() for x in alo_hip.msda_backward(v32, shapes, start, loc32, attn32, go, 64))
    rgv, rgl, rga = O.msda_backward(v32.double().cpu().numpy(), shapes_np, start_np, loc32.double().cpu().numpy(),
                                    attn32.double().cpu().numpy(), go.double().cpu().numpy())
    if not np.abs(gv - rgv).max() <= 3e-4 * max(1.0, np.abs(rgv).max()):
        return f"grad_value: {np.abs(gv - rgv).max():.4g} of {np.abs(rgv).max():.4g}"
    if not np.abs(ga - rga).max() <= 3e-4 * max(1.0, np.abs(rga).max()):
        return f"grad_attn: {np.abs(ga - rga).max():.4g} of {np.abs(rga).max():.4g}"
    # grad_loc is discontinuous at pixel edges: ignore samples within 1e-3 px of one
    ln = loc32.cpu().numpy()
    ok = np.ones(ln.shape[:-1], bool)
    for lvl, (h, w) in enumerate(shapes_l):
        for axis, size in ((0, w), (1, h)):
            px = ln[:, :, :, lvl, :, axis].astype(np.float64) * size - 0.5
            ok[:, :, :, lvl] &= np.abs(px - np.round(px)) > 1e-3
    d = np.abs((gl - rgl) * ok[..., None]).max()
    if not d <= 3e-4 * max(1.0, np.abs(rgl).max()):
        return f"grad_loc: {d:.4g} of {np.abs(rgl).max():.4g}"
    # --- backward with bf16 values / grad_out (gradients fp32: the wide kernel for encoder-shaped launches, the generic one otherwise),
    #     against the fp32 result on the same bf16-rounded operands
    vbf, gobf = v32.bfloat16(), go.bfloat16()
    gvb, glb, gab = alo_hip.msda_backward(vbf, shapes, start, loc32.bfloat16(), attn32.bfloat16(), gobf, 64)
    gvr, glr, gar = alo_hip.msda_backward(vbf.float(), shapes, start, loc32.bfloat16().float(), attn32.bfloat16().float(), gobf.float(), 64)
    if not (gvb.float() - gvr).abs().max().item() <= 1e-2 * max(1.0, gvr.abs().max().item()):   # the wrapper narrows grad_value to bf16
        return "bf16 backward: grad_value"
    if not (gab.float() - gar).abs().max().item() <= 1e-2 * max(1.0, gar.abs().max().item()):
        return "bf16 backward: grad_attn"
    if not (glb.float() - glr).abs().max().item() <= 1e-2 * max(1.0, glr.abs().max().item()):
        return "bf16 backward: grad_loc"
    # --- backward at D = 64 (two value tensors side by side): each 32-channel half must reproduce the D = 32 sums of its own half
    if c["encoder"] and c["N"] <= 2:
        v64 = torch.cat([v32, v32.flip(-1) * 0.5], -1).contiguous()
        go64 = torch.cat([go.view(N, Lq, 8, 32), go.view(N, Lq, 8, 32).flip(-1) * 0.25], -1).reshape(N, Lq, 512).contiguous()
        gv64, gl64, ga64 = (x.cpu().numpy() for x in alo_hip.msda_backward(v64, shapes, start, loc32, attn32, go64, 64))
        r64 = O.msda_backward(v64.double().cpu().numpy(), shapes_np, start_np, loc32.double().cpu().numpy(),
                              attn32.double().cpu().numpy(), go64.double().cpu().numpy())
        if not np.abs(gv64 - r64[0]).max() <= 3e-4 * max(1.0, np.abs(r64[0]).max()):
            return "D = 64 backward: grad_value"
        if not np.abs(ga64 - r64[2]).max() <= 3e-4 * max(1.0, np.abs(r64[2]).max()):
            return "D = 64 backward: grad_attn"
        if not np.abs((gl64 - r64[1]) * ok[..., None]).max() <= 3e-4 * max(1.0, np.abs(r64[1]).max()):
            return "D = 64 backward: grad_loc"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    O.build()
    rng = np.random.default_rng(a.seed)
    t0, n = time.time(), 0
    while time.time() - t0 < a.seconds:
        c = case(rng)
        msg = run_case(c)
        n += 1
        if msg:
            print(f"FAIL case {n}: {msg}; shapes {c['shapes_l']} N {c['N']} Lq {c['Lq']} encoder {c['encoder']} spread {c['spread']}")
            return 1
    print(f"{n} cases, no failure")
    return 0


if __name__ == "__main__":
    sys.exit(main())
