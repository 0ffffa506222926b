"""Randomised differential run of the mask-decoder kernels of round 4 (alo_conv3x3_small_nhwc, alo_groupnorm_rows_act,
alo_upsample_add_nhwc) against the stock torch ops on the same bf16 values (a development tool; the committed parity tests are
tests/test_fused_gpu.py).      python tools/exp/fuzz_maskhead.py [--seconds 120] [--seed 0]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "aloception-oss_amd"))
import alo_hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(a.seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g, device="cuda").item())  # noqa: E731
    cl = lambda t: t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    t0, cases = time.time(), 0
    with torch.no_grad():
        while time.time() - t0 < a.seconds:
            kind = cases % 3
            if kind == 0:
                n, cin, h, w = ri(1, 5), (16, 32, 64)[ri(0, 2)], ri(1, 70), ri(1, 70)
                cout = (1, 4, 8, 12, 16, 20, 24, 28, 32)[ri(0, 8)]
                x = cl(torch.randn(n, cin, h, w, device="cuda", generator=g))
                conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=bool(ri(0, 1))).cuda().to(torch.bfloat16).to(memory_format=torch.channels_last)
                conv.weight.copy_(torch.randn(conv.weight.shape, device="cuda", generator=g) / (9 * cin) ** 0.5)
                ref = F.conv2d(x.float(), conv.weight.float(), None if conv.bias is None else conv.bias.float(), padding=1)
                got = alo_hip.conv3x3_small(x, conv).float()
                tol = 2.0 ** -8 * max(1.0, ref.abs().max().item()) + 1e-3
            elif kind == 1:
                groups = 8
                c = groups * (2, 4, 8, 16)[ri(0, 3)]
                n, h, w = ri(1, 6), ri(1, 60), ri(1, 60)
                x = cl(torch.randn(n, c, h, w, device="cuda", generator=g) * (0.1 + 3 * torch.rand(1, device="cuda", generator=g)) + ri(-2, 2))
                norm = torch.nn.GroupNorm(groups, c).cuda().to(torch.bfloat16)
                norm.weight.copy_(torch.randn(c, device="cuda", generator=g))
                norm.bias.copy_(torch.randn(c, device="cuda", generator=g))
                relu = bool(ri(0, 1))
                ref = F.group_norm(x.float(), groups, norm.weight.float(), norm.bias.float(), norm.eps)
                ref = F.relu(ref) if relu else ref
                got = alo_hip.groupnorm_nhwc(x, norm, relu=relu).float()
                tol = 2.0 ** -8 * max(1.0, ref.abs().max().item()) + 2e-3
            else:
                b, q, c = ri(1, 3), ri(1, 6), 8 * ri(1, 16)
                h, w = ri(1, 30), ri(1, 30)
                H, W = ri(h, 3 * h), ri(w, 3 * w)
                x = cl(torch.randn(b * q, c, h, w, device="cuda", generator=g))
                fpn = cl(torch.randn(b, c, H, W, device="cuda", generator=g))
                ref = (fpn.unsqueeze(1).repeat(1, q, 1, 1, 1).flatten(0, 1) + F.interpolate(x, size=(H, W), mode="nearest")).float()
                got = alo_hip.upsample_add(x, fpn).float()
                tol = 0.0
            err = (got - ref).abs().max().item() if ref.numel() else 0.0
            if not err <= tol:
                print("FAIL", kind, tuple(ref.shape), err, tol)
                sys.exit(1)
            cases += 1
    print(f"{cases} cases, no failure")


if __name__ == "__main__":
    main()
