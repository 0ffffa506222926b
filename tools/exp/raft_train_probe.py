#!/usr/bin/env python
"""Development probe (round 4): one RAFT training step (forward + backward, 12 iterations, FlyingChairs-size crops 368 x 496, batch 6)
with the HIP CorrBlock (kernel forward, adjoint-kernel + GEMM backward) against TorchCorrBlock (the reference's formulation on stock
torch ops, differentiated by autograd), and the correlation block alone (build + 12 lookups + backward) at the same size.

    python tools/exp/raft_train_probe.py [--batch 6] [--iters 12] [--reps 5] [--height 368 --width 496] [--block-only]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "aloception-oss_amd"))
import aloscene  # noqa: E402
from alonet.raft import RAFT  # noqa: E402
from alonet.raft.corr import CorrBlock, TorchCorrBlock  # noqa: E402
from alonet.raft.utils.utils import coords_grid  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--height", type=int, default=368)
    ap.add_argument("--width", type=int, default=496)
    ap.add_argument("--block-only", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    H, W = a.height, a.width
    f1 = torch.rand(a.batch, 3, H, W) * 2 - 1
    f2 = torch.roll(f1, shifts=(2, -3), dims=(2, 3))
    mk = lambda x: aloscene.Frame(x, normalization="minmax_sym", names=("B", "C", "H", "W")).to(DEV)  # noqa: E731
    fr1, fr2 = mk(f1), mk(f2)
    for name, cls in (() if a.block_only else (("hip", CorrBlock), ("torch", TorchCorrBlock))):
        model = RAFT(corr_block=cls).to(DEV).train()
        model.freeze_bn()

        def step():
            model.zero_grad(set_to_none=True)
            outs = model(fr1, fr2, iters=a.iters)
            sum(o["up_flow"].abs().mean() for o in outs).backward()

        torch.cuda.reset_peak_memory_stats()
        ms = timed(step, a.reps)
        print(json.dumps({"probe": "RAFT training step", "corr_block": name, "batch": a.batch, "iters": a.iters, "crop": [H, W],
                          "ms": round(ms, 2), "peak_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
        del model
        torch.cuda.empty_cache()
    # the block alone
    h, w = H // 8, W // 8
    gen = torch.Generator(device="cpu").manual_seed(1)
    a1 = torch.randn(a.batch, 256, h, w, generator=gen).to(DEV)
    a2 = torch.randn(a.batch, 256, h, w, generator=gen).to(DEV)
    cs = [coords_grid(a.batch, h, w, device=DEV) + torch.randn(a.batch, 2, h, w, generator=gen).to(DEV) * 2 for _ in range(a.iters)]
    ws = [torch.randn(a.batch, 324, h, w, generator=gen).to(DEV) for _ in range(a.iters)]
    for name, cls in (("hip", CorrBlock), ("torch", TorchCorrBlock)):
        def block():
            x, y = a1.clone().requires_grad_(True), a2.clone().requires_grad_(True)
            blk = cls(x, y)
            sum((blk(c) * w_).sum() for c, w_ in zip(cs, ws)).backward()

        torch.cuda.reset_peak_memory_stats()
        ms = timed(block, a.reps)
        print(json.dumps({"probe": "correlation block alone: build + lookups + backward", "corr_block": name, "batch": a.batch,
                          "lookups": a.iters, "grid": [h, w], "ms": round(ms, 2),
                          "peak_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)


if __name__ == "__main__":
    main()
