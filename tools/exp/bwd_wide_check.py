#!/usr/bin/env python
"""Wide (16x16 block, sorted) against tiled (4x4 wave, window-dense) MSDA backward: same inputs, results and time.

    python tools/exp/bwd_wide_check.py [--N 4] [--kinds ring,survey,trained,uniform] [--dtype f32]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kbench  # noqa: E402
import alo_hip  # noqa: E402


def run(policy, args):
    os.environ["ALO_MSDA_BWD"] = policy
    return alo_hip.msda_backward(*args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4)
    ap.add_argument("--kinds", default="ring,survey,trained,uniform")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--D", type=int, default=32)
    a = ap.parse_args()
    dtype = torch.float32 if a.dtype == "f32" else torch.bfloat16
    S = sum(h * w for h, w in kbench.DETR_SHAPES)
    for kind in a.kinds.split(","):
        value, shapes, start, loc, attn = kbench.msda_inputs(a.N, S, "encoder" if kind == "ring" else kind, dtype)
        if a.D != 32:
            value = torch.randn(a.N, S, 8, a.D, device="cuda").to(dtype)
        go = torch.randn(a.N, S, 8 * a.D, device="cuda").to(dtype)
        args = (value, shapes, start, loc, attn, go)
        ref = run("tiled", args)
        got = run("wide", args)
        torch.cuda.synchronize()
        rec = {"kind": kind, "N": a.N, "dtype": a.dtype}
        for name, r, g in zip(("grad_value", "grad_loc", "grad_attn"), ref, got):
            r, g = r.float(), g.float()
            rec[name + "_maxdiff"] = float((r - g).abs().max())
            rec[name + "_scale"] = float(r.abs().max())
            rec[name + "_nan"] = bool(torch.isnan(g).any())
        for pol in ("tiled", "wide"):
            os.environ["ALO_MSDA_BWD"] = pol
            t = kbench.time_launches(lambda: alo_hip.msda_backward(*args), a.reps)
            rec[pol + "_ms"] = round(t * 1e3, 4)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
