#!/usr/bin/env python
"""Timing of the wide backward with parts switched off (ALO_WIDE_DBG bits; results are wrong on purpose).

Needs a development build of the library:  touch csrc/msda_bwd_wide.hip && make -C aloception-oss_amd/csrc EXTRA=-DALO_WIDE_DBG"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kbench, alo_hip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kinds = (sys.argv[2] if len(sys.argv) > 2 else "ring,trained").split(",")
modes = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,3,4,7,8,15").split(",")]
S = sum(h * w for h, w in kbench.DETR_SHAPES)
os.environ["ALO_MSDA_BWD"] = "wide"
for kind in kinds:
    value, shapes, start, loc, attn = kbench.msda_inputs(N, S, "encoder" if kind == "ring" else kind, torch.float32)
    go = torch.randn(N, S, 256, device="cuda")
    out = {}
    for d in modes:
        os.environ["ALO_WIDE_DBG"] = str(d)
        out[d] = round(kbench.time_launches(lambda: alo_hip.msda_backward(value, shapes, start, loc, attn, go), 20) * 1e3, 4)
    print(kind, json.dumps(out), flush=True)
