#!/usr/bin/env python
"""A few launches of the MSDA backward on one sampling distribution (for rocprofv3 passes).  argv: kind policy N reps"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kbench, alo_hip
kind = sys.argv[1] if len(sys.argv) > 1 else "trained"
os.environ["ALO_MSDA_BWD"] = sys.argv[2] if len(sys.argv) > 2 else "wide"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dtype = torch.bfloat16 if (len(sys.argv) > 5 and sys.argv[5] == "bf16") else torch.float32
S = sum(h * w for h, w in kbench.DETR_SHAPES)
value, shapes, start, loc, attn = kbench.msda_inputs(N, S, "encoder" if kind == "ring" else kind, dtype)
go = torch.randn(N, S, 256, device="cuda").to(dtype)
for _ in range(reps):
    alo_hip.msda_backward(value, shapes, start, loc, attn, go)
torch.cuda.synchronize()
