#!/usr/bin/env python
"""Would a head-major fp32 value pay for the generic forward?  The same launch expressed as N*M single-head items
(value (N*M, S, 1, D), loc / attn permuted to (N*M, Lq, 1, L, P[, 2])) runs the generic kernel on 128-byte rows that neighbouring
pixels share lines with — the layout the bf16 fast path uses — against the boundary layout (N, S, M, D)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kbench, alo_hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = sum(h * w for h, w in kbench.DETR_SHAPES)
for kind in ("ring", "survey", "trained"):
    value, shapes, start, loc, attn = kbench.msda_inputs(N, S, "encoder" if kind == "ring" else kind, torch.float32)
    t_pm = kbench.time_launches(lambda: alo_hip.msda_forward(value, shapes, start, loc, attn), 30)
    ref = alo_hip.msda_forward(value, shapes, start, loc, attn)
    v_hm = value.permute(0, 2, 1, 3).reshape(N * 8, S, 1, 32).contiguous()
    l_hm = loc.permute(0, 2, 1, 3, 4, 5).reshape(N * 8, S, 1, 4, 4, 2).contiguous()
    a_hm = attn.permute(0, 2, 1, 3, 4).reshape(N * 8, S, 1, 4, 4).contiguous()
    t_hm = kbench.time_launches(lambda: alo_hip.msda_forward(v_hm, shapes, start, l_hm, a_hm), 30)
    out = alo_hip.msda_forward(v_hm, shapes, start, l_hm, a_hm).view(N, 8, S, 32).permute(0, 2, 1, 3).reshape(N, S, 256)
    t_tr = kbench.time_launches(lambda: value.permute(0, 2, 1, 3).contiguous(), 30)
    print(json.dumps({"kind": kind, "N": N, "pixel_major_ms": round(t_pm * 1e3, 4), "head_major_ms": round(t_hm * 1e3, 4),
                      "transpose_ms": round(t_tr * 1e3, 4), "maxdiff": float((out - ref).abs().max())}), flush=True)
