"""Where does the LDS-resident forward differ from the plain head-major kernel?  (dev script)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "aloception-oss_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import alo_hip
from test_msda_gpu import _bench_path_case, _resident_case
from helpers import DETR_SHAPES
for name, case in (("full", lambda: _bench_path_case(8, True, DETR_SHAPES, 32, True)),
                   ("small", lambda: _resident_case(2, [(40, 50), (20, 25), (10, 13), (5, 7)], None, 2, 43))):
    value, mask, offsets, logits, ref, shapes, start = case()
    vhm = alo_hip.value_head_major(value, mask)
    a = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref).float()
    b = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=False).float()
    a2 = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref).float()
    d = (a - b).abs()
    bad = d > 0
    print(name, "mismatch", int(bad.sum()), "of", d.numel(), "max", float(d.max()), "rel max", float((d / (b.abs() + 1e-6)).max()), "run-to-run equal", bool(torch.equal(a, a2)))
    if bad.any():
        idx = torch.nonzero(bad)
        print(" first", idx[:8].tolist())
        qs = idx[:, 1]
        print(" queries with mismatch:", int(qs.unique().numel()), "min/max q", int(qs.min()), int(qs.max()))
        N, Lq, C = a.shape
        heads = (idx[:, 2] // 32)
        print(" per head", torch.bincount(heads, minlength=8).tolist(), " per image", torch.bincount(idx[:, 0], minlength=N).tolist())
        print(" sample diffs", [(float(a[tuple(i)]), float(b[tuple(i)])) for i in idx[:6]])

print("---- which levels carry the difference (logits of the other levels at -60) ----")
value, mask, offsets, logits, ref, shapes, start = _resident_case(2, [(40, 50), (20, 25), (10, 13), (5, 7)], None, 2, 43)
vhm = alo_hip.value_head_major(value, mask)
for keep in ([0], [1], [2], [3], [0, 1], [2, 3]):
    lg = logits.clone().view(*logits.shape[:-1], 4, 4)
    for l in range(4):
        if l not in keep:
            lg[..., l, :] = -60.0
    lg = lg.view(logits.shape)
    a = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, lg, ref).float()
    b = alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, lg, ref, resident=False).float()
    print(keep, "mismatch", int(((a - b).abs() > 0).sum()))
