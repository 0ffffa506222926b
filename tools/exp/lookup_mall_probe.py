#!/usr/bin/env python
"""Why the lookup runs 12-17 % slower inside the RAFT step than stand-alone although its L2 counters are identical
(tools/pmc_lookup.sh): stand-alone, back-to-back launches find the strips they read (~168 MB of sectors) in the 256 MB Infinity
Cache; in the step ~600 MB of convolution activations pass through it between two lookups.  Timed here: the same launch (events
around the lookup only) back to back, and with a streaming pass over `--flush-mb` of other memory between launches."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kbench, alo_hip
from alonet.raft.corr import CorrBlock

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=4)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
f1, f2 = kbench.corr_inputs(a.B)
blk = CorrBlock(f1, f2)
H, W = 90, 160
ys, xs = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
coords = (torch.stack([xs, ys]).float()[None] + 4 * torch.randn(a.B, 2, H, W, device="cuda")).contiguous()
for flush_mb in (0, 64, 256, 600, 1200):
    junk = torch.empty(max(flush_mb, 1) * 1024 * 1024 // 4, device="cuda")
    for _ in range(5):
        blk(coords)
    torch.cuda.synchronize()
    times = []
    for _ in range(a.reps):
        if flush_mb:
            junk.add_(1.0)          # reads and writes flush_mb of other memory
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); blk(coords); e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e))
    times.sort()
    print(json.dumps({"flush_mb_between_lookups": flush_mb, "lookup_ms_median": round(times[len(times) // 2], 4),
                      "min": round(times[0], 4)}), flush=True)
