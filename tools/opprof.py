"""Operator-level profile of one DeformableDETR-R50 inference step (torch.profiler; dev tool, not part of the bench)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aloception-oss_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

dev = torch.device("cuda", 0)
if "--raft" in sys.argv:
    from alonet.raft import RAFT

    torch.manual_seed(0)
    model = RAFT().eval().to(dev)
    f1, f2 = bench.flow_inputs(4, 0, dev)

    def step():
        with torch.no_grad():
            return model.inference(model(f1, f2, iters=32, only_last=True), only_last=True)
else:
    model = bench.build_detector(dev, torch.bfloat16)
    frames = bench.detection_inputs(8, 0, dev, torch.bfloat16)

    def step():
        with torch.no_grad():
            return model.inference(model(frames))


for _ in range(2 if "--raft" in sys.argv else 3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(1 if "--raft" in sys.argv else 3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=40, max_shapes_column_width=70))
