"""Operator-level profile of the panoptic leg (dev tool)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aloception-oss_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from alonet.deformable_detr_panoptic import DeformableDetrR50Panoptic
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = DeformableDetrR50Panoptic(num_classes=250, device=dev).eval().to(torch.bfloat16).to(memory_format=torch.channels_last)
frames = bench.detection_inputs(8, 0, dev, torch.bfloat16)
keep = []
for _ in range(8):
    k = torch.zeros(300, dtype=torch.bool, device=dev); k[torch.arange(16, device=dev) * (300 // 16)] = True; keep.append(k)
def step():
    with torch.no_grad():
        return model.inference(model(frames, filters=keep), filters=keep)
try:
    step()
except Exception as e:
    print("step failed:", type(e).__name__, e)
    raise
step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total, str(e.input_shapes)[:110]) for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 50 and e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[2])
for k, c, t, sh in rows[:28]:
    print("%-34s x%-4d %8.1f us  %s" % (k[:34], c, t, sh))
