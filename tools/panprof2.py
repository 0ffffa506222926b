import os, sys
sys.path.insert(0, "/root/repo/aloception-oss_amd"); sys.path.insert(0, "/root/repo")
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
def fwd():
    with torch.no_grad(): return model(frames, filters=keep)
def inf(o):
    with torch.no_grad(): return model.inference(o, filters=keep)
o = fwd(); inf(o); o = fwd(); inf(o); torch.cuda.synchronize()
for name, fn in (("forward", lambda: fwd()), ("inference", lambda: inf(o))):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        fn(); torch.cuda.synchronize()
    ev = prof.key_averages(group_by_input_shape=True)
    tot = sum(e.self_device_time_total for e in ev)
    print("==", name, "device total us", round(tot))
    rows = [(e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:100]) for e in ev if e.self_device_time_total > 80]
    rows.sort(key=lambda r: -r[2])
    for k, c, t, sh in rows[:22]:
        print("  %-40s x%-4d %8.1f us  %s" % (k[:40], c, t, sh))
