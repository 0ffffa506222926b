"""Graph capture with a live RCCL process group (world_size 1): does the watchdog thread disturb the capture?  (dev tool)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aloception-oss_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
from alonet.common import GraphedForward
dev = torch.device("cuda", 0)
model = bench.build_detector(dev, torch.bfloat16)
frames = bench.detection_inputs(2, 0, dev, torch.bfloat16)
g = GraphedForward(model)
with torch.no_grad():
    out = g(frames); want = model(frames)
    dist.barrier()
    out2 = g(frames)
print("capture with live process group ok:", torch.equal(out2["pred_boxes"], want["pred_boxes"]))
dist.destroy_process_group()
