"""Does a HIP graph of the detector's forward beat eager launches?  (dev tool)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aloception-oss_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

dev = torch.device("cuda", 0)
model = bench.build_detector(dev, torch.bfloat16)
frames = bench.detection_inputs(8, 0, dev, torch.bfloat16)

def eager():
    with torch.no_grad():
        return model.inference(model(frames))

for _ in range(3): eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): eager()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3, flush=True)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(3): out = model(frames)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):
    out = model(frames)
torch.cuda.synchronize()
def graphed():
    g.replay()
    with torch.no_grad():
        return model.inference(out)
for _ in range(3): graphed()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): graphed()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
ref = eager()
got = graphed()
print("same boxes:", all(torch.equal(a.as_tensor(), b.as_tensor()) for a, b in zip(ref, got)))
